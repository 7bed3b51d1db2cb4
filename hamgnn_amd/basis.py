"""Orbital-basis tables of the read-out head (data extracted from the reference: hamgnn/models/hamgnn_output.py:345-810;
row shells, index_change, minus_index, basis_def per (ham_type, nao_max)) shipped as package data."""
import json
import os

_P = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def basis_table(ham_type: str, nao_max: int):
    with open(os.path.join(_P, "basis_tables.json")) as f:
        t = json.load(f)
    key = f"{ham_type.lower()}_{nao_max}"
    if key not in t:
        raise NotImplementedError(f"ham_type={ham_type!r} nao_max={nao_max} not supported")      # hamgnn_output.py:343,526,594,810
    e = t[key]
    return {"row": e["row"], "index_change": e["index_change"], "minus_index": e.get("minus_index"),
            "basis_def": {int(k): v for k, v in e["basis_def"].items()},
            "num_valence": {int(k): v for k, v in e.get("num_valence", {}).items()}}


def atomic_radii(kind="openmx"):
    with open(os.path.join(_P, "atomic_radii.json")) as f:
        return json.load(f)[kind]

"""k_path='auto' (hamgnn_output.py:3812-3841 of the reference: pymatgen's KPathSeek).  pymatgen is not in the image: the wiring is pinned against stub modules
that record what they are given (Structure in Angstrom with element symbols, Cartesian coordinates) and return a path; without pymatgen the call raises."""
import sys
import types

import numpy as np
import pytest
import torch

from hamgnn_amd import kspace
from hamgnn_amd.data import synthetic as S


def _stub_pymatgen(monkeypatch, record, path, kpoints):
    class Element:
        _SYM = {1: "H", 6: "C", 8: "O", 14: "Si"}

        def __init__(self, symbol):
            self.symbol = symbol

        @classmethod
        def from_Z(cls, z):
            return cls(cls._SYM[int(z)])

    class Structure:
        def __init__(self, lattice, species, coords, coords_are_cartesian=False):
            record.append(dict(lattice=np.asarray(lattice), species=list(species), coords=np.asarray(coords), cart=coords_are_cartesian))

    class KPathSeek:
        def __init__(self, structure):
            self.kpath = {"path": path, "kpoints": kpoints}
    mods = {"pymatgen": types.ModuleType("pymatgen"), "pymatgen.core": types.ModuleType("pymatgen.core"), "pymatgen.symmetry": types.ModuleType("pymatgen.symmetry"),
            "pymatgen.core.periodic_table": types.ModuleType("pymatgen.core.periodic_table"), "pymatgen.core.structure": types.ModuleType("pymatgen.core.structure"),
            "pymatgen.symmetry.kpath": types.ModuleType("pymatgen.symmetry.kpath")}
    mods["pymatgen.core.periodic_table"].Element = Element
    mods["pymatgen.core.structure"].Structure = Structure
    mods["pymatgen.symmetry.kpath"].KPathSeek = KPathSeek
    for k, m in mods.items():
        monkeypatch.setitem(sys.modules, k, m)


def test_auto_k_path_follows_the_reference_wiring(monkeypatch):
    record = []
    path = [["GAMMA", "X", "M"], ["M", "GAMMA", "R"], ["R", "X"]]          # segments; consecutive repeats (M | M, R | R) are dropped, a later X stays
    kp = {"GAMMA": [0.0, 0.0, 0.0], "X": [0.5, 0.0, 0.0], "M": [0.5, 0.5, 0.0], "R": [0.5, 0.5, 0.5]}
    _stub_pymatgen(monkeypatch, record, path, kp)
    g1, g2 = S.si_diamond(1, 1, 1), S.random_cell(3, [14, 8, 1], seed=2, density=0.01)
    from hamgnn_amd.data import collate
    batch = collate([g1, g2])
    nk = 24
    kv = kspace.make_k_vectors("auto", nk, batch.cell, data=batch)
    nodes = [kp[k] for k in ("GAMMA", "X", "M", "GAMMA", "R", "X")]
    want = kspace.make_k_vectors(nodes, nk, batch.cell)
    assert kv.shape == (2, nk, 3) and torch.equal(kv, want)
    assert len(record) == 2
    for rec, g in zip(record, (g1, g2)):                                  # what pymatgen was handed: Angstrom, symbols, Cartesian
        assert rec["cart"] is True and rec["species"] == [{1: "H", 6: "C", 8: "O", 14: "Si"}[int(z)] for z in g.z]
        assert np.allclose(rec["lattice"], g.cell.reshape(3, 3).double().numpy() * kspace.AU2ANG) and np.allclose(rec["coords"], g.pos.double().numpy() * kspace.AU2ANG)
    # a path that cannot be sampled (more nodes than points): random reduced k-points, as the reference's bare except
    rng = np.random.RandomState(0)
    kv2 = kspace.make_k_vectors("AUTO", 3, g1.cell, rng=rng, data=g1)
    assert kv2.shape == (1, 3, 3) and torch.isfinite(kv2).all()


def test_band_cal_auto_mode_labels_and_nodes(monkeypatch):
    """band_cal's auto_mode (DFT_interfaces/openmx/band_cal.py:135-145): labels with adjacent duplicates removed + their nodes, per crystal"""
    record = []
    _stub_pymatgen(monkeypatch, record, [["GAMMA", "X"], ["X", "M"], ["M", "GAMMA"]], {"GAMMA": [0, 0, 0], "X": [0.5, 0, 0], "M": [0.5, 0.5, 0]})
    g = S.si_diamond(1, 1, 1)
    labels, nodes = kspace.auto_k_path(g.cell.reshape(3, 3).double().numpy(), g.pos.double().numpy(), g.z.numpy())
    assert labels == ["GAMMA", "X", "M", "GAMMA"] and nodes == [[0, 0, 0], [0.5, 0, 0], [0.5, 0.5, 0], [0, 0, 0]]
    assert kspace.auto_k_path_nodes(g.cell.reshape(3, 3).double().numpy(), g.pos.double().numpy(), g.z.numpy()) == nodes
    from hamgnn_amd import band_cal
    with pytest.raises(ValueError, match="auto_mode"):                    # neither nodes nor auto_mode
        band_cal.band_structure([g], None, k_path=None, device="cpu")


def test_auto_k_path_without_pymatgen_raises(monkeypatch):
    for k in [m for m in sys.modules if m.startswith("pymatgen")]:
        monkeypatch.delitem(sys.modules, k)
    monkeypatch.setitem(sys.modules, "pymatgen", None)                    # import pymatgen -> ImportError
    g = S.si_diamond(1, 1, 1)
    with pytest.raises(NotImplementedError, match="pymatgen"):
        kspace.make_k_vectors("auto", 8, g.cell, data=g)
    with pytest.raises(NotImplementedError):
        kspace.make_k_vectors("gamma-x", 8, g.cell, data=g)

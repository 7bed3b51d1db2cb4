"""Kernel tables as a FILE, for hosts that are not Python ("a host without the planner").

The C ABI (include/hamgnn_hip.h) takes launch tables that only `hamgnn_amd/plan.py` knows how to build.  A host written in C / C++ / Go does not need
the planner at run time: the tables of a model are a function of its irreps and weights, so they can be produced once (here) and loaded (there).
`export_tp_is(dp, path, rows)` writes what ONE `hg_tp_is` launch of a MessagePackBlock program needs; `include/hamgnn_tables.h` documents the container and
carries a 60-line C loader; `examples/run_tp_is.c` is a complete host: load, upload, launch, compare.

Container (little-endian): 8 bytes magic "HGPROG1\\0" | uint64 header length | JSON header | 64-byte aligned raw arrays.  The header lists, per array, its name,
dtype ("f32" / "i32"), shape, byte offset from the start of the file and byte length, and the scalar launch parameters (hidden, out_dim, lds_bytes, nparts)."""
from __future__ import annotations

import json
import struct

import numpy as np

MAGIC = b"HGPROG1\0"
ARRAYS = ("weights", "seg_table", "block_table", "phase_table", "group_table", "item_table", "part_table", "row_table")


def tp_is_arrays(prog, sched, weights: np.ndarray) -> dict:
    """the eight tables of a hg_tp_is launch as numpy arrays (plan.Program + plan.IsSchedule)"""
    return {"weights": np.ascontiguousarray(weights, dtype=np.float32), "seg_table": sched.seg_table, "block_table": sched.block_table,
            "phase_table": sched.phase_table, "group_table": sched.group_table, "item_table": sched.item_table, "part_table": sched.part_table,
            "row_table": sched.rowtab}


def write_container(path: str, arrays: dict, scalars: dict) -> dict:
    """write the container; returns the header"""
    entries, blobs = [], []
    for name in ARRAYS:
        a = np.ascontiguousarray(arrays[name])
        a = a.astype("<f4") if a.dtype.kind == "f" else a.astype("<i4")
        entries.append({"name": name, "dtype": "f32" if a.dtype.kind == "f" else "i32", "shape": list(a.shape), "offset": 10 ** 11, "nbytes": int(a.nbytes)})
        blobs.append(a.tobytes())
    header = dict(scalars, format=1, arrays=entries)
    hlen = len(json.dumps(header).encode())                    # with 12-digit placeholder offsets: the final text is never longer
    hlen += (-(16 + hlen)) % 64
    pos = 16 + hlen
    for e, b in zip(entries, blobs):
        pos += (-pos) % 64
        e["offset"] = pos
        pos += len(b)
    raw = json.dumps(header).encode()
    assert len(raw) <= hlen
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", hlen))
        f.write(raw + b" " * (hlen - len(raw)))
        for e, b in zip(entries, blobs):
            f.write(b"\0" * (e["offset"] - f.tell()))
            f.write(b)
    return header


def read_container(path: str):
    """(header, {name: array}) -- what include/hamgnn_tables.h:hg_prog_load does in C"""
    with open(path, "rb") as f:
        data = f.read()
    assert data[:8] == MAGIC, "not a HGPROG1 container"
    (hlen,) = struct.unpack("<Q", data[8:16])
    header = json.loads(data[16:16 + hlen].decode())
    out = {}
    for e in header["arrays"]:
        dt = "<f4" if e["dtype"] == "f32" else "<i4"
        out[e["name"]] = np.frombuffer(data, dtype=dt, count=e["nbytes"] // 4, offset=e["offset"]).reshape(e["shape"])
    return header, out


def export_tp_is(dp, path: str, rows: int) -> dict:
    """tables of the hg_tp_is launch `ops.tp_fused(dp, ..., rows)` would make (dp: ops.DeviceProgram with an input-stationary schedule)"""
    parts = dp.is_parts_for(rows)
    sc = dp.is_tables(parts)[0]
    # the blob the launch READS: the device copy (after nn.MessagePackBlock.refresh -- the device-side repack that follows an optimiser step -- only that one is
    # current; dp.prog.weights is the host blob of compile time: ADVICE r5), with the schedule's own streams behind it
    w = dp.is_weights(parts).detach().cpu().numpy()
    scalars = {"entry": "hg_tp_is", "hidden": int(dp.hidden), "out_dim": int(dp.out_dim), "lds_bytes": int(sc.lds_floats * 4), "nparts": int(sc.part_table.shape[0]),
               "zero_fill_out": bool(getattr(sc, "atomic_out", False)), "rows_planned_for": int(rows)}
    return write_container(path, tp_is_arrays(dp.prog, sc, w), scalars)

"""hamgnn_amd/export.py + include/hamgnn_tables.h: the launch tables of a MessagePackBlock program as a file that a C host loads without the
Python planner (VERDICT r4 weak point 10).  CPU: the container round-trips in Python, and the header-only C loader (compiled with gcc) reads the same
scalars, shapes and bytes."""
import hashlib
import os
import shutil
import subprocess

import numpy as np
import pytest

from hamgnn_amd import export as X, plan as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINI, SH = "8x0e+4x0o+4x1o+2x1e+2x2o+3x2e+2x3o", "0e+1o+2e+3o"

C_MAIN = r"""
#include "hamgnn_tables.h"
static unsigned long long fnv(const void* p, long long n) { const unsigned char* c = p; unsigned long long h = 1469598103934665603ULL; for (long long i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ULL; } return h; }
int main(int argc, char** argv) {
    HgProgFile f;
    int rc = hg_prog_load(argv[1], &f);
    if (rc) { printf("load failed %d\n", rc); return 1; }
    printf("hidden %d out_dim %d lds_bytes %d nparts %d zero_fill_out %d\n", f.hidden, f.out_dim, f.lds_bytes, f.nparts, f.zero_fill_out);
    const HgArray* a[8] = {&f.weights, &f.seg_table, &f.block_table, &f.phase_table, &f.group_table, &f.item_table, &f.part_table, &f.row_table};
    const char* n[8] = {"weights", "seg_table", "block_table", "phase_table", "group_table", "item_table", "part_table", "row_table"};
    for (int i = 0; i < 8; ++i) printf("%s %d %lld %lld %lld %llu\n", n[i], a[i]->is_f32, (long long)a[i]->shape[0], (long long)a[i]->shape[1], (long long)a[i]->nbytes, fnv(a[i]->data, a[i]->nbytes));
    hg_prog_free(&f);
    return 0;
}
"""


def _fnv(b: bytes) -> int:
    h = 1469598103934665603
    for c in b:
        h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def _program():
    import torch
    from hamgnn_amd import nn as hnn
    torch.manual_seed(0)
    m = hnn.MessagePackBlock(MINI, MINI, SH, MINI, 8, [16, 16])
    prog = P.build_message_pack_program(hnn._np_sd(m), MINI, MINI, SH, MINI, False)
    return prog


@pytest.mark.parametrize("parts", [1, 3, ("2d", 3, 2)], ids=["single", "split", "shared_segments"])
def test_container_round_trip_and_c_loader(tmp_path, parts):
    prog = _program()
    sc = P.is_schedule(prog, parts)
    arrays = X.tp_is_arrays(prog, sc, prog.weights)
    path = str(tmp_path / "block.hgprog")
    hdr = X.write_container(path, arrays, {"entry": "hg_tp_is", "hidden": int(prog.hidden_pad), "out_dim": int(prog.out_layout.dim), "lds_bytes": int(sc.lds_floats * 4),
                                           "nparts": int(sc.part_table.shape[0]), "zero_fill_out": bool(sc.atomic_out)})
    h2, back = X.read_container(path)
    assert h2 == hdr and set(back) == set(X.ARRAYS)
    for k in X.ARRAYS:
        assert back[k].shape == np.asarray(arrays[k]).shape and np.array_equal(back[k], np.asarray(arrays[k]).astype(back[k].dtype)), k
        assert h2["arrays"][X.ARRAYS.index(k)]["offset"] % 64 == 0
    if shutil.which("gcc") is None:
        pytest.skip("gcc not found")
    src = tmp_path / "main.c"
    src.write_text(C_MAIN)
    exe = str(tmp_path / "loader")
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe], check=True)
    out = subprocess.run([exe, path], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert out[0] == f"hidden {int(prog.hidden_pad)} out_dim {int(prog.out_layout.dim)} lds_bytes {int(sc.lds_floats * 4)} nparts {int(sc.part_table.shape[0])} zero_fill_out {int(bool(sc.atomic_out))}"
    for line, k in zip(out[1:], X.ARRAYS):
        name, is_f32, s0, s1, nbytes, h = line.split()
        a = back[k]
        assert name == k and int(is_f32) == int(a.dtype.kind == "f") and int(s0) == a.shape[0] and int(s1) == (a.shape[1] if a.ndim > 1 else 1)
        assert int(nbytes) == a.nbytes and int(h) == _fnv(a.tobytes()), k


def test_c_loader_refuses_corrupt_containers(tmp_path):
    """a corrupt header length must not wrap `16 + hlen`, an array's byte count must equal its shape product x 4, offsets must lie inside the
    file -- the loader returns an error (and frees what it allocated: run under the sanitizers when gcc has them) instead of reading out of bounds"""
    import struct
    if shutil.which("gcc") is None:
        pytest.skip("gcc not found")
    prog = _program()
    sc = P.is_schedule(prog, 1)
    good = str(tmp_path / "good.hgprog")
    X.write_container(good, X.tp_is_arrays(prog, sc, prog.weights), {"entry": "hg_tp_is", "hidden": 16, "out_dim": 8, "lds_bytes": 1024, "nparts": 1, "zero_fill_out": False})
    data = open(good, "rb").read()
    (hlen,) = struct.unpack("<Q", data[8:16])
    hdr = data[16:16 + hlen].decode()
    cases = {"good": data,
             "hlen_wraps": data[:8] + struct.pack("<Q", 2 ** 64 - 8) + data[16:],
             "hlen_past_eof": data[:8] + struct.pack("<Q", len(data)) + data[16:],
             "truncated": data[:len(data) // 2],
             "bad_magic": b"HGPROG2\0" + data[8:]}
    e = json_entry = None
    import json
    h = json.loads(hdr)
    e = h["arrays"][1]
    for name, (key, val) in {"nbytes_ne_shape": ("nbytes", e["nbytes"] - 4), "offset_past_eof": ("offset", (len(data) + 64) // 64 * 64)}.items():
        h2 = json.loads(hdr)
        h2["arrays"][1][key] = val
        raw = json.dumps(h2).encode()
        assert len(raw) <= hlen
        cases[name] = data[:16] + raw + b" " * (hlen - len(raw)) + data[16 + hlen:]
    src = tmp_path / "main.c"
    src.write_text(C_MAIN)
    exe = str(tmp_path / "loader")
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    if subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror"] + san + ["-I", os.path.join(ROOT, "include"), str(src), "-o", exe], capture_output=True).returncode:
        subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe], check=True)
    for name, blob in cases.items():
        pth = str(tmp_path / (name + ".hgprog"))
        open(pth, "wb").write(blob)
        cp = subprocess.run([exe, pth], capture_output=True, text=True)
        if name == "good":
            assert cp.returncode == 0, cp.stdout + cp.stderr
        else:
            assert cp.returncode == 1 and "load failed" in cp.stdout and "ERROR" not in cp.stderr, (name, cp.stdout, cp.stderr[-600:])

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

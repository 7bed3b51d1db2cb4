"""One-off GPU sweep (not a pytest): data gradient of the fused MessagePackBlock for random irreps sets with l up to 6 and sh up to l = 5
(merged items, odd-column skip, multi-part adjoint schedules) vs torch.autograd through the fp64 oracle.  python tests/sweep_random_irreps.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import gpu_checks as G
from tests.test_plan_emu import _random_irreps
from hamgnn_amd import plan as P
torch.set_num_threads(16)
worst_f = worst_b = 0.0; nmerge = 0; bad = []
for seed in range(30):
    rng = np.random.default_rng(900 + seed)
    lmax = int(rng.integers(4, 7))
    irr = _random_irreps(rng, lmax)
    if "0e" not in irr: irr = "5x0e+" + irr
    lsh = int(rng.integers(3, 6))
    sh = "+".join(f"{l}{'e' if l % 2 == 0 else 'o'}" for l in range(lsh + 1))
    nmerge += bool(P.choose_merge_groups(irr, irr, sh, irr, 16))
    try:
        rf = G.check_message_pack_random(seed=seed, irr=irr, sh=sh)
        worst_f = max(worst_f, rf["rel_err"])
        if rf["rel_err"] > 1e-5: bad.append(("fwd", irr, sh, rf))
        r = G.check_message_pack_backward(seed=seed, irr=irr, sh=sh, E=40)
        worst_b = max(worst_b, r["g_src_rel_err"], r["g_dst_rel_err"], r["g_edge_rel_err"])
        if worst_b > 1e-5: bad.append(("bwd", irr, sh, r))
    except NotImplementedError as e:
        print("skip", irr, str(e)[:60])
print("forward worst", worst_f, "backward worst", worst_b, "programs with merge groups", nmerge, "bad", bad[:2])

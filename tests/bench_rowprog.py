"""Micro-benchmark of the fused HamLayer row program (csrc/rowprog.hip) on synthetic planar rows: ms per launch; with a -DHG_PROF build
(HG_LIB_PATH) the per-stage cycle counts of wave 0 of workgroup 0."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hamgnn_amd import ops, plan as P
from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=822350); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--irreps", default="A")
a = ap.parse_args()
irr = bench.IRREPS[a.irreps]
torch.manual_seed(0)
dev = torch.device("cuda")
head = HamGNNPlusPlusOut(irr, irr, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=False, soc_switch=False, calculate_sparsity=False)
head.compile(dev)
hl = head.offsite_hamiltonian_network
lay = P.PlanarLayout(irr)
x = torch.randn(a.rows, lay.dim, device=dev)
for mode in ("1", "0"):
    os.environ["HG_ROWPROG"] = mode
    hl._rowprog = None
    for _ in range(2):
        y = hl(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        y = hl(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print(json.dumps({"rowprog": mode, "rows": a.rows, "ms": dt * 1e3, "GBs_algorithmic": a.rows * (lay.dim + y.shape[1]) * 4 / dt / 1e9, "checksum": float(y.double().abs().mean())}))
os.environ["HG_ROWPROG"] = "1"
hl._rowprog = None
if os.environ.get("HG_PROF"):
    import ctypes as C
    from hamgnn_amd import _lib
    L = _lib.lib()
    buf = (C.c_ulonglong * 16)()
    L.hg_prof_rp_read(buf, 1)
    hl(x); torch.cuda.synchronize()
    L.hg_prof_rp_read(buf, 0)
    names = ["stage-in + barrier", "L1", "L1 barrier", "gate", "gate barrier", "L2", "L2 barrier", "L3", "L3 barrier", "", "", "", "write-out", "end barrier"]
    tot = float(sum(buf))
    print(json.dumps({"total_cycles": tot, **{n: round(buf[k] / tot, 4) for k, n in enumerate(names) if n}}))

"""Micro-benchmark of the E-row o3.Linears of the read-out head (ResidualBlock linear1 877 -> 1012, linear2 877 -> 877 + residual,
HamLayer 877 -> ham irreps) on 822 350 rows: streaming kernel (csrc/linear.hip) vs the segment-stationary program kernel."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hamgnn_amd import nn as hnn, ops, plan as P
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=822350); ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
irr = "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e"
dev = torch.device("cuda")
gin, gout, _ = P.gate_tables(P.Irreps(irr))
x = torch.randn(a.rows, P.PlanarLayout(irr).dim, device=dev)
res = {}
for name, (i, o, use_res) in {"linear1 877->1012": (irr, gin, False), "linear2 877->877 + residual": (gout, irr, True)}.items():
    for kern in ("stream", "seg"):
        os.environ["HG_LINEAR_KERNEL"] = kern
        torch.manual_seed(0)
        lin = hnn.E3Linear(i, o).compile(dev)
        xin = x if P.PlanarLayout(i).dim == x.shape[1] else torch.randn(a.rows, P.PlanarLayout(i).dim, device=dev)
        r = [torch.randn(a.rows, P.PlanarLayout(o).dim, device=dev)] if use_res else []
        for _ in range(3): y = lin(xin, res=r)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.reps): y = lin(xin, res=r)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / a.reps
        nbytes = a.rows * 4 * (P.PlanarLayout(i).dim + P.PlanarLayout(o).dim * (2 if use_res else 1))
        res[f"{name} [{kern}]"] = {"ms": round(ms, 3), "GBs": round(nbytes / ms / 1e6, 1), "checksum": float(y.double().abs().mean())}
print(json.dumps(res, indent=1))

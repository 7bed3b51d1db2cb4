"""Micro-benchmark of ONE launch of the fused weight-gradient kernel (csrc/tp_wgrad.hip) for a set-A / set-B MessagePackBlock on synthetic
edge-frame rows: ms per launch and issued-MFMA TFLOP/s for several split counts (HG_LIB_PATH selects the .so)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hamgnn_amd import nn as hnn, ops, plan as P, backward_mp as BM
IRR = {"A": "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e", "B": "64x0e+32x1o+16x1e+8x2o+20x2e+8x3o+4x3e+4x4e"}
ap = argparse.ArgumentParser(); ap.add_argument("--irreps", default="A"); ap.add_argument("--edges", type=int, default=44032)
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--splits", default="6,16,32,64"); ap.add_argument("--tag", default="")
a = ap.parse_args()
irr, sh = IRR[a.irreps], "0e+1o+2e+3o+4e+5o"
torch.manual_seed(0)
m = hnn.MessagePackBlock(irr, irr, sh, irr, 64, [64, 64])
dev = torch.device("cuda")
sd = {k: v.detach().double().numpy() for k, v in m.state_dict().items()}
wg = BM.MessagePackWeightGrad(sd, irr, irr, sh, irr)
wf = P.build_tp_wgrad_fused(wg.branches, sh, irr, wg.H)
dwf = ops.DeviceWgFused(wf, dev)
E = a.edges
lay = P.PlanarLayout(irr)
g_ = torch.Generator(device="cpu").manual_seed(1)
xs, xd, fe, g = (torch.randn(E, lay.dim, generator=g_).to(dev) for _ in range(4))
hn, he = (torch.randn(E, 64, generator=g_).to(dev) for _ in range(2))
for S in [int(s) for s in a.splits.split(",")]:
    for _ in range(2):
        acc, gs = ops.tp_wgrad(dwf, [xs, xd, fe], g, hn, he, nsplit=S)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        acc, gs = ops.tp_wgrad(dwf, [xs, xd, fe], g, hn, he, nsplit=S)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print(json.dumps({"tag": a.tag, "lib": os.path.basename(os.environ.get("HG_LIB_PATH", "default")), "irreps": a.irreps, "E": E, "nsplit": S, "units": int(wf.units.shape[0]),
                      "ms": dt * 1e3, "issued_TF": wf.mfma_per_tile * 2048 * (E / 16) / dt / 1e12, "checksum": float(acc.sum(0).double().abs().mean()), "gs": float(gs[0].double().abs().mean())}))

"""Micro-benchmark of hg_radial_hidden_multi (csrc/aux_kernels.hip): the 13 weight generators (64 -> 64 -> 64) of a 3-layer backbone on
822 350 basis rows, one launch; checked against torch on a slice.  HG_LIB_PATH selects the .so."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hamgnn_amd import ops
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=822350); ap.add_argument("--nmlp", type=int, default=13)
ap.add_argument("--reps", type=int, default=5); ap.add_argument("--tag", default="")
a = ap.parse_args()
dev = torch.device("cuda"); g = torch.Generator().manual_seed(0)
rbf = torch.randn(a.rows, 64, generator=g).to(dev)
gens = [[(torch.randn(64, 64, generator=g) / 8).to(dev) for _ in range(2)] for _ in range(a.nmlp)]
cst = 1.679
for _ in range(2):
    H = ops.radial_hidden_multi(rbf, gens, cst)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    H = ops.radial_hidden_multi(rbf, gens, cst)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
sl = slice(a.rows - 1000, a.rows)
silu = lambda v: cst * torch.nn.functional.silu(v)
err = max(float((H[m][sl] - silu(silu(rbf[sl] @ gens[m][0]) @ gens[m][1])).abs().max()) for m in (0, a.nmlp - 1))
print(json.dumps({"tag": a.tag, "lib": os.path.basename(os.environ.get("HG_LIB_PATH", "default")), "rows": a.rows, "nmlp": a.nmlp, "ms": dt * 1e3,
                  "TF": a.rows * a.nmlp * 2 * 2 * 64 * 64 / dt / 1e12, "max_err_vs_torch_tail": err, "checksum": float(H.double().abs().mean())}))

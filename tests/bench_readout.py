"""Micro-benchmark of the one-pass read-out kernel (csrc/head.hip: hg_ham_readout) on synthetic coefficient rows: ms per launch for the
off-site rows of a sio2_10k-sized graph (random perfect matching as inverse pairs, random species); HG_LIB_PATH selects the .so."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hamgnn_amd import ops, plan as P
from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, default=822350); ap.add_argument("--atoms", type=int, default=10002)
ap.add_argument("--reps", type=int, default=5); ap.add_argument("--tag", default="")
a = ap.parse_args()
irr = bench.IRREPS["A"]
torch.manual_seed(0)
dev = torch.device("cuda")
head = HamGNNPlusPlusOut(irr, irr, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True, soc_switch=False, calculate_sparsity=False)
head.compile(dev)
E = a.rows - (a.rows & 1)
g = torch.Generator().manual_seed(1)
pos = torch.rand(a.atoms, 3, generator=g) * 40
src = torch.randint(0, a.atoms, (E,), generator=g); dst = torch.randint(0, a.atoms, (E,), generator=g)
shift = torch.randn(E, 3, generator=g) * 3
geo = ops.Geometry(pos.to(dev), torch.stack([src, dst]).to(dev), shift.to(dev), 26.0, 64, 6, torch.from_numpy(P.wigner_jtab(6)).to(dev))
perm = torch.randperm(E, generator=g)
pairs = (perm[: E // 2].to(dev).contiguous(), perm[E // 2:].to(dev).contiguous())
z = torch.randint(1, 20, (a.atoms,), generator=g).to(dev)
n = 19
cwid = head.offsite_hamiltonian_network(torch.randn(16, P.PlanarLayout(irr).dim, device=dev)).shape[1]
coeff = torch.randn(E, cwid, generator=g).to(dev)
H0 = torch.randn(E, n * n, generator=g).to(dev)
out = torch.empty(E, n * n, device=dev)
def launch():
    return ops.ham_readout(coeff, geo, head._slot, *head._cg, n, pairs, H0, head._mask, z, geo.src, geo.dst, out, head.hamiltonian_irreps.lmax, 1.0, True)
for _ in range(2):
    launch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    launch()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
nW = 165
print(json.dumps({"tag": a.tag, "lib": os.path.basename(os.environ.get("HG_LIB_PATH", "default")), "rows": E, "ms": dt * 1e3,
                  "GBs_algorithmic": E * (coeff.shape[1] + nW + 2 * n * n) * 4 / dt / 1e9, "checksum": float(out.double().abs().mean())}))

"""Micro-benchmark of the two attention kernels (csrc/attention.hip) on the receiver CSR of the a-SiO2 10k-atom graph (BASELINE
config #4): ms per call and the HBM rate on the algorithmic bytes (value rows read once + output rows; key rows are cache-resident)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hamgnn_amd import ops, plan as P
from hamgnn_amd.data import synthetic as S
from hamgnn_amd.topo import get_topology
ap = argparse.ArgumentParser(); ap.add_argument("--atoms", type=int, default=10002); ap.add_argument("--heads", type=int, default=4); ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
irr = "64x0e+64x0o+32x1o+16x1e+12x2o+24x2e+16x3o+8x3e+4x4o+8x4e+4x5o+4x5e+4x6e"
dev = torch.device("cuda")
g = S.amorphous_sio2(a.atoms, seed=1).to(dev)
lay = P.PlanarLayout(irr)
tab, hd = P.attention_head_table(irr, a.heads)
tab = torch.from_numpy(tab).to(dev)
geo = ops.Geometry(g.pos, g.edge_index, g.nbr_shift, 26.0, 64, 6, torch.from_numpy(P.wigner_jtab(6)).to(dev))
rowptr, perm = get_topology(g).receiver_csr()
N, E, Dp = g.num_nodes, g.num_edges, lay.dim
K, V = torch.randn(N, Dp, device=dev), torch.randn(E, Dp, device=dev)
cut = torch.tensor([4.0], device=dev)
from hamgnn_amd._lib import lib, ptr, i64, i32, f32, check
st = lambda: __import__("ctypes").c_void_p(torch.cuda.current_stream().cuda_stream)
logits = torch.empty(E, a.heads, device=dev); out = torch.empty(N, Dp, device=dev)
def f1(): check(lib().hg_attn_logits(ptr(K), i64(Dp), ptr(geo.src), ptr(geo.dst), ptr(geo.length), ptr(tab), i32(Dp), i32(a.heads), ptr(cut), f32(26.0), f32(hd ** -0.5), i64(E), ptr(logits), st()))
def f2(): check(lib().hg_attn_aggregate(ptr(logits), i32(a.heads), ptr(V), i64(Dp), ptr(rowptr), ptr(perm), ptr(tab), i64(N), i32(Dp), ptr(out), i64(Dp), st()))
def f3(): ops.segment_sum(V, rowptr, perm, N)
res = {"N": N, "E": E, "Dp": Dp, "heads": a.heads}
for name, fn, nbytes in (("attn_logits", f1, E * (a.heads * 4 + 20)), ("attn_aggregate", f2, E * (Dp * 4 + a.heads * 4 + 8) + N * Dp * 4), ("segment_sum (for scale)", f3, E * (Dp * 4 + 8) + N * Dp * 4)):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.reps): fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / a.reps
    res[name] = {"ms": round(ms, 4), "algorithmic_GBs": round(nbytes / ms / 1e6, 1)}
print(json.dumps(res))

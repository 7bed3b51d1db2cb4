"""Micro-benchmark of ONE fused MessagePackBlock launch (set-A or set-B irreps) on synthetic rotated rows: quick A/B of
kernel variants (HG_LIB_PATH selects the .so).  Prints ms per launch and issued-MFMA TFLOP/s."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hamgnn_amd import nn as hnn, ops, plan as P
IRR = {"A": "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e", "B": "64x0e+32x1o+16x1e+8x2o+20x2e+8x3o+4x3e+4x4e"}
ap = argparse.ArgumentParser(); ap.add_argument("--irreps", default="A"); ap.add_argument("--edges", type=int, default=131072)
ap.add_argument("--reps", type=int, default=5); ap.add_argument("--tag", default=""); ap.add_argument("--same-rows", type=int, default=0, help="alias the B-operand rows onto this many distinct rows (cache-residency experiment)")
ap.add_argument("--nodes", type=int, default=0, help="feed NODE rows + random sender/receiver indices (gather + rotation fused into the input-stationary kernel, or hg_rotate_gather + kernel otherwise) instead of pre-rotated edge rows")
ap.add_argument("--lite", action="store_true", help="lite_mode block (its own instantiation of the input-stationary kernel)")
ap.add_argument("--adjoint", action="store_true", help="time the data-gradient launch (adjoint program, hamgnn_amd.nn.MessagePackBlock.backward_data) instead of the forward")
a = ap.parse_args()
irr, sh = IRR[a.irreps], "0e+1o+2e+3o+4e+5o"
torch.manual_seed(0)
m = hnn.MessagePackBlock(irr, irr, sh, irr, 64, [64, 64], lite_mode=a.lite)
dev = torch.device("cuda")
m.compile(dev, unrotate=True)
E = a.edges
lay = P.PlanarLayout(irr)
g = torch.Generator(device="cpu").manual_seed(1)
pos = torch.zeros(2, 3, device=dev)
ei = torch.stack([torch.zeros(E, dtype=torch.long), torch.ones(E, dtype=torch.long)]).to(dev)
shift = (torch.randn(E, 3, generator=g) * 4).to(dev)
geo = ops.Geometry(pos, ei, shift, 26.0, 64, 6, torch.from_numpy(P.wigner_jtab(6)).to(dev))
xs, xd, fe = (torch.randn(E, lay.dim, generator=g).to(dev) for _ in range(3))
if a.same_rows:
    k = a.same_rows
    idx = (torch.arange(E, device=dev) % k)
    if k == 1:
        xs, xd, fe = (t[:1].expand(E, lay.dim) for t in (xs, xd, fe))
    else:   # rows repeat with period k: footprint k * 3 * Dp * 4 bytes
        xs, xd, fe = (t[:k].repeat((E + k - 1) // k, 1)[:E].contiguous() for t in (xs, xd, fe))
hn = ops.radial_hidden(geo.rbf, m._hn, 1.679); he = ops.radial_hidden(geo.rbf, m._he, 1.679) if m._he is not None else None
if a.nodes:
    node = torch.randn(a.nodes, lay.dim, generator=g).to(dev)
    geo.src = torch.randint(0, a.nodes, (E,), generator=g).to(dev)
    geo.dst = torch.randint(0, a.nodes, (E,), generator=g).to(dev)
    rot = torch.from_numpy(P.rotate_table(lay)).to(dev)
    launch = lambda: m.run_nodes(node, node, fe, geo, rot)
else:
    launch = lambda: ops.tp_fused(m._dp, [xs, xd, fe], E, hn, he, geo)
if a.adjoint:
    m.compile_adjoint(dev)
    gout = torch.randn(a.nodes if a.nodes else E, lay.dim, generator=g).to(dev)
    launch = (lambda: m.backward_data(gout, geo, True, gather=geo.dst)[0]) if a.nodes else (lambda: m.backward_data(gout, geo, True)[0])
if os.environ.get("HG_BENCH_LDS"):        # diagnostic: request this much LDS per workgroup (163840: ONE workgroup per CU instead of two)
    m._dp.sched.lds_floats = int(os.environ["HG_BENCH_LDS"]) // 4
for _ in range(2):
    out = launch()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    out = launch()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.reps
prog = m._dp_adj.prog if a.adjoint else m._dp.prog
print(json.dumps({"tag": a.tag, "kernel": "is" if m._dp.sched is not None else "seg", "lib": os.path.basename(os.environ.get("HG_LIB_PATH", "default")), "irreps": a.irreps, "E": E, "ms": dt * 1e3,
                  "issued_TF": (prog.mfma_per_wave - (prog.mfma_odd_skipped if (m._dp_adj if a.adjoint else m._dp).sched is not None else 0)) * 2048 / 16 * E / dt / 1e12, "useful_TF": prog.flops_per_row * E / dt / 1e12,
                  "Medges_s": E / dt / 1e6, "checksum": float(out.double().abs().mean())}))
if os.environ.get("HG_PROF") and m._dp.sched is not None:
    import ctypes as C
    from hamgnn_amd import _lib
    L = _lib.lib()
    buf = (C.c_ulonglong * 16)()
    L.hg_prof_is_read(buf, 1)
    launch()
    L.hg_prof_is_read(buf, 0)
    names = ["dispatch", "radial scale", "GEMM1", "scale-mul + GEMM2 + write-back", "zero fill", "phase barrier (imbalance)", "staging: wait + barrier", "epilogue",
             "staging: plain rows (DMA issue)", "staging: gathered l=0", "staging: rotated l=1..3", "staging: rotated l>=4"]
    tot = float(buf[15])
    print(json.dumps({"prof_total_wave_cycles": tot, "balance": m._dp.sched.balance, **{n: round(buf[k] / tot, 4) for k, n in enumerate(names)}}))
elif os.environ.get("HG_PROF"):
    import ctypes as C
    from hamgnn_amd import _lib
    L = _lib.lib()
    buf = (C.c_ulonglong * 16)()
    L.hg_prof_read(buf, 1)
    ops.tp_fused(m._dp, [xs, xd, fe], E, hn, he, geo)
    L.hg_prof_read(buf, 0)
    names = ["dispatch", "span issue + scale", "span wait", "GEMM1", "scale-mul + a2/cf loads", "GEMM2 + write-back", "linear write-back", "tile zero", "epilogue"]
    tot = float(buf[15])
    print(json.dumps({"prof_total_wave_cycles": tot, **{n: round(buf[k] / tot, 4) for k, n in enumerate(names)}}))

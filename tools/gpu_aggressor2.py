"""Which kernels are disturbed by a co-running kernel that only issues v_mfma_f32_16x16x32_f16 (tests/csrc/xdl_aggressor.hip)?  Victims: one launch each of the shipped library's
edge kernel (default / lite_mode / the older segment-stationary kernel), its row program / linear kernels through a ResidualBlock, its weight-gradient kernel path is left out;
and the vendor library's fp32 and bf16 GEMMs through torch.  profiles/r06_tp_is.md section 8.     python tools/gpu_aggressor2.py /tmp/libxdl_aggressor.so"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hamgnn_amd import nn as hnn, ops, plan as P
ap = argparse.ArgumentParser()
ap.add_argument("lib"); ap.add_argument("--launches", type=int, default=5); ap.add_argument("--grid", type=int, default=256); ap.add_argument("--modes", default="3,0,1")
a = ap.parse_args()
AG = ctypes.CDLL(a.lib)
AG.aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda")
irr, sh = "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e", "0e+1o+2e+3o+4e+5o"
E, nodes = 131072, 16384
lay = P.PlanarLayout(irr)
g = torch.Generator(device="cpu").manual_seed(1)
pos = torch.zeros(2, 3, device=dev)
ei = torch.stack([torch.zeros(E, dtype=torch.long), torch.ones(E, dtype=torch.long)]).to(dev)
shift = (torch.randn(E, 3, generator=g) * 4).to(dev)
geo = ops.Geometry(pos, ei, shift, 26.0, 64, 6, torch.from_numpy(P.wigner_jtab(6)).to(dev))
fe = torch.randn(E, lay.dim, generator=g).to(dev)
node = torch.randn(nodes, lay.dim, generator=g).to(dev)
geo.src = torch.randint(0, nodes, (E,), generator=g).to(dev)
geo.dst = torch.randint(0, nodes, (E,), generator=g).to(dev)
rot = torch.from_numpy(P.rotate_table(lay)).to(dev)
xs, xd = (torch.randn(E, lay.dim, generator=g).to(dev) for _ in range(2))
victims = {}
def mp(lite=False, seg=False):
    torch.manual_seed(0)
    if seg:
        os.environ["HG_MP_KERNEL"] = "seg"
    m = hnn.MessagePackBlock(irr, irr, sh, irr, 64, [64, 64], lite_mode=lite)
    m.compile(dev, unrotate=True)
    os.environ.pop("HG_MP_KERNEL", None)
    return m
m0 = mp(); victims["edge kernel tp_is (default, node-fed)"] = lambda: m0.run_nodes(node, node, fe, geo, rot)
hn0 = ops.radial_hidden(geo.rbf, m0._hn, 1.679); he0 = ops.radial_hidden(geo.rbf, m0._he, 1.679)
victims["edge kernel tp_is (pre-rotated rows: no gather / rotation in the staging)"] = lambda: ops.tp_fused(m0._dp, [xs, xd, fe], E, hn0, he0, geo)
m1 = mp(lite=True); victims["edge kernel tp_is (lite_mode instantiation)"] = lambda: m1.run_nodes(node, node, fe, geo, rot)
try:
    m2 = mp(seg=True)
    hn2 = ops.radial_hidden(geo.rbf, m2._hn, 1.679); he2 = ops.radial_hidden(geo.rbf, m2._he, 1.679)
    E2 = 32768
    victims["segment-stationary kernel tp_fused (r1-r2)"] = lambda: ops.tp_fused(m2._dp, [xs[:E2].contiguous(), xd[:E2].contiguous(), fe[:E2].contiguous()], E2, hn2[:E2].contiguous(), he2[:E2].contiguous(), geo)
except Exception as ex:
    print("seg victim not built:", str(ex)[:200])
torch.manual_seed(0)
rb = hnn.ResidualBlock(irr, irr); rb.compile(dev)
xn = torch.randn(65536, lay.dim, generator=g).to(dev)
victims["ResidualBlock (row program: linear + gate + linear)"] = lambda: rb(xn)
A32 = torch.randn(4096, 4096, generator=g).to(dev); B32 = torch.randn(4096, 4096, generator=g).to(dev)
victims["vendor fp32 GEMM 4096^3 (torch.mm)"] = lambda: A32 @ B32
A16, B16 = A32.bfloat16(), B32.bfloat16()
victims["vendor bf16 GEMM 4096^3 (torch.mm)"] = lambda: A16 @ B16
victims["elementwise (torch: x * 1.5 + y)"] = lambda: A32 * 1.5 + B32
side = torch.cuda.Stream()
names = {0: "dependent chains of v_mfma_f32_16x16x32_f16", 1: "independent v_mfma_f32_16x16x32_f16", 3: "dependent chains of v_mfma_f32_16x16x4_f32 (control)"}
for vname, launch in victims.items():
    ref = launch().clone(); torch.cuda.synchronize()
    same = all(torch.equal(launch(), ref) for _ in range(3))
    t0 = time.perf_counter(); launch(); torch.cuda.synchronize(); t_alone = (time.perf_counter() - t0) * 1e3
    for mode in [int(x) for x in a.modes.split(",")]:
        iters = 20000
        torch.cuda.synchronize(); t0 = time.perf_counter(); AG.aggressor_launch(mode, a.grid, iters, ctypes.c_void_p(side.cuda_stream)); torch.cuda.synchronize(); t_ag = (time.perf_counter() - t0) * 1e3
        iters = max(1000, int(iters * (4 * t_alone + 30.0) / max(t_ag, 1e-3)))
        bad, worst = 0, 0.0
        for n in range(a.launches):
            torch.cuda.synchronize()
            assert AG.aggressor_launch(mode, a.grid, iters, ctypes.c_void_p(side.cuda_stream)) == 0
            time.sleep(0.003)
            out = launch()
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                worst = max(worst, float((out.float() - ref.float()).abs().max() / ref.float().abs().max()))
        print(json.dumps({"victim": vname, "deterministic_alone": same, "aggressor": names[mode], "launches": a.launches, "launches_that_differ": bad, "worst_rel": worst, "victim_ms_alone": round(t_alone, 2)}), flush=True)

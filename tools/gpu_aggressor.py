"""The UNMODIFIED shipped edge kernel (one node-fed MessagePackBlock launch, 131 072 edges) while a SEPARATE kernel that only issues MFMAs in registers runs on a side stream
(tests/csrc/xdl_aggressor.hip: no LDS, no memory traffic, <= 50 VGPRs -- its waves fit on the SIMDs next to the edge kernel's two).  profiles/r06_tp_is.md section 8.
    python tools/gpu_aggressor.py /tmp/libxdl_aggressor.so [--launches 6]"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hamgnn_amd import nn as hnn, ops, plan as P
ap = argparse.ArgumentParser()
ap.add_argument("lib"); ap.add_argument("--launches", type=int, default=6); ap.add_argument("--edges", type=int, default=131072); ap.add_argument("--grid", type=int, default=1024)
ap.add_argument("--modes", default="5,3,2,0,1,4")
a = ap.parse_args()
AG = ctypes.CDLL(a.lib)
AG.aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
irr, sh = "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e", "0e+1o+2e+3o+4e+5o"
torch.manual_seed(0)
m = hnn.MessagePackBlock(irr, irr, sh, irr, 64, [64, 64])
dev = torch.device("cuda")
m.compile(dev, unrotate=True)
E, nodes = a.edges, 16384
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
launch = lambda: m.run_nodes(node, node, fe, geo, rot)
ref = launch().clone()
torch.cuda.synchronize()
t0 = time.perf_counter(); launch(); torch.cuda.synchronize(); t_alone = (time.perf_counter() - t0) * 1e3
side = torch.cuda.Stream()
names = {0: "dependent chains of v_mfma_f32_16x16x32_f16", 1: "independent v_mfma_f32_16x16x32_f16", 2: "dependent chains of v_mfma_f32_16x16x16_f16", 3: "dependent chains of v_mfma_f32_16x16x4_f32 (control)",
         4: "dependent chains of v_mfma_f32_16x16x32_bf16", 5: "VALU only (control)"}
# iterations for roughly 60 ms of aggressor per launch of the victim: calibrate on mode 3
def run_aggr(mode, iters):
    rc = AG.aggressor_launch(mode, a.grid, iters, ctypes.c_void_p(side.cuda_stream))
    assert rc == 0, rc
for mode in [int(x) for x in a.modes.split(",")]:
    iters = 20000
    torch.cuda.synchronize(); t0 = time.perf_counter(); run_aggr(mode, iters); torch.cuda.synchronize(); t_ag = (time.perf_counter() - t0) * 1e3
    iters = max(1000, int(iters * 80.0 / max(t_ag, 1e-3)))       # ~80 ms alone
    wrong_tiles, bad_launches, t_v = 0, 0, []
    for n in range(a.launches):
        torch.cuda.synchronize()
        run_aggr(mode, iters)
        time.sleep(0.005)                                       # the aggressor is resident before the victim's workgroups arrive
        t0 = time.perf_counter()
        out = launch()
        torch.cuda.current_stream().synchronize()
        t_v.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        d = (out - ref).abs().amax(1)
        nb = int((d.view(-1, 16).amax(1) > 0).sum())
        wrong_tiles += nb
        bad_launches += nb > 0
    print(json.dumps({"aggressor": names[mode], "aggressor_workgroups": a.grid, "victim": "shipped library, unmodified", "launches": a.launches, "launches_with_wrong_tiles": bad_launches,
                      "wrong_tiles": wrong_tiles, "of_tiles": a.launches * (E // 16), "victim_ms_alone": round(t_alone, 2), "victim_ms_with_aggressor": round(sum(t_v) / len(t_v), 2)}), flush=True)

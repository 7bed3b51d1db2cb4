"""The C host example end to end on the GPU box: this script (Python host) builds a set-A PairInteraction-type MessagePackBlock program, runs it through
ops.tp_fused, writes tables (hamgnn_amd/export.py), inputs and its own result to files; then examples/run_tp_is (C, no Python / torch / planner) loads
them, launches hg_tp_is and compares.  Three launch shapes: single part, split by output segment, segments shared by several workgroups (atomic adds).
    python tools/gpu_c_host.py [outdir]"""
import os, struct, subprocess, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from hamgnn_amd import export as X, nn as hnn, ops, plan as P

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/c_host"
os.makedirs(out, exist_ok=True)
dev = torch.device("cuda:0")
irr = B.IRREPS["A"]
torch.manual_seed(0)
blk = hnn.MessagePackBlock(irr, irr, B.SH, irr, 64, [64, 64])
skip = np.random.default_rng(0).normal(size=sum(m * m for m, _, _ in P.Irreps(irr)))
prog = P.build_message_pack_program(hnn._np_sd(blk), irr, irr, B.SH, irr, False, skip)
dp = ops.DeviceProgram(prog, dev, schedule="is")
Dp = P.PlanarLayout(irr).dim
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "run_tp_is")
ok = True
for tag, E, replay in (("single_part", 8192, False), ("split_by_segment", 2048, False), ("shared_segments", 160, True)):
    g = torch.Generator(device="cpu").manual_seed(E)
    srcs = [torch.randn(E, Dp, generator=g).to(dev) for _ in range(3)]
    hn, he = (torch.randn(E, dp.hidden, generator=g).to(dev) * 0.3 for _ in range(2))
    ops.REPLAY_SPLIT = replay
    want = ops.tp_fused(dp, srcs, E, hn, he, None, tag="message_pack")
    hdr = X.export_tp_is(dp, f"{out}/{tag}.hgprog", E)
    ops.REPLAY_SPLIT = False
    torch.cuda.synchronize()
    with open(f"{out}/{tag}_inputs.bin", "wb") as f:
        f.write(struct.pack("<4q", E, 3, Dp, dp.hidden))
        for t in srcs + [hn, he]:
            f.write(t.cpu().numpy().astype("<f4").tobytes())
    want.cpu().numpy().astype("<f4").tofile(f"{out}/{tag}_expected.bin")
    r = subprocess.run([exe, f"{out}/{tag}.hgprog", f"{out}/{tag}_inputs.bin", f"{out}/{tag}_expected.bin"], capture_output=True, text=True)
    print(tag, "nparts", hdr["nparts"], "zero_fill_out", hdr["zero_fill_out"], "|", r.stdout.strip(), r.stderr.strip()[:200], "| exit", r.returncode, flush=True)
    ok &= r.returncode == 0
    for fn in (f"{tag}_inputs.bin", f"{tag}_expected.bin", f"{tag}.hgprog"):
        os.remove(f"{out}/{fn}")
print("C HOST OK" if ok else "C HOST FAILED")
sys.exit(0 if ok else 1)

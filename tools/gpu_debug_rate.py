import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
from hamgnn_amd import ops
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
tag = os.environ.get("TAG", "")
irr = bench.IRREPS["A"]
torch.manual_seed(666)
m = HamGNNConvE3(bench.make_cfg(irr)).cuda()
g = bench.make_graph("sio2_10k", 19).to("cuda")
if os.environ.get("ONE_WG_PER_CU") == "1":                 # diagnostic: 160 KB of LDS per workgroup = one workgroup (one wave per SIMD) per CU
    with torch.no_grad():
        m(g)
    for blk in list(m.convolutions) + list(m.pair_interactions):
        for dp in (blk.conv_tp._dp, getattr(blk.conv_tp, "_dp_z", None)):
            if dp is not None and dp.sched is not None:
                dp.is_tables(1)[0].lds_floats = 163840 // 4
def fwd():
    with torch.no_grad():
        rep = m(g)
        return rep["_edge_planar_rot"]
ops.S_SPLIT_OFF = os.environ.get("HG_S_SPLIT", "1") == "0"
outs = [fwd().clone() for _ in range(3)]
ref = torch.stack(outs).median(0).values
del outs
sc = float(ref.abs().max())
bad = []
N = int(os.environ.get("N", "40"))
for i in range(N):
    d = float((fwd() - ref).abs().max()) / sc
    if d > 0:
        bad.append(round(d, 7))
print(json.dumps({"tag": tag, "forwards": N, "bad": len(bad), "devs": bad[:8]}), flush=True)

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
def fwd(off):
    ops.S_SPLIT_OFF = off
    with torch.no_grad():
        rep = m(g)
        return rep["_edge_planar_rot"].clone()
ref_on = fwd(False); ref_on2 = fwd(False)
flush = torch.empty(1 << 28, device="cuda")          # 1 GiB: evicts L2 / Infinity Cache when written
res = {"warm_pair_equal": bool(torch.equal(ref_on, ref_on2)), "after_off": [], "after_flush": [], "back_to_back": []}
sc = float(ref_on.abs().max())
for i in range(6):
    fwd(True)
    res["after_off"].append(float((fwd(False) - ref_on).abs().max()) / sc)
for i in range(6):
    flush.fill_(float(i))
    res["after_flush"].append(float((fwd(False) - ref_on).abs().max()) / sc)
for i in range(6):
    res["back_to_back"].append(float((fwd(False) - ref_on).abs().max()) / sc)
print(json.dumps({"tag": tag, **res}), flush=True)

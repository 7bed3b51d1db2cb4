import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
from hamgnn_amd import ops
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
tag = os.environ.get("TAG", "")
for wl, which in (("si512", "B"), ("sio2_300", "A"), ("mos2_48", "A"), ("si64", "A")):
    irr = bench.IRREPS[which]
    torch.manual_seed(666)
    m = HamGNNConvE3(bench.make_cfg(irr)).cuda()
    g = bench.make_graph(wl, 19).to("cuda")
    outs = []
    with torch.no_grad():
        for _ in range(8):
            rep = m(g)
            outs.append((rep["_node_planar"].clone(), rep["_edge_planar_rot"].clone()))
    torch.cuda.synchronize()
    sc = [float(t.abs().max()) for t in outs[0]]
    print(json.dumps({"tag": tag, "workload": wl, "irreps": which, "repeat_dev": [max(float((r[i] - outs[0][i]).abs().max()) / sc[i] for i in (0, 1)) for r in outs[1:]]}), flush=True)

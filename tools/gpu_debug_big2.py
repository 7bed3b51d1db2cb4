import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
from hamgnn_amd import ops
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
irr = bench.IRREPS["A"]
torch.manual_seed(666)
m = HamGNNConvE3(bench.make_cfg(irr)).cuda()
g = bench.make_graph("sio2_10k", 19).to("cuda")
for off in (True, False, True, False):
    ops.S_SPLIT_OFF = off
    outs = []
    with torch.no_grad():
        for _ in range(4):
            rep = m(g)
            outs.append((rep["_node_planar"].clone(), rep["_edge_planar_rot"].clone()))
            del rep
    torch.cuda.synchronize()
    sc = [float(t.abs().max()) for t in outs[0]]
    print(json.dumps({"s_split_off": off, "node_dev": [float((r[0] - outs[0][0]).abs().max()) / sc[0] for r in outs[1:]], "edge_dev": [float((r[1] - outs[0][1]).abs().max()) / sc[1] for r in outs[1:]],
                      "node_rows_differing": [int(((r[0] - outs[0][0]).abs().max(1).values > 0).sum()) for r in outs[1:]]}), flush=True)
    del outs

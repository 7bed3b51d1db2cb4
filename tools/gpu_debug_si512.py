import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
from hamgnn_amd import ops
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
for wl, which in (("si512", "B"), ("si512", "A"), ("mos2_1200", "A"), ("sio2_300", "A")):
    irr = bench.IRREPS[which]
    torch.manual_seed(666)
    m = HamGNNConvE3(bench.make_cfg(irr)).cuda()
    g = bench.make_graph(wl, 19).to("cuda")
    res = {}
    for off in (True, False):
        ops.S_SPLIT_OFF = off
        outs = []
        with torch.no_grad():
            for _ in range(6):
                rep = m(g)
                outs.append((rep["_node_planar"].clone(), rep["_edge_planar_rot"].clone()))
        torch.cuda.synchronize()
        res[off] = outs
    on, offr = res[False], res[True]
    sc = [float(t.abs().max()) for t in offr[0]]
    print(json.dumps({"workload": wl, "irreps": which, "E": int(g.num_edges),
                      "off_repeat": max(float((r[i] - offr[0][i]).abs().max()) for r in offr[1:] for i in (0, 1)),
                      "on_repeat": [max(float((r[i] - on[0][i]).abs().max()) / sc[i] for i in (0, 1)) for r in on[1:]],
                      "on_vs_off": [float((on[0][i] - offr[0][i]).abs().max()) / sc[i] for i in (0, 1)],
                      "parts": sorted({str(b.conv_tp._dp_for(int(g.num_edges), True).is_parts_for(int(g.num_edges))) for b in m.convolutions})}), flush=True)

import os, sys, json, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, bench, numpy as np
from hamgnn_amd import ops, _lib
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
irr = bench.IRREPS["A"]
torch.manual_seed(666)
m = HamGNNConvE3(bench.make_cfg(irr)).cuda()
g = bench.make_graph("sio2_10k", 19).to("cuda")
L = _lib.lib()
buf = (C.c_float * (256 * 16))()
n = C.c_uint(0)
ref = None
for i in range(30):
    with torch.no_grad():
        o = m(g)["_edge_planar_rot"].clone()
    L.hg_dbg_read(buf, C.byref(n))
    if ref is None:
        ref = o
    dev = float((o - ref).abs().max() / ref.abs().max())
    if n.value or dev > 0:
        arr = np.frombuffer(buf, dtype=np.float32).reshape(256, 16)[:min(n.value, 256)].copy()
        print(json.dumps({"forward": i, "mismatches": int(n.value), "dev_vs_first": dev}))
        for row in arr[:10]:
            print("   ", json.dumps({"item": int(row[0]), "tile": int(row[1]), "wave": int(row[2]), "lane": int(row[3]), "S": float(row[4]), "R": float(row[5]), "S0c0": float(row[6]), "S1c1": float(row[7]),
                                     "rt": int(row[8]), "r": int(row[9]), "RTM": int(row[10]), "MM": int(row[11]), "mlp": int(row[12]), "hbr_cls": int(row[13])}))
        lanes = sorted({int(r[3]) for r in arr}); tiles = sorted({int(r[1]) for r in arr}); items = sorted({int(r[0]) for r in arr})
        print("    lanes", lanes[:70], "tiles", tiles[:10], "items", items[:10], "rts", sorted({(int(r[10]), int(r[8])) for r in arr}))
print("done")

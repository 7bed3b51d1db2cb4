import os, sys, json, ctypes as C
sys.path.insert(0, os.getcwd())
import torch, bench, numpy as np
from hamgnn_amd import ops, _lib
from hamgnn_amd.data import synthetic as S
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
irr = bench.IRREPS["A"]
os.environ["HG_IS_PARTS"] = os.environ.get("PARTS", "8")
torch.manual_seed(666)
m = HamGNNConvE3(bench.make_cfg(irr)).cuda()
g = S.add_random_targets(S.mos2_monolayer(4, 4), 19, seed=0).to("cuda")
orig = ops.tp_fused
rec = []
def spy(dp, srcs, rows, *a, **k):
    out = orig(dp, srcs, rows, *a, **k)
    rec.append((k.get("tag", "linear"), rows, dp, [s.clone() for s in srcs], a, k))
    return out
ops.tp_fused = spy
with torch.no_grad():
    m(g)
ops.tp_fused = orig
L = _lib.lib()
buf = (C.c_float * (256 * 16))()
n = C.c_uint(0)
L.hg_dbg_read(buf, C.byref(n))
print("during the forward:", n.value)
t, r, dp, srcs, a, k = rec[3]
sc = dp.is_tables(dp.is_parts_for(r))[0]
ops.S_SPLIT_OFF = True
ref = orig(dp, srcs, r, *a, **k).clone()
ops.S_SPLIT_OFF = False
L.hg_dbg_read(buf, C.byref(n))
for rep in range(12):
    o_ = orig(dp, srcs, r, *a, **k)
    print(json.dumps({"rep": rep, "vs_off": float((o_ - ref).abs().max() / ref.abs().max())}))
    L.hg_dbg_read(buf, C.byref(n))
    arr = np.frombuffer(buf, dtype=np.float32).reshape(256, 16)[:min(n.value, 256)]
    print(json.dumps({"rep": rep, "mismatches": int(n.value)}))
    seen = set()
    for row in arr:
        key = (int(row[0]), int(row[1]), int(row[2]), int(row[3]))
        if key in seen:
            continue
        seen.add(key)
        it = sc.item_table[int(row[0])]
        print("   ", json.dumps({"item": int(row[0]), "tile": int(row[1]), "part": int(row[2]), "wave": int(row[3]), "lane": int(row[4]), "S": float(row[5]), "S_ref": float(row[6]), "hbr_cls": int(row[7]), "mlp": int(row[8]),
                                 "rt": int(row[9]), "RTM": int(row[11]), "MM": int(row[12]), "typ": int(it[0]), "seg": int(it[19]), "it_mlp": int(it[10]), "w3_off": int(row[14]), "it12": int(it[12])}))
        if len(seen) > 12:
            break

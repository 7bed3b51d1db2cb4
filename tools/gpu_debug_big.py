import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
from hamgnn_amd import ops
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
irr = bench.IRREPS["A"]
torch.manual_seed(666)
m = HamGNNConvE3(bench.make_cfg(irr)).cuda()
g = bench.make_graph(os.environ.get("WL", "sio2_10k"), 19).to("cuda")
orig = ops.tp_fused
rec = []
def spy(dp, srcs, rows, *a, **k):
    out = orig(dp, srcs, rows, *a, **k)
    if k.get("tag") == "message_pack":
        rec.append((dp, [s for s in srcs], a, dict(k), rows))
    return out
ops.tp_fused = spy
with torch.no_grad():
    m(g)
ops.tp_fused = orig
torch.cuda.synchronize()
for i, (dp, srcs, a, k, rows) in enumerate(rec):
    first = orig(dp, srcs, rows, *a, **k).clone()
    sc = float(first.abs().max())
    stats = []
    for rep in range(int(os.environ.get("REPS", "6"))):
        o = orig(dp, srcs, rows, *a, **k)
        d = (o - first).abs()
        nrows = int((d.max(1).values > 0).sum())
        stats.append((nrows, float(d.max()) / sc))
    torch.cuda.synchronize()
    print(json.dumps({"call": i, "rows_out": int(first.shape[0]), "reduce": k.get("reduce") is not None, "items": int(dp.prog.item_table.shape[0]), "differing_rows_and_rel": stats}), flush=True)

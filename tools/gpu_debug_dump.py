import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench, numpy as np
from hamgnn_amd import ops
from hamgnn_amd.data import synthetic as S
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
irr = bench.IRREPS["A"]
os.environ["HG_IS_PARTS"] = "8"
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
t, r, dp, srcs, a, k = rec[3]
ops.S_SPLIT_OFF = True
ref = orig(dp, srcs, r, *a, **k).clone()
ops.S_SPLIT_OFF = False
outs = [orig(dp, srcs, r, *a, **k).clone() for _ in range(12)]
torch.cuda.synchronize()
# the "good" split result: the element-wise median of the replays
good = torch.stack(outs).median(0).values
print(json.dumps({"good_vs_off": float((good - ref).abs().max() / ref.abs().max())}))
segs = dp.prog.seg_table
for n, o in enumerate(outs):
    d = (o - good).abs()
    if float(d.max()) == 0:
        print(json.dumps({"replay": n, "equal_to_median": True}))
        continue
    rows_bad = torch.nonzero(d.max(1).values > 0).flatten()
    cols_bad = torch.nonzero(d.max(0).values > 0).flatten()
    tiles = sorted({int(x) // 16 for x in rows_bad})
    segs_bad = {}
    for sg in segs:
        lk, mul_k, out_off, out_mulp = int(sg[0]), int(sg[1]), int(sg[3]), int(sg[4])
        w = (2 * lk + 1) * out_mulp
        blk = d[:, out_off:out_off + w]
        if float(blk.max()) > 0:
            # which components (m) and channels differ
            bb = blk.reshape(-1, 2 * lk + 1, out_mulp)
            segs_bad[f"l{lk}x{mul_k}"] = {"max": float(blk.max()), "rel_to_seg": float(blk.max() / good[:, out_off:out_off + w].abs().max()), "n_elem": int((blk > 0).sum()),
                                          "components": [int(x) for x in torch.nonzero(bb.amax((0, 2)) > 0).flatten()], "n_channels": int((bb.amax((0, 1)) > 0).sum())}
    lanes = sorted({int(x) % 16 for x in rows_bad})
    print(json.dumps({"replay": n, "n_rows": int(rows_bad.numel()), "tiles": tiles[:30], "n_tiles": len(tiles), "slots_in_tile": lanes, "segments": segs_bad}), flush=True)

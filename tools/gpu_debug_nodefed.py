import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
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
    rec.append((k.get("tag", "linear"), rows, dp, [s.clone() for s in srcs], a, dict(k)))
    return out
ops.tp_fused = spy
with torch.no_grad():
    m(g)
ops.tp_fused = orig
t, r, dp, srcs, a, k = rec[3]
def trial(name, srcs_, k_):
    first = orig(dp, srcs_, r, *a, **k_).clone()
    nbad = 0
    for _ in range(30):
        nbad += float((orig(dp, srcs_, r, *a, **k_) - first).abs().max()) > 0
    torch.cuda.synchronize()
    print(json.dumps({"variant": name, "differing_of_30": nbad, "rot_mask": k_.get("rot_mask"), "gather": [x is not None for x in (k_.get("gather") or [])]}), flush=True)
trial("as launched", srcs, k)
trial("no rotation (rot_mask 0)", srcs, dict(k, rot_mask=0))
gat = k["gather"]
pre = [s if gi is None else s[gi].contiguous() for s, gi in zip(srcs, gat)]
trial("rows gathered by the host, rotation in the kernel", pre, dict(k, gather=[None] * len(gat)))
trial("rows gathered by the host, no rotation", pre, dict(k, gather=[None] * len(gat), rot_mask=0))

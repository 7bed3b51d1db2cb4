import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
from hamgnn_amd import ops
from hamgnn_amd.data import synthetic as S
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
tag = os.environ.get("TAG", "")
irr = bench.IRREPS["A"]
for parts in ("1", "8", "13"):
    os.environ["HG_IS_PARTS"] = parts
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
    for i, (t, r, dp, srcs, a, k) in enumerate(rec):
        if t != "message_pack" or i not in (2, 3):
            continue
        first = orig(dp, srcs, r, *a, **k).clone()
        nbad, worst = 0, 0.0
        for _ in range(40):
            o = orig(dp, srcs, r, *a, **k)
            d = float((o - first).abs().max())
            nbad += d > 0
            worst = max(worst, d)
        torch.cuda.synchronize()
        print(json.dumps({"tag": tag, "parts": parts, "call": i, "replays": 40, "differing": nbad, "worst_abs": worst, "scale": float(first.abs().max())}), flush=True)

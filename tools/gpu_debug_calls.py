import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
from hamgnn_amd import ops
from hamgnn_amd.data import synthetic as S
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
os.environ["HG_IS_PARTS"] = os.environ.get("PARTS", "8")
irr = bench.IRREPS["A"]
torch.manual_seed(666)
m = HamGNNConvE3(bench.make_cfg(irr)).cuda()
g = S.add_random_targets(S.mos2_monolayer(4, 4), 19, seed=0).to("cuda")
orig = ops.tp_fused
rec = []
def spy(dp, srcs, rows, *a, **k):
    out = orig(dp, srcs, rows, *a, **k)
    rec.append((k.get("tag", "linear"), rows, dp, out.clone(), [s.clone() for s in srcs], a, k))
    return out
import hamgnn_amd.nn as hnn
ops.tp_fused = spy
runs = {}
for name, off in (("off", True), ("on1", False), ("on2", False)):
    ops.S_SPLIT_OFF = off
    rec.clear()
    with torch.no_grad():
        m(g)
    torch.cuda.synchronize()
    runs[name] = [(t, r, o) for (t, r, dp, o, s, a, k) in rec]
    if name == "on1":
        keep = list(rec)
for i, ((t, r, a), (_, _, b), (_, _, c)) in enumerate(zip(runs["off"], runs["on1"], runs["on2"])):
    sc = float(a.abs().max())
    print(json.dumps({"call": i, "tag": t, "rows": r, "on1_vs_off": float((a - b).abs().max()) / sc, "on2_vs_on1": float((c - b).abs().max()) / sc, "nan": bool(torch.isnan(b).any())}), flush=True)
# replay the first bad message_pack call in isolation, several times, split on: is the launch itself non-deterministic given identical inputs?
ops.tp_fused = orig
ops.S_SPLIT_OFF = False
for i, (t, r, dp, o, srcs, a, k) in enumerate(keep):
    if t != "message_pack":
        continue
    outs = [orig(dp, srcs, r, *a, **k).clone() for _ in range(4)]
    torch.cuda.synchronize()
    ops.S_SPLIT_OFF = True
    ref = orig(dp, srcs, r, *a, **k).clone()
    ops.S_SPLIT_OFF = False
    sc = float(ref.abs().max())
    print(json.dumps({"replay_call": i, "parts": str(dp.is_parts_for(r)), "kind": "z" if dp is not getattr(dp, "_x", None) else "", "vs_off": [float((x - ref).abs().max()) / sc for x in outs],
                      "items": int(dp.prog.item_table.shape[0]), "h2n_ptr_mod": int(a[0].data_ptr() % 256) if a and a[0] is not None else None,
                      "h2n_shape": list(a[0].shape) if a and a[0] is not None else None, "h2n_stride": list(a[0].stride()) if a and a[0] is not None else None,
                      "h2e_stride": list(a[1].stride()) if len(a) > 1 and a[1] is not None else None}), flush=True)
# locality of the error of one bad launch: by output segment (column block) and by 16-edge tile
import numpy as np
t, r, dp, o, srcs, a, k = keep[3]
ops.S_SPLIT_OFF = True
ref = orig(dp, srcs, r, *a, **k).clone()
ops.S_SPLIT_OFF = False
sc_ = dp.is_tables(dp.is_parts_for(r))[0]
for rep in range(3):
    out = orig(dp, srcs, r, *a, **k).clone()
    torch.cuda.synchronize()
    d = (out - ref).abs()
    lay = dp.prog.out_layout
    seg_err = {}
    for sg in dp.prog.seg_table:
        lk, mul_k, out_off, out_mulp = int(sg[0]), int(sg[1]), int(sg[3]), int(sg[4])
        w = (2 * lk + 1) * out_mulp
        seg_err[f"l{lk}x{mul_k}@{out_off}"] = float(d[:, out_off:out_off + w].max())
    tile_err = d.max(1).values.reshape(-1, 16).max(1).values
    bad_tiles = [int(i) for i in torch.nonzero(tile_err > 1e-4 * float(ref.abs().max())).flatten()[:40]]
    print(json.dumps({"rep": rep, "seg_err": {k_: round(v, 6) for k_, v in seg_err.items()}, "n_bad_tiles": int((tile_err > 1e-4 * float(ref.abs().max())).sum()), "bad_tiles": bad_tiles,
                      "copy_stride": [int(p[7]) for p in sc_.part_table], "part_segs": [[int(p[0]), int(p[1])] for p in sc_.part_table], "lds": int(sc_.lds_floats * 4)}), flush=True)

"""Which 16-edge tiles of ONE MessagePackBlock launch (node-fed, 131 072 edges = 8 192 workgroups, tile = blockIdx.x) come out different under a kernel-variant library
(profiles/r06_tp_is.md section 8)?   python tools/gpu_dummy_tiles.py save /tmp/o.pt;   HG_LIB_PATH=... python tools/gpu_dummy_tiles.py check /tmp/o.pt [--launches 6]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hamgnn_amd import nn as hnn, ops, plan as P
ap = argparse.ArgumentParser()
ap.add_argument("what", choices=["save", "check"]); ap.add_argument("path"); ap.add_argument("--launches", type=int, default=6); ap.add_argument("--edges", type=int, default=131072)
a = ap.parse_args()
irr, sh = "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e", "0e+1o+2e+3o+4e+5o"
torch.manual_seed(0)
m = hnn.MessagePackBlock(irr, irr, sh, irr, 64, [64, 64])
dev = torch.device("cuda")
m.compile(dev, unrotate=True)
E, nodes = a.edges, 16384
lay = P.PlanarLayout(irr)
g = torch.Generator(device="cpu").manual_seed(1)
pos = torch.zeros(2, 3, device=dev)
ei = torch.stack([torch.zeros(E, dtype=torch.long), torch.ones(E, dtype=torch.long)]).to(dev)
shift = (torch.randn(E, 3, generator=g) * 4).to(dev)
geo = ops.Geometry(pos, ei, shift, 26.0, 64, 6, torch.from_numpy(P.wigner_jtab(6)).to(dev))
fe = torch.randn(E, lay.dim, generator=g).to(dev)
node = torch.randn(nodes, lay.dim, generator=g).to(dev)
geo.src = torch.randint(0, nodes, (E,), generator=g).to(dev)
geo.dst = torch.randint(0, nodes, (E,), generator=g).to(dev)
rot = torch.from_numpy(P.rotate_table(lay)).to(dev)
launch = lambda: m.run_nodes(node, node, fe, geo, rot)
lib = os.path.basename(os.environ.get("HG_LIB_PATH", "default"))
if a.what == "save":
    ref = launch().clone()
    same = sum(int(torch.equal(launch(), ref)) for _ in range(5))
    torch.save(ref.cpu(), a.path)
    print(json.dumps({"library": lib, "saved": a.path, "replays_identical": f"{same} of 5", "rows": list(ref.shape)}))
    sys.exit(0)
ref = torch.load(a.path).to(dev)
segs = m._dp.prog.seg_table
tot_tiles, by_role, slots, seg_hits, n_bad_launch = 0, [0, 0], {}, {}, 0
examples = []
for n in range(a.launches):
    d = (launch() - ref).abs()
    rows_bad = torch.nonzero(d.amax(1) > 0).flatten()
    if rows_bad.numel() == 0:
        continue
    n_bad_launch += 1
    tiles = sorted({int(x) // 16 for x in rows_bad})
    tot_tiles += len(tiles)
    for t in tiles:
        by_role[(t >> 8) & 1] += 1
    for x in rows_bad:
        slots[int(x) % 16] = slots.get(int(x) % 16, 0) + 1
    for sg in segs:
        lk, mul_k, out_off, out_mulp = int(sg[0]), int(sg[1]), int(sg[3]), int(sg[4])
        w = (2 * lk + 1) * out_mulp
        blk = d[:, out_off:out_off + w]
        if float(blk.max()) > 0:
            key = f"l{lk}x{mul_k}@{out_off}"
            seg_hits[key] = seg_hits.get(key, 0) + int((blk.amax(1) > 0).sum())
    if len(examples) < 3:
        t0 = tiles[0]
        blk = d[16 * t0:16 * t0 + 16]
        examples.append({"tile": t0, "rows_bad_in_tile": int((blk.amax(1) > 0).sum()), "cols_bad": int((blk.amax(0) > 0).sum()), "max_abs": float(blk.max()), "ref_scale": float(ref[16 * t0:16 * t0 + 16].abs().max())})
print(json.dumps({"library": lib, "launches": a.launches, "launches_with_wrong_tiles": n_bad_launch, "wrong_tiles": tot_tiles, "wrong_tiles_by_bit8_of_workgroup": by_role,
                  "row_slot_histogram": dict(sorted(slots.items())), "rows_hit_per_segment": seg_hits, "examples": examples}))

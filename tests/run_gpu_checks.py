"""Run every GPU check independently (one failure does not hide the others) and dump JSON to gpurun_out/."""
import json, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tests import gpu_checks as G

checks = [("geometry", G.check_geometry, {}), ("message_pack_unrot", G.check_message_pack, {"unrotate": True}),
          ("message_pack_rot", G.check_message_pack, {"unrotate": False}), ("backbone", G.check_backbone, {}), ("backbone_lite", G.check_backbone, {"name": "backbone_lite"}),
          ("head19", G.check_head, {}), ("head_abacus13", G.check_head, {"name": "head_abacus_13", "ham_type": "abacus", "nao": 13}),
          ("head_soc_so3", G.check_head_soc, {}), ("random_cell", G.oracle_vs_hip_random, {}), ("batch_of_3", G.oracle_vs_hip_random, {"n_graphs": 3, "seed": 5}),
          ("si2_setA", G.check_default_irreps_si2, {"which": "A"}), ("si2_setB", G.check_default_irreps_si2, {"which": "B"}),
          ("message_pack_seg", G.check_message_pack, {"unrotate": True, "schedule": "seg"}), ("message_pack_is", G.check_message_pack, {"unrotate": True, "schedule": "is"}),
          ("head_soc_su2", G.check_head_su2, {}), ("zero_point_shift", G.check_zero_point_shift, {}),
          ("si512_full_size", G.check_full_size_properties, {"workload": "si512", "which": "B"}),
          ("mos2_1200_soc_full_size", G.check_full_size_properties, {"workload": "mos2_1200", "which": "A", "soc": True})]
out = {}
for name, fn, kw in checks:
    t = time.time()
    try:
        out[name] = fn(**kw)
    except Exception as e:
        out[name] = {"error": repr(e), "tb": traceback.format_exc()[-1500:]}
    out[name]["seconds"] = round(time.time() - t, 2)
    print(name, json.dumps(out[name])[:600], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gpu_checks.json"), "w"), indent=1)

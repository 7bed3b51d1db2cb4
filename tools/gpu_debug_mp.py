import os, sys, json
sys.path.insert(0, os.getcwd())
import torch, bench
from tests import gpu_checks as G
from hamgnn_amd import ops
tag = os.environ.get("TAG", "")
for split_off in (False, True):
    ops.S_SPLIT_OFF = split_off
    for parts in (1, 8, 13):
        errs = []
        for rep in range(3):
            r = G.check_message_pack_random(irr=bench.IRREPS["A"], sh=bench.SH, seed=7, E=int(os.environ.get("E", "592")), radial=(64, 64), parts=parts)
            errs.append(r["rel_err"])
        print(json.dumps({"tag": tag, "s_split_off": split_off, "parts": parts, "rel_err": errs}), flush=True)

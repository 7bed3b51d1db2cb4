import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from tests import gpu_checks as G
from hamgnn_amd import ops
for split_off in (False, True):
    ops.S_SPLIT_OFF = split_off
    for parts in (None, "1", "8", "13"):
        if parts is None:
            os.environ.pop("HG_IS_PARTS", None)
        else:
            os.environ["HG_IS_PARTS"] = parts
        for rep in range(2):
            r = G.check_default_irreps_si2("cuda", "A", "mos2_4", soc=True)
            print(json.dumps({"s_split_off": split_off, "parts": parts, "rep": rep, **{k: r[k] for k in ("node_rel_err", "edge_rel_err", "Hnet_rel_err")}}), flush=True)

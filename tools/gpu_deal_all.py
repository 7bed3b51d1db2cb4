"""Does the shipped edge kernel's result depend on how its waves line up?  (profiles/r06_tp_is.md section 7)
Single-part launches CLAIM their work groups (atomic counter), so the four waves of a workgroup drift apart; the variant library `-DIS_DEAL_ALL` deals the same group table
round-robin instead (group g0 + k NW + w to wave w): every wave starts an item at the same instant after each staging barrier -- the alignment under which the round's
half-precision experiment failed in every forward.  A segment's items stay on one wave in table order either way, so the two builds must agree BIT FOR BIT.
  python tools/gpu_deal_all.py save /tmp/h.pt          (default library)
  HG_LIB_PATH=hamgnn_amd/lib/variants/lib_dealall.so python tools/gpu_deal_all.py check /tmp/h.pt [--forwards 30]"""
import argparse, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
from hamgnn_amd.models.model import Model

ap = argparse.ArgumentParser()
ap.add_argument("what", choices=["save", "check"])
ap.add_argument("path")
ap.add_argument("--workload", default="sio2_10k")
ap.add_argument("--forwards", type=int, default=30)
a = ap.parse_args()
irr = B.IRREPS["A"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Model(HamGNNConvE3(B.make_cfg(irr)), HamGNNPlusPlusOut(irr, irr, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True,
                                                             soc_switch=False, calculate_sparsity=True, zero_point_shift=False)).to(dev)
g = B.make_graph(a.workload, 19).to(dev)
def fwd():
    with torch.no_grad():
        return model(g)["hamiltonian"]
if a.what == "save":
    h0 = fwd().clone()
    same = sum(int(torch.equal(fwd(), h0)) for _ in range(5))
    torch.save(h0.cpu(), a.path)
    print(json.dumps({"library": os.environ.get("HG_LIB_PATH", "default"), "saved": a.path, "replays_identical": f"{same} of 5", "rows": list(h0.shape)}))
else:
    ref = torch.load(a.path).to(dev)
    scale = float(ref.abs().max())
    bad, worst = 0, 0.0
    for _ in range(a.forwards):
        h = fwd()
        if not torch.equal(h, ref):
            bad += 1
            worst = max(worst, float((h - ref).abs().max()) / scale)
    print(json.dumps({"library": os.environ.get("HG_LIB_PATH", "default"), "forwards": a.forwards, "forwards_that_differ_from_the_default_library": bad, "worst_rel": worst}))

"""Si 2-atom cell forward (set-A, 3 layers, head): eager launches vs hipGraph replay (hamgnn_amd.graph_capture.CapturedForward), per HG_REPLAY_SPLIT mode of the process.
python tools/gpu_si2_replay.py [--workload si2] [--steps 300]"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from hamgnn_amd.graph_capture import CapturedForward
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
from hamgnn_amd.models.model import Model

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="si2")
ap.add_argument("--steps", type=int, default=300)
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
for _ in range(10):
    ref = fwd().clone()
torch.cuda.synchronize()
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps * 1e3
te = timed(fwd)
cap = CapturedForward(fwd)
out = cap()
torch.cuda.synchronize()
err = float((out - ref).abs().max() / ref.abs().max())
tr = timed(cap)
print(f"HG_REPLAY_SPLIT={os.environ.get('HG_REPLAY_SPLIT', '')} {a.workload}: eager {te:.3f} ms, graph replay {tr:.3f} ms per forward (replay vs eager rows {err:.1e}); E = {g.num_edges}")

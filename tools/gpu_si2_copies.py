"""where do the small device copies / fills of a 2-atom-cell forward come from?  torch.profiler with stacks on the GPU box"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
dev = torch.device("cuda")
irreps = bench.IRREPS["A"]
torch.manual_seed(666)
model = HamGNNConvE3(bench.make_cfg(irreps))
head = HamGNNPlusPlusOut(irreps, irreps, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True, soc_switch=False, calculate_sparsity=True)
g = bench.make_graph("si2", 19).to(dev)
def step():
    with torch.no_grad():
        return head(g, model(g))
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::_to_copy", "aten::contiguous", "aten::clone", "aten::fill_", "aten::zeros", "aten::cat", "aten::index", "aten::index_select", "aten::empty_like",
                   "aten::item", "aten::_local_scalar_dense", "aten::nonzero", "aten::sort", "aten::bincount", "aten::cumsum", "aten::arange"):
        st = [s for s in (ev.stack or []) if "hamgnn_amd" in s or "bench" in s]
        rows.append((ev.name, st[0] if st else "?"))
import collections
c = collections.Counter(rows)
for (name, where), n in c.most_common(60):
    print(n, name, where)
print("kernels launched in one forward:", sum(1 for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA))

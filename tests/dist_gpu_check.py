"""Sharded (multi-rank) HIP forward vs the single-rank forward, runnable on a ONE-GPU box:
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tests/dist_gpu_check.py
All ranks share cuda:0 and talk over gloo (RCCL refuses two ranks on one device); the product code path is the same
(parallel.shard_graph + parallel.allreduce_nodes inside HamGNNConvE3.forward)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
import bench
from hamgnn_amd import parallel
from hamgnn_amd.data import synthetic as S
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut

irreps = bench.IRREPS[os.environ.get("HG_DIST_IRREPS", "B")]
workload = os.environ.get("HG_DIST_WORKLOAD", "si64")
torch.manual_seed(666)
model = HamGNNConvE3(bench.make_cfg(irreps))
head = HamGNNPlusPlusOut(irreps, irreps, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True, soc_switch=False,
                         calculate_sparsity=False)
g = S.si_diamond(2, 2, 2, jitter=0.05, seed=0) if workload == "si64" else S.amorphous_sio2(int(workload.split("_")[1]), seed=1)
g = S.add_random_targets(g, 19, seed=0)
N, E = g.num_nodes, g.num_edges
sg = parallel.shard_graph(g, rank, world).to(dev)
with torch.no_grad():
    out = head(sg, model(sg))["hamiltonian"]
H_on, H_off = out[:N].cpu(), out[N:].cpu()
gathered = [None] * world
dist.all_gather_object(gathered, (sg["_hg_edge_ids"].cpu(), H_on, H_off))
if rank == 0:
    with torch.no_grad():
        gf = g.to(dev)
        ref = head(gf, model(gf))["hamiltonian"].cpu()
    off = torch.zeros(E, H_off.shape[1])
    for ids, on, of in gathered:
        off[ids] = of
        assert torch.allclose(on, gathered[0][1], atol=0, rtol=0) or (on - gathered[0][1]).abs().max() < 1e-5
    full = torch.cat([gathered[0][1], off], 0)
    err = ((full - ref).abs().max() / ref.abs().max()).item()
    assert err < 1e-5, err
    print("DIST_CHECK", json.dumps({"world": world, "workload": workload, "N": N, "E": E, "edges_per_rank": [int(x[0].numel()) for x in gathered], "rel_err": err}))
dist.destroy_process_group()

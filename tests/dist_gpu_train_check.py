"""The multi-rank TRAINING paths on the HIP kernels, runnable on a ONE-GPU box (all ranks share cuda:0 and talk over gloo; RCCL refuses
two ranks on one device):
    HG_DIST_MODE=train_conv | train_attn | dp  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 ... tests/dist_gpu_train_check.py
  train_conv / train_attn   model-parallel training step on an edge-sharded crystal (HamGNNConvE3 / HamGNNTransformer; the head with the
                            defaults of build_hamgnn_model: zero_point_shift + calculate_sparsity) == the single-process step on the whole
                            crystal: loss and every parameter gradient (sharded attention forward + backward included)
  dp                        data-parallel step (one crystal per rank, training.allreduce_gradients = the reference's DDP mean) == the mean of
                            the two single-process gradients"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
from hamgnn_amd import parallel, training as T
from hamgnn_amd.data import synthetic as S
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
from hamgnn_amd.models.model import Model
from tests.gpu_checks import MINI, SH

mode = os.environ.get("HG_DIST_MODE", "train_conv")
attn = mode == "train_attn"
irr = "8x0e+4x0o+4x1o+2x1e+2x2o+4x2e+2x3o" if attn else MINI
cfg = dict(num_types=20, irreps_edge_sh=SH, edge_sh_normalization="component", edge_sh_normalize=True, build_internal_graph=False,
           cutoff=26.0, rbf_func="bessel", num_radial=8, num_layers=2, irreps_node_features=irr, use_kan=False, radial_MLP=[16, 16],
           correlation=2, num_hidden_features=4, use_corr_prod=True)


def make():
    torch.manual_seed(9)
    if attn:
        from hamgnn_amd.models.hamgnn_transformer import HamGNNTransformer
        back = HamGNNTransformer(dict(cfg, num_heads=2))
    else:
        back = HamGNNConvE3(cfg)
    return Model(back, HamGNNPlusPlusOut(irr, irr, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True, soc_switch=False,
                                         calculate_sparsity=True, zero_point_shift=True)).to(dev)


def crystal(seed):
    g = S.add_random_targets(S.random_cell(6, [14, 8, 6, 1], seed=seed, density=0.004), 19, seed=seed)
    gen = torch.Generator().manual_seed(80 + seed)
    g["Son"] = torch.eye(19).reshape(1, -1).repeat(g.num_nodes, 1) + 0.01 * torch.randn(g.num_nodes, 361, generator=gen)
    g["Soff"] = 0.05 * torch.randn(g.num_edges, 361, generator=gen)
    return g


def single_process_step(g):
    keep = T.allreduce_gradients
    T.allreduce_gradients = lambda m, average=True: None       # a plain single-process step
    try:
        ref = make()
        r0 = T.training_step(ref, g.to(dev), metric="mae")
    finally:
        T.allreduce_gradients = keep
    return float(r0["loss"]), {k: p.grad.clone() for k, p in ref.named_parameters()}


worst = lambda a, b: max(float((a[k] - b[k]).abs().max()) / max(float(b[k].abs().max()), 1e-6) for k in b)
if mode == "dp":
    model = make()
    r = T.training_step(model, crystal(11 + rank).to(dev), metric="mae")            # allreduce_gradients inside: the mean over the ranks
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    if rank == 0:
        l0, g0 = single_process_step(crystal(11))
        l1, g1 = single_process_step(crystal(12))
        mean = {k: 0.5 * (g0[k] + g1[k]) for k in g0}
        res = {"mode": mode, "loss_err": abs(float(r["loss"]) - l0) / abs(l0), "grad_err": worst(grads, mean), "n": len(grads),
               "differs_from_rank0_alone": worst(grads, g0)}
        print("DIST_TRAIN", json.dumps(res))
else:
    g = crystal(11)
    model = make()
    r = T.training_step(model, parallel.shard_graph(g, rank, world).to(dev), metric="mae")
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    if rank == 0:
        l0, g0 = single_process_step(g)
        print("DIST_TRAIN", json.dumps({"mode": mode, "loss_err": abs(float(r["loss"]) - l0) / abs(l0), "grad_err": worst(grads, g0), "n": len(grads)}))
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()

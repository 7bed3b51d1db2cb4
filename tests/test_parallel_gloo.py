"""N>1 path on CPU: world_size-2 gloo.  The HIP kernels need a GPU, so the compute stand-in here is the oracle's
ConvBlockE3 message function; what is under test is the product's partition / local-inverse / all-reduce wiring."""
import json
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hamgnn_amd import parallel
from hamgnn_amd.data import synthetic as S


def test_partition_invariants():
    g = S.si_diamond(2, 2, 2, jitter=0.05, seed=3)
    E = g.num_edges
    for world in (2, 3, 8):
        owner = parallel.partition_pairs(g.edge_index, world)
        assert torch.equal(owner, owner[g.inv_edge_idx])                       # pairs co-located
        counts = torch.bincount(owner, minlength=world)
        assert counts.sum() == E and counts.max() < 1.6 * E / world           # load balance
        seen = torch.zeros(E, dtype=torch.long)
        for r in range(world):
            sg = parallel.shard_graph(g, r, world)
            ids = sg["_hg_edge_ids"]
            seen[ids] += 1
            assert torch.equal(sg.edge_index, g.edge_index[:, ids])
            assert torch.equal(ids[sg.inv_edge_idx], g.inv_edge_idx[ids])      # local inverse is the same pairing
            assert (sg.edge_index[0][1:] >= sg.edge_index[0][:-1]).all()      # still centre-major
        assert (seen == 1).all()


def test_partition_balance_on_the_benchmark_crystal():
    """BASELINE config #4 (a-SiO2, 10 002 atoms, 822 350 directed edges): the pair partition that `bench.py --gpus N` uses must be
    balanced to a few per mille -- the slowest rank bounds the strong-scaling efficiency (>= 6x at 8 GPUs needs <= 1.33x imbalance
    before any other overhead; measured 1.0006x at 8 ranks)."""
    g = S.amorphous_sio2(10002, seed=1)
    for world in (2, 4, 8):
        owner = parallel.partition_pairs(g.edge_index, world)
        assert torch.equal(owner, owner[g.inv_edge_idx])
        counts = torch.bincount(owner, minlength=world).double()
        assert counts.sum() == g.num_edges and (counts.max() / counts.mean()).item() < 1.01, (world, counts.tolist())
        # contiguous node blocks: a rank's pairs are keyed by a contiguous range of min(src, dst)
        key = torch.minimum(*g.edge_index)
        lo = torch.stack([key[owner == r].min() for r in range(world)])
        hi = torch.stack([key[owner == r].max() for r in range(world)])
        assert (lo[1:] > hi[:-1]).all()


def _worker_n(rank, world, port, tmp):
    """world ranks (some of them with very few or no edges) on a small random cell: sharded oracle backbone through the product's
    partition + local inverse + all-reduce hook == the unsharded run"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle import hamgnn_ref as R
    from tests.test_oracle_golden import load
    torch.set_default_dtype(torch.float64)
    f = load(os.path.join(os.path.dirname(__file__), "golden"), "backbone")
    cfg = json.loads(str(f["meta"]["cfg"]))
    m = R.HamGNNConvE3(cfg)
    m.load_state_dict(f["weights"], strict=False)
    g = S.random_cell(16, [14, 8], seed=4, density=0.004)
    g = type(g)({k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in g.items()})
    sg = parallel.shard_graph(g, rank, world)
    orig = R.scatter_sum
    R.scatter_sum = lambda src, index, dim_size: parallel.allreduce_nodes(orig(src, index, dim_size), sg)
    out = m(sg)
    R.scatter_sum = orig
    gathered = [None] * world
    dist.all_gather_object(gathered, (sg["_hg_edge_ids"], out["edge_attr"], out["node_attr"]))
    if rank == 0:
        ref = m(g)
        edge = torch.zeros_like(ref["edge_attr"])
        for ids, ea, na in gathered:
            edge[ids] = ea
            assert torch.allclose(na, gathered[0][2], atol=1e-12)
        err = max(((edge - ref["edge_attr"]).abs().max() / ref["edge_attr"].abs().max()).item(),
                  ((gathered[0][2] - ref["node_attr"]).abs().max() / ref["node_attr"].abs().max()).item())
        open(tmp, "w").write(json.dumps({"err": err, "edges_per_rank": [int(x[0].numel()) for x in gathered]}))
    dist.destroy_process_group()


def test_sharded_forward_eight_ranks_gloo(tmp_path):
    """the 8-rank layout of BASELINE config #4 (one all-reduce of the node aggregates per ConvBlock), on CPU"""
    port = 31500 + (os.getpid() % 2000)
    tmp = str(tmp_path / "err8.txt")
    mp.spawn(_worker_n, args=(8, port, tmp), nprocs=8, join=True)
    r = json.loads(open(tmp).read())
    assert r["err"] < 1e-10 and len(r["edges_per_rank"]) == 8 and min(r["edges_per_rank"]) > 0, r


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import hamgnn_ref as R
    from tests.test_oracle_golden import load
    torch.set_default_dtype(torch.float64)
    f = load(os.path.join(os.path.dirname(__file__), "golden"), "backbone")
    cfg = json.loads(str(f["meta"]["cfg"]))
    m = R.HamGNNConvE3(cfg)
    m.load_state_dict(f["weights"], strict=False)
    from hamgnn_amd.data import Graph
    g = Graph({k: v for k, v in f["graph"].items()})
    sg = parallel.shard_graph(g, rank, world)
    # sharded run of the oracle backbone: every scatter is followed by the product's all-reduce hook
    orig = R.scatter_sum
    R.scatter_sum = lambda src, index, dim_size: parallel.allreduce_nodes(orig(src, index, dim_size), sg)
    out = m(sg)
    R.scatter_sum = orig
    gathered = [None] * world
    dist.all_gather_object(gathered, (sg["_hg_edge_ids"], out["edge_attr"], out["node_attr"]))
    if rank == 0:
        E = g.edge_index.shape[1]
        edge = torch.zeros(E, out["edge_attr"].shape[1])
        for ids, ea, na in gathered:
            edge[ids] = ea
            assert torch.allclose(na, gathered[0][2], atol=1e-12)              # node features replicated
        ref_e, ref_n = f["outputs"]["edge_attr"], f["outputs"]["node_attr"]
        err = max(((edge - ref_e).abs().max() / ref_e.abs().max()).item(), ((gathered[0][2] - ref_n).abs().max() / ref_n.abs().max()).item())
        open(tmp, "w").write(str(err))
    dist.destroy_process_group()


def test_sharded_forward_matches_unsharded_gloo(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    tmp = str(tmp_path / "err.txt")
    mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
    assert float(open(tmp).read()) < 1e-10


def _worker_grads(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from hamgnn_amd.training import allreduce_gradients
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    params = list(m.parameters())
    for i, p in enumerate(params):
        p.grad = None if (rank == 1 and i == 2) else torch.full_like(p, float(rank + 1) * (i + 1))
    allreduce_gradients(m)
    want = [(i + 1) * 1.5 for i in range(len(params))]
    want[2] = 3 * 1.0 / 2                                       # rank 1 had no gradient for parameter 2: zeros
    ok = all(torch.allclose(p.grad, torch.full_like(p, w)) for p, w in zip(params, want))
    with open(os.path.join(tmp, f"g{rank}.json"), "w") as f:
        json.dump({"ok": bool(ok)}, f)
    dist.destroy_process_group()


def test_gradient_allreduce_two_ranks_gloo(tmp_path):
    """data-parallel training: the mean of the ranks' gradients in one flat bucket (hamgnn_amd.training.allreduce_gradients)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tmp = str(tmp_path)
    mp.spawn(_worker_grads, args=(2, port, tmp), nprocs=2, join=True)
    for r in range(2):
        assert json.load(open(os.path.join(tmp, f"g{r}.json")))["ok"]


def _worker_product(rank, world, port, tmp, transformer=False, node_shard=False):
    """the PRODUCT's own sharded forward (parallel.shard_graph + the all-reduce hook inside HamGNNConvE3.forward + the head on the local
    edges) on the CPU stand-ins of the kernels, vs the unsharded run"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if node_shard:
        os.environ["HG_NODE_SHARD"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from tests import cpu_ops
    from tests.gpu_checks import MINI, SH
    from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
    from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut

    class _MP:
        @staticmethod
        def setattr(o, n, v):
            setattr(o, n, v)
    cpu_ops.install(_MP)
    cfg = dict(num_types=20, irreps_edge_sh=SH, edge_sh_normalization="component", edge_sh_normalize=True, build_internal_graph=False,
               cutoff=26.0, rbf_func="bessel", num_radial=8, num_layers=2, irreps_node_features=MINI, use_kan=False, radial_MLP=[16, 16],
               correlation=2, num_hidden_features=4, use_corr_prod=bool(node_shard))
    torch.manual_seed(666)
    irr = MINI
    if transformer:                                             # attention backbone: the soft-max of a node spans the edges of both ranks
        from hamgnn_amd.models.hamgnn_transformer import HamGNNTransformer
        irr = "8x0e+4x0o+4x1o+2x1e+2x2o+4x2e+2x3o"
        model = HamGNNTransformer(dict(cfg, irreps_node_features=irr, num_heads=2))
    else:
        model = HamGNNConvE3(cfg)
    head = HamGNNPlusPlusOut(irr, irr, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True, soc_switch=False,
                             calculate_sparsity=False)
    g = S.add_random_targets(S.random_cell(5, [14, 8, 6, 1], seed=7, density=0.004), 19, seed=7)
    N, E = g.num_nodes, g.num_edges
    sg = parallel.shard_graph(g, rank, world)
    dead = model.declare_consumer(head)                         # (r5) what Model(...) does: the last pair block leaves out the irreps the head never reads
    assert dead, "the MINI irreps hold 0o, which an openmx nao 19 head never reads"
    with torch.no_grad():
        rep = model(sg)
        assert rep.get("_edge_alive") is not None
        out = head(sg, rep)["hamiltonian"]
    gathered = [None] * world
    dist.all_gather_object(gathered, (sg["_hg_edge_ids"], out[:N], out[N:]))
    if rank == 0:
        model.declare_consumer(object())                        # the unsharded reference runs the COMPLETE programs
        with torch.no_grad():
            ref = head(g, model(g))["hamiltonian"]
        off = torch.zeros(E, out.shape[1])
        for ids, on, of in gathered:
            off[ids] = of
        on_err = max(float((on - gathered[0][1]).abs().max()) for _, on, _ in gathered)
        err = float((torch.cat([gathered[0][1], off], 0) - ref).abs().max() / ref.abs().max())
        with open(tmp, "w") as f:
            json.dump({"err": err, "on_err": on_err, "edges_per_rank": [int(x[0].numel()) for x in gathered]}, f)
    dist.destroy_process_group()


def test_sharded_attention_backbone_gloo(tmp_path):
    """HamGNNTransformer on an edge-sharded crystal: per-node soft-max statistics merged across the ranks (max / sum all-reduces)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tmp = str(tmp_path / "att.json")
    mp.spawn(_worker_product, args=(2, port, tmp, True), nprocs=2, join=True)
    r = json.loads(open(tmp).read())
    assert r["err"] < 1e-5 and r["on_err"] < 1e-5 and min(r["edges_per_rank"]) > 0, r


def test_product_sharded_forward_on_cpu_stand_ins_gloo(tmp_path):
    """world_size 2: what bench.py --gpus N runs per rank (pair-sharded edges, one all-reduce of the node aggregates per ConvBlock, the
    head on the rank's edges), with the product's host code and the CPU stand-ins of the kernels (tests/cpu_ops.py)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tmp = str(tmp_path / "prod.json")
    mp.spawn(_worker_product, args=(2, port, tmp), nprocs=2, join=True)
    r = json.loads(open(tmp).read())
    assert r["err"] < 1e-5 and r["on_err"] < 1e-5 and len(r["edges_per_rank"]) == 2 and min(r["edges_per_rank"]) > 0, r


def test_product_sharded_forward_row_sharded_node_level_gloo(tmp_path):
    """HG_NODE_SHARD=1: the node-level chain of every ConvBlock (skip Linear, ResidualBlock, CorrProductBlock) on this rank's block of
    rows, reduce-scatter before / all-gather after (gloo: all-reduce + slice) -- three ranks on a 5-atom cell (ragged last block) == the unsharded run"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tmp = str(tmp_path / "prod_ns.json")
    mp.spawn(_worker_product, args=(3, port, tmp, False, True), nprocs=3, join=True)
    r = json.loads(open(tmp).read())
    assert r["err"] < 1e-5 and r["on_err"] < 1e-5 and len(r["edges_per_rank"]) == 3, r


def _worker_ddp(rank, world, port, tmp):
    """data-parallel training step: every rank its own crystal; training_step's gradient all-reduce == the mean of the ranks' local gradients"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from tests import cpu_ops
    from tests.gpu_checks import MINI, SH
    from hamgnn_amd import training as T
    from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
    from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
    from hamgnn_amd.models.model import Model

    class _MP:
        @staticmethod
        def setattr(o, n, v):
            setattr(o, n, v)
    cpu_ops.install(_MP)
    cfg = dict(num_types=20, irreps_edge_sh=SH, edge_sh_normalization="component", edge_sh_normalize=True, build_internal_graph=False,
               cutoff=26.0, rbf_func="bessel", num_radial=8, num_layers=1, irreps_node_features=MINI, use_kan=False, radial_MLP=[16, 16],
               correlation=2, num_hidden_features=4, use_corr_prod=False)
    torch.manual_seed(5)                                        # the same model on every rank
    model = Model(HamGNNConvE3(cfg), HamGNNPlusPlusOut(MINI, MINI, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True,
                                                       soc_switch=False, calculate_sparsity=False, zero_point_shift=False))
    g = S.add_random_targets(S.random_cell(3 + rank, [14, 8, 6, 1], seed=20 + rank, density=0.004), 19, seed=rank)      # a different crystal per rank
    keep = T.allreduce_gradients
    T.allreduce_gradients = lambda m, average=True: None         # local gradients first
    T.training_step(model, g, metric="mse")
    local = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    for p in model.parameters():
        p.grad = None
    T.allreduce_gradients = keep
    T.training_step(model, g, metric="mse")
    reduced = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    both = [None] * world
    dist.all_gather_object(both, (local, reduced))
    if rank == 0:
        mean = sum(b[0] for b in both) / world
        err = float((both[0][1] - mean).abs().max() / mean.abs().max())
        same = float((both[0][1] - both[1][1]).abs().max())
        with open(tmp, "w") as f:
            json.dump({"err": err, "same": same, "differs_from_local": float((both[0][0] - mean).abs().max() / mean.abs().max())}, f)
    dist.destroy_process_group()


def test_data_parallel_training_step_gloo(tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tmp = str(tmp_path / "ddp.json")
    mp.spawn(_worker_ddp, args=(2, port, tmp), nprocs=2, join=True)
    r = json.loads(open(tmp).read())
    assert r["err"] < 1e-6 and r["same"] == 0.0 and r["differs_from_local"] > 1e-3, r


def _worker_sharded_training(rank, world, port, tmp, transformer=False, defaults=False):
    """model-parallel training step on an edge-sharded crystal == the single-process step on the whole crystal (loss and every gradient)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from tests import cpu_ops
    from tests.gpu_checks import MINI, SH
    from hamgnn_amd import training as T
    from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
    from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
    from hamgnn_amd.models.model import Model

    class _MP:
        @staticmethod
        def setattr(o, n, v):
            setattr(o, n, v)
    cpu_ops.install(_MP)
    cfg = dict(num_types=20, irreps_edge_sh=SH, edge_sh_normalization="component", edge_sh_normalize=True, build_internal_graph=False,
               cutoff=26.0, rbf_func="bessel", num_radial=8, num_layers=2, irreps_node_features=MINI, use_kan=False, radial_MLP=[16, 16],
               correlation=2, num_hidden_features=4, use_corr_prod=True)

    irr = "8x0e+4x0o+4x1o+2x1e+2x2o+4x2e+2x3o" if transformer else MINI

    def make():
        torch.manual_seed(9)
        if transformer:
            from hamgnn_amd.models.hamgnn_transformer import HamGNNTransformer
            back = HamGNNTransformer(dict(cfg, irreps_node_features=irr, num_heads=2))
        else:
            back = HamGNNConvE3(cfg)
        # defaults: what hamgnn.main.build_hamgnn_model builds -- zero_point_shift and calculate_sparsity on: ONE dE and ONE ratio per crystal
        return Model(back, HamGNNPlusPlusOut(irr, irr, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True,
                                             soc_switch=False, calculate_sparsity=defaults, zero_point_shift=defaults))
    g = S.add_random_targets(S.random_cell(5, [14, 8, 6, 1], seed=11, density=0.004), 19, seed=11)
    if defaults:
        gen_s = torch.Generator().manual_seed(81)
        g["Son"] = torch.eye(19).reshape(1, -1).repeat(g.num_nodes, 1) + 0.01 * torch.randn(g.num_nodes, 361, generator=gen_s)
        g["Soff"] = 0.05 * torch.randn(g.num_edges, 361, generator=gen_s)
    model = make()
    r = T.training_step(model, parallel.shard_graph(g, rank, world), metric="mae")
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    if rank == 0:
        keep = T.allreduce_gradients
        T.allreduce_gradients = lambda m, average=True: None     # the reference run is a plain single-process step
        ref = make()
        r0 = T.training_step(ref, g, metric="mae")
        T.allreduce_gradients = keep
        worst = max(float((grads[k] - p.grad).abs().max()) / max(float(p.grad.abs().max()), 1e-6) for k, p in ref.named_parameters())
        with open(tmp, "w") as f:
            json.dump({"loss_err": abs(float(r["loss"]) - float(r0["loss"])) / abs(float(r0["loss"])), "grad_err": worst, "n": len(grads)}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_model_parallel_training_step_attention_backbone_gloo(tmp_path):
    """the same for HamGNNTransformer: the soft-max statistics, the per-node dot products of the soft-max backward, the key gradient and the
    learnable cutoff's gradient span the edges of both ranks"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tmp = str(tmp_path / "mpt.json")
    mp.spawn(_worker_sharded_training, args=(2, port, tmp, True), nprocs=2, join=True)
    r = json.loads(open(tmp).read())
    assert r["loss_err"] < 1e-6 and r["grad_err"] < 2e-5 and r["n"] > 120, r


def test_model_parallel_training_step_default_head_options_gloo(tmp_path):
    """zero_point_shift and calculate_sparsity as build_hamgnn_model switches them on: the shift's dE (numerator, denominator) and the
    adjoint's sum(g S) are summed over the ranks with the replicated on-site rows counted once, the sparsity ratio is the whole crystal's"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tmp = str(tmp_path / "mpd.json")
    mp.spawn(_worker_sharded_training, args=(2, port, tmp, False, True), nprocs=2, join=True)
    r = json.loads(open(tmp).read())
    assert r["loss_err"] < 1e-6 and r["grad_err"] < 1e-5 and r["n"] > 100, r


def test_model_parallel_training_step_gloo(tmp_path):
    """SURVEY 8e x 8f-3: pair-sharded edges in the BACKWARD: the partial node-level sums are all-reduced where the forward all-reduced the
    aggregates, the per-edge parameters' gradients are summed over the ranks, the loss counts the replicated on-site rows once"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    tmp = str(tmp_path / "mp.json")
    mp.spawn(_worker_sharded_training, args=(2, port, tmp), nprocs=2, join=True)
    r = json.loads(open(tmp).read())
    assert r["loss_err"] < 1e-6 and r["grad_err"] < 1e-5 and r["n"] > 100, r

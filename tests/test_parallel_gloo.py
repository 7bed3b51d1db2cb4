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

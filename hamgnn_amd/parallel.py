"""Multi-GPU path: one process per GPU, edges sharded by UNDIRECTED PAIR, node features replicated, one RCCL all-reduce of
the node aggregates per ConvBlockE3.  The reference has no counterpart (its only distribution is Lightning
DDP over graphs, hamgnn/main.py:318-321); pairs are kept together because off-site symmetrisation reads H[inv(e)]
(hamgnn/models/hamgnn_output.py:1273) and the PairInteractionBlock keeps per-edge state.

Message size per layer = N * Dp * 4 B (35 MB at 10k atoms, set-A): on 8 GPUs with 7 direct xGMI links a single all-reduce
is tens of microseconds of wire time against tens of milliseconds of edge compute, so one unbucketed all-reduce per layer
on the compute stream is the right granularity (nothing to overlap it with: the next kernel needs the reduced features)."""
from __future__ import annotations

import torch

from .data.graph import Graph

_EDGE_KEYS = ("nbr_shift", "cell_shift", "Hoff", "Hoff0", "Soff", "iHoff", "iHoff0", "Loff", "Hoff_nonsoc")


def partition_pairs(edge_index: torch.Tensor, world: int) -> torch.Tensor:
    """owner rank of every directed edge; both directions of a pair share the owner (keyed by min(src, dst)); contiguous
    node blocks balanced by edge count."""
    src, dst = edge_index
    key = torch.minimum(src, dst)
    n = int(max(int(src.max()), int(dst.max()))) + 1 if src.numel() else 0
    cnt = torch.bincount(key, minlength=n)
    csum = torch.cumsum(cnt, 0)
    total = int(csum[-1]) if n else 0
    bounds = torch.tensor([total * (r + 1) / world for r in range(world)], dtype=torch.float64)
    node_owner = torch.searchsorted(bounds, (csum - cnt).to(torch.float64) + 0.5 * cnt.to(torch.float64), right=False).clamp_(max=world - 1)
    return node_owner[key]


def shard_graph(g: Graph, rank: int, world: int) -> Graph:
    """Local graph of `rank`: all atoms (replicated), the rank's edges in the original centre-major order, local inverse map."""
    if world == 1:
        return g
    if "batch" in g and int(g["batch"].max()) > 0:
        raise ValueError("edge sharding works on a single crystal; multi-graph batches run as replicas (one batch per GPU)")
    owner = partition_pairs(g["edge_index"], world)
    sel = torch.nonzero(owner == rank).flatten()
    E = g["edge_index"].shape[1]
    newid = torch.full((E,), -1, dtype=torch.long)
    newid[sel] = torch.arange(sel.numel())
    inv_local = newid[g["inv_edge_idx"][sel]]
    assert (inv_local >= 0).all(), "pair split across ranks"
    # derived per-batch tensors (topology cache, the combined targets the head attaches on its first forward) are not carried over
    out = Graph({k: v for k, v in g.items() if k not in _EDGE_KEYS and k not in ("edge_index", "inv_edge_idx", "_hg_topology", "hamiltonian", "overlap", "hamiltonian_real", "hamiltonian_imag")})
    out["edge_index"] = g["edge_index"][:, sel].contiguous()
    out["inv_edge_idx"] = inv_local
    for k in _EDGE_KEYS:
        if k in g:
            out[k] = g[k][sel].contiguous()
    out["_hg_shard"] = (rank, world)
    out["_hg_edge_ids"] = sel
    out["_hg_inv_is_local_global"] = True
    return out


def is_sharded(data) -> bool:
    shard = data.get("_hg_shard") if hasattr(data, "get") else None
    return shard is not None and shard[1] > 1


def allreduce_nodes(t: torch.Tensor, data) -> torch.Tensor:
    """Sum the per-rank partial node aggregates in place (RCCL over xGMI: torch.distributed backend 'nccl')."""
    shard = data.get("_hg_shard") if hasattr(data, "get") else None
    if shard is None or shard[1] == 1:
        return t
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("graph is sharded but torch.distributed is not initialised")
    if PROFILE_EVENTS is not None:                             # bench.py: HIP event pairs around the collective (launch stream)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    if PROFILE_EVENTS is not None:
        ev1.record()
        PROFILE_EVENTS.append((ev0, ev1, int(t.numel()) * t.element_size()))
    return t


PROFILE_EVENTS = None        # bench.py sets this to a list: (start, end, bytes) around every node all-reduce / reduce-scatter / all-gather


# ---- row-sharded node-level work (off by default: HG_NODE_SHARD=1).  After the edge kernel every rank holds partial node aggregates
# [N, Dp]; the default sums them on every rank (all-reduce) and runs the node-level chain of the ConvBlock (skip Linear, ResidualBlock, CorrProductBlock:
# convolution.py:116-160) redundantly on all N rows.  With the flag each rank receives the SUM of ITS block of rows (reduce-scatter), runs the chain on
# N / world rows and the blocks are all-gathered: the same wire traffic as a ring all-reduce (2 (W - 1) / W of the tensor), the node-level compute
# divided by the world size.  At 10 k atoms that compute is ~1.5 ms of a ~37 ms step at 8 ranks: the flag exists so that the
# first multi-GPU run can A/B it, not because it is expected to matter.
def node_shard_enabled(data) -> bool:
    import os
    return is_sharded(data) and os.environ.get("HG_NODE_SHARD", "0") == "1"


def node_rows(data, N: int):
    """(first row, end row, rows per rank incl. padding) of this rank's block of the N node rows"""
    rank, world = data.get("_hg_shard")
    chunk = -(-N // world)
    return min(N, rank * chunk), min(N, (rank + 1) * chunk), chunk


def reduce_scatter_nodes(t: torch.Tensor, data) -> torch.Tensor:
    """sum of the ranks' partial aggregates, this rank's block of rows only -> [rows of the block, Dp]"""
    import torch.distributed as dist
    N, D = t.shape
    rank, world = data.get("_hg_shard")
    r0, r1, chunk = node_rows(data, N)
    ev = None
    if PROFILE_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if dist.get_backend() == "nccl":
        pad = t if N == chunk * world else torch.cat([t, t.new_zeros(chunk * world - N, D)], 0)
        out = t.new_empty(chunk, D)
        dist.reduce_scatter_tensor(out, pad.contiguous(), op=dist.ReduceOp.SUM)
        out = out[:r1 - r0]
    else:                                                      # gloo (the CPU / shared-device tests) has no reduce-scatter: all-reduce, keep the block
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out = t[r0:r1].contiguous()
    if ev is not None:
        ev[1].record()
        PROFILE_EVENTS.append((ev[0], ev[1], int(t.numel()) * t.element_size()))
    return out


def allgather_nodes(part: torch.Tensor, data, N: int) -> torch.Tensor:
    """the ranks' row blocks back to the full [N, Dp] node tensor on every rank"""
    import torch.distributed as dist
    rank, world = data.get("_hg_shard")
    r0, r1, chunk = node_rows(data, N)
    D = part.shape[1]
    buf = part if part.shape[0] == chunk else torch.cat([part, part.new_zeros(chunk - part.shape[0], D)], 0)
    out = part.new_empty(chunk * world, D)
    ev = None
    if PROFILE_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    dist.all_gather_into_tensor(out, buf.contiguous())
    if ev is not None:
        ev[1].record()
        PROFILE_EVENTS.append((ev[0], ev[1], int(out.numel()) * out.element_size()))
    return out[:N]


# ---- training on an edge-sharded crystal (model-parallel; hamgnn_amd.training): which parameter gradients are sums over the edges
def is_edge_summed(name: str) -> bool:
    """In the sharded backward every node-level tensor is replicated (each partial sum over a rank's edges is all-reduced before it is
    used), so the gradients of node-level parameters come out identical on all ranks; the gradients of parameters applied PER EDGE are
    sums over the rank's own edges and still have to be added up across the ranks: the tensor-product blocks (message blocks, pair
    embedding), the edge-row skip Linear of a PairInteractionBlock, the pair embedding's element tables, the off-site read-out networks."""
    return (".conv_tp." in name or name.startswith("conv_tp.") or ".conv_tp_value." in name or name.endswith(".linear_up_edge.weight")
            or (name.startswith("pair_interactions.") and ".skip_linear." in name) or name.startswith("pair_embedding.") or name.startswith("offsite_"))


def allreduce_edge_summed_gradients(named_grads, data):
    """SUM (not mean) over the ranks of the per-edge parameter gradients of a sharded training step, one flat bucket"""
    if not is_sharded(data):
        return
    import torch.distributed as dist
    keys = [k for k in named_grads if is_edge_summed(k)]
    if not keys:
        return
    flat = torch.cat([named_grads[k].reshape(-1).float() for k in keys])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    o = 0
    for k in keys:
        n = named_grads[k].numel()
        named_grads[k] = flat[o:o + n].reshape(named_grads[k].shape).to(named_grads[k].dtype)
        o += n

"""High-level builders: a reference module's state_dict -> Program (MessagePackBlock forward / data-gradient, the pair embedding, plain o3.Linear programs, lite_mode)."""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import so3
from ..so3 import Irreps
from ._blas import single_thread_blas
from .layout import IT_LINC, IT_LINM, IT_POST, PlanarLayout, SEG_UNROTATE, ceil_div, rtm_max, tp_instructions
from .program import Program, _add_item, _frag_A, add_linear_items, add_tp_adjoint_items, add_tp_items, new_program, seg_rows_cap, use_x4

# ------------------------------------------------------------------------------------------------ high-level builders

SRC_XS, SRC_XD, SRC_F = 0, 1, 2          # source slots of the fused kernel: rotated src-node rows, dst-node rows, edge rows


def _last_layer(sd, prefix):
    ks = sorted(k for k in sd if k.startswith(prefix + ".layer") and k.endswith(".weight"))
    return ks, np.asarray(sd[ks[-1]], dtype=np.float64)


@single_thread_blas
def choose_merge_groups(irreps_node, irreps_edge, irreps_sh, irreps_out, hidden: int, dead_out: Sequence[int] = ()) -> List[List[int]]:
    """Which small output irreps share their MFMA row tiles (add_tp_items merge_groups): per parity class (l + [p odd]) mod 2, the
    partition of the irreps with <= 16 channels that minimises the issued MFMAs of the block (exhaustive over the handful of
    candidates; cost = the planner's own count: radial scale + GEMM1 + GEMM2 per super-path, node and edge branch).  dead_out: output irreps the
    program does not compute (build_message_pack_program): never grouped."""
    irreps_node, irreps_edge, irreps_sh, irreps_out = Irreps(irreps_node), Irreps(irreps_edge), Irreps(irreps_sh), Irreps(irreps_out)
    dead_out = set(int(k) for k in dead_out)
    H4 = ceil_div(hidden, 16) * 4
    branches = []
    for nsrc, irr in ((2, irreps_node), (1, irreps_edge)):
        irr_in = Irreps([(m * nsrc, l, p) for m, l, p in irr])
        ins = tp_instructions(irr_in, irreps_sh, irreps_out)
        npath: Dict[Tuple[int, int], int] = {}
        for (i, j, k, slot) in ins:
            npath[(i, k)] = npath.get((i, k), 0) + 1
        branches.append((nsrc, PlanarLayout(irr), irr_in, npath))

    def cost(groups):
        tot = 0
        for G in groups:
            lmin = min(irreps_out[k][1] for k in G)
            rto = ceil_div(sum(irreps_out[k][0] for k in G), 16)
            for nsrc, lay, irr_in, npath in branches:
                for i, (mi2, li, pi) in enumerate(irr_in):
                    if li <= lmin and len(G) > 1:
                        stacks = [(sum(npath.get((i, k), 0) * irreps_out[k][0] for k in G), li, rto)]
                    else:
                        stacks = [(npath.get((i, k), 0) * irreps_out[k][0], min(li, irreps_out[k][1]), ceil_div(irreps_out[k][0], 16)) for k in G]
                    for nrows, mm, rt_o in stacks:
                        nc, ch = 2 * mm + 1, rtm_max(2 * mm + 1) * 16
                        for r0 in range(0, nrows, ch):
                            n = min(nrows, r0 + ch) - r0
                            rtm = ceil_div(n, 16)
                            tot += H4 * rtm + nsrc * (lay.mulp[i] // 4) * rtm * nc + rt_o * ceil_div(n, 4) * nc + 60
        return tot

    def partitions(xs):
        if not xs:
            yield []
            return
        for p in partitions(xs[1:]):
            yield [[xs[0]]] + p
            for n in range(len(p)):
                yield p[:n] + [[xs[0]] + p[n]] + p[n + 1:]

    out: List[List[int]] = []
    for cls in (0, 1):
        cand = [k for k, (m, l, p) in enumerate(irreps_out) if m <= 16 and (l + (p == -1)) % 2 == cls and m <= seg_rows_cap(l) and k not in dead_out]
        if len(cand) < 2 or len(cand) > 7:
            continue
        best = min((p for p in partitions(cand) if all(sum(irreps_out[k][0] for k in G) <= 64 for G in p)), key=cost)
        out += [sorted(G, key=lambda k: (irreps_out[k][1], k)) for G in best if len(G) > 1]
    return out


@single_thread_blas
def build_message_pack_program(sd: Dict[str, np.ndarray], irreps_node, irreps_edge, irreps_sh, irreps_out, unrotate: bool,
                               skip_weight: Optional[np.ndarray] = None, merge_groups: Sequence[Sequence[int]] = (),
                               zero_node: Sequence[int] = (), zero_edge: Sequence[int] = (), dead_out: Sequence[int] = ()) -> Program:
    """MessagePackBlock (non-lite, message_passing.py:216-229) [+ the PairInteractionBlock skip o3.Linear on the edge
    features, interaction_blocks.py:151-152] as ONE fused-kernel program.  `sd`: reference-named arrays of the block.
    zero_node / zero_edge (r5): irreps of the node / edge feature rows that are STRUCTURALLY zero where this block runs -- the first layer reads node rows
    that come out of an o3.Linear from `num_types x 0e` (only 0e blocks can be non-zero: _atomwise.py:55-57) and edge rows that come out of the pair embedding's
    0e (x) Y^l product (only the irreps of the spherical harmonics: embeddings.py:310-337).  The reference multiplies those zeros through every path
    (message_passing.py:216-229); here the super-paths (and skip-Linear paths) that read them are not emitted: same rows, bit for bit in exact arithmetic,
    because a dropped item would have added +0.0 to its tile cells.
    dead_out (r5): output irreps whose rows NOBODY reads where this block runs -- the edge rows of the last PairInteractionBlock feed only the read-out head,
    whose o3.Linear / Gate chain connects equal (l, p) only (hamgnn_output.py:38-58: it reads the irreps of the Hamiltonian blocks + the 0e gate scalars;
    e.g. 0o, 4o, 5o, 5e, 6e of the shipped set are never read for nao_max 19).  Their super-paths and skip-Linear paths are not emitted and their blocks of
    the output rows are written as ZEROS: a caller may use such a program only if it can hand the complete rows to anyone who asks later
    (HamGNNConvE3.declare_consumer keeps the inputs and re-runs the complete program on first access of the public `edge_attr`)."""
    irreps_node, irreps_edge, irreps_sh, irreps_out = Irreps(irreps_node), Irreps(irreps_edge), Irreps(irreps_sh), Irreps(irreps_out)
    _, w3n = _last_layer(sd, "node_weight_generator")
    _, w3e = _last_layer(sd, "edge_weight_generator")
    H = w3n.shape[0]
    assert H % 4 == 0 and w3e.shape[0] == H
    prog, seg_of_k = new_program(irreps_out, H, lambda k, ir: SEG_UNROTATE if unrotate else 0)
    add_tp_items(prog, seg_of_k, PlanarLayout(irreps_node), 2, [SRC_XS, SRC_XD], irreps_sh, irreps_out,
                 np.asarray(sd["node_tensor_product.weight"]), w3n / math.sqrt(H),
                 np.asarray(sd["node_linear_scaler.linear_out.weight"]), np.asarray(sd["node_linear_out.weight"]), mlp=0,
                 merge_groups=merge_groups, zero_inputs=zero_node, dead_out=dead_out)
    add_tp_items(prog, seg_of_k, PlanarLayout(irreps_edge), 1, [SRC_F], irreps_sh, irreps_out,
                 np.asarray(sd["edge_tensor_product.weight"]), w3e / math.sqrt(H),
                 np.asarray(sd["edge_linear_scaler.linear_out.weight"]), np.asarray(sd["edge_linear_out.weight"]), mlp=1,
                 merge_groups=merge_groups, zero_inputs=zero_edge, dead_out=dead_out)
    if skip_weight is not None:
        add_linear_items(prog, seg_of_k, PlanarLayout(irreps_edge), SRC_F, irreps_out, np.asarray(skip_weight), zero_inputs=zero_edge, dead_out=dead_out)
    return prog.finalize()


def message_pack_adjoint_layout(irreps_node, irreps_edge):
    """output irreps of the data-gradient program: the doubled node irreps (sender channels, then receiver channels, per irrep -- the
    reference's concatenated node-branch input, message_passing.py:207-214) followed by the edge irreps; + for every planar column of a
    node / edge feature row its column in that layout: (imap_src, imap_dst, imap_edge), each int32[Dp]."""
    irreps_node, irreps_edge = Irreps(irreps_node), Irreps(irreps_edge)
    adj = Irreps([(2 * m, l, p) for m, l, p in irreps_node] + [(m, l, p) for m, l, p in irreps_edge])
    lay, ln, le = PlanarLayout(adj), PlanarLayout(irreps_node), PlanarLayout(irreps_edge)
    imap_s, imap_d, imap_e = (np.full(ln.dim, -1, np.int32), np.full(ln.dim, -1, np.int32), np.full(le.dim, -1, np.int32))
    for i, (m, l, p) in enumerate(irreps_node):
        for a in range(2 * l + 1):
            o, oc = ln.off[i] + a * ln.mulp[i], lay.off[i] + a * lay.mulp[i]
            imap_s[o:o + m] = oc + np.arange(m)
            imap_d[o:o + m] = oc + m + np.arange(m)
    nb = len(irreps_node)
    for i, (m, l, p) in enumerate(irreps_edge):
        for a in range(2 * l + 1):
            o, oc = le.off[i] + a * le.mulp[i], lay.off[nb + i] + a * lay.mulp[nb + i]
            imap_e[o:o + m] = oc + np.arange(m)
    return adj, (imap_s, imap_d, imap_e)


@single_thread_blas
def build_message_pack_adjoint_program(sd: Dict[str, np.ndarray], irreps_node, irreps_edge, irreps_sh, irreps_out,
                                       zero_node: Sequence[int] = (), zero_edge: Sequence[int] = ()) -> Program:
    """DATA GRADIENT of a (non-lite) MessagePackBlock forward (message_passing.py:191-231) as a program for the same fused kernels:
    source slot 0 = the gradient with respect to the block's output rows [E, planar(irreps_out)] in the edge-aligned frame, output rows =
    [gradient of the doubled node-branch input | gradient of the edge-feature input] (message_pack_adjoint_layout), the node part
    un-rotated to the global frame in the epilogue (the adjoint of the rotation the forward applies while staging the gathered node
    rows), the edge part left in the edge frame (where the forward read it).  The radial hidden activations are those of the forward.
    Weight gradients are NOT part of this program.
    zero_node / zero_edge: structurally zero input irreps of the forward (build_message_pack_program) -- what produced those rows (the 0e chemical embedding,
    the 0e x Y^l pair embedding) has no path into them, so nobody reads their gradient: the items that compute it are dropped (zeros in those blocks)."""
    irreps_node, irreps_edge, irreps_sh, irreps_out = Irreps(irreps_node), Irreps(irreps_edge), Irreps(irreps_sh), Irreps(irreps_out)
    _, w3n = _last_layer(sd, "node_weight_generator")
    _, w3e = _last_layer(sd, "edge_weight_generator")
    H = w3n.shape[0]
    adj, _ = message_pack_adjoint_layout(irreps_node, irreps_edge)
    nb = len(irreps_node)
    prog, _ = new_program(adj, H, lambda k, ir: SEG_UNROTATE if k < nb else 0)
    gl = PlanarLayout(irreps_out)
    add_tp_adjoint_items(prog, PlanarLayout(irreps_node), 2, 0, gl, irreps_sh, irreps_out, np.asarray(sd["node_tensor_product.weight"]),
                         w3n / math.sqrt(H), np.asarray(sd["node_linear_scaler.linear_out.weight"]), np.asarray(sd["node_linear_out.weight"]),
                         mlp=0, target_base=0, skip_inputs=zero_node)
    add_tp_adjoint_items(prog, PlanarLayout(irreps_edge), 1, 0, gl, irreps_sh, irreps_out, np.asarray(sd["edge_tensor_product.weight"]),
                         w3e / math.sqrt(H), np.asarray(sd["edge_linear_scaler.linear_out.weight"]), np.asarray(sd["edge_linear_out.weight"]),
                         mlp=1, target_base=nb, skip_inputs=zero_edge)
    return prog.finalize()


@single_thread_blas
def build_embedding_adjoint_program(sd: Dict[str, np.ndarray], num_types: int, irreps_sh, irreps_out) -> Program:
    """data gradient of PairInteractionEmbeddingBlock.conv_tp with respect to its num_types x 0e input rows (what linear_up_src / linear_up_dst backpropagate),
    as a program for the fused kernels: source slot 0 = the gradient of the block's edge rows (edge frame), output = planar [E, num_types] (0e: frame-free)"""
    irreps_sh, irreps_out = Irreps(irreps_sh), Irreps(irreps_out)
    _, w3 = _last_layer(sd, "weight_generator")
    H = w3.shape[0]
    adj = Irreps([(num_types, 0, 1)])
    prog, _ = new_program(adj, H)
    add_tp_adjoint_items(prog, PlanarLayout(adj), 1, 0, PlanarLayout(irreps_out), irreps_sh, irreps_out, np.asarray(sd["tensor_product.weight"]),
                         w3 / math.sqrt(H), np.asarray(sd["linear_scaler.linear_out.weight"]), None, mlp=0, target_base=0)
    return prog.finalize()


@single_thread_blas
def build_embedding_program(sd: Dict[str, np.ndarray], num_types, irreps_sh, irreps_out, lite_mode=False) -> Program:
    """PairInteractionEmbeddingBlock.conv_tp (embeddings.py:328-334, tensor_products.py:170-189): source slot 0 holds
    x = Lin_src(onehot[src]) + Lin_dst(onehot[dst])  (num_types x 0e; identical in every frame)."""
    irreps_sh, irreps_out = Irreps(irreps_sh), Irreps(irreps_out)
    _, w3 = _last_layer(sd, "weight_generator")
    H = w3.shape[0]
    prog, seg_of_k = new_program(irreps_out, H)
    add_tp_items(prog, seg_of_k, PlanarLayout([(num_types, 0, 1)]), 1, [SRC_XS], irreps_sh, irreps_out,
                 None if lite_mode else np.asarray(sd["tensor_product.weight"]), w3 / math.sqrt(H),
                 np.asarray(sd["linear_scaler.linear_out.weight"]), None, mlp=0, uvu=lite_mode)
    return prog.finalize()


@single_thread_blas
def build_linear_program(weight: np.ndarray, irreps_in, irreps_out) -> Program:
    irreps_in, irreps_out = Irreps(irreps_in), Irreps(irreps_out)
    prog, seg_of_k = new_program(irreps_out, 0)
    add_linear_items(prog, seg_of_k, PlanarLayout(irreps_in), 0, irreps_out, np.asarray(weight))
    return prog.finalize()


def radial_hidden_weights(sd: Dict[str, np.ndarray], prefix: str, act_cst: float):
    """All but the last layer of an e3nn FullyConnectedNet, with 1/sqrt(h_in) folded in.  Returns [(W [h_in,h_out])...]."""
    ks, _ = _last_layer(sd, prefix)
    out = []
    for k in ks[:-1]:
        W = np.asarray(sd[k], dtype=np.float64)
        out.append((W / math.sqrt(W.shape[0])).astype(np.float32))
    return out


def add_lite_branch_items(prog: Program, seg_of_k, in_layout: PlanarLayout, nsrc, srcs, irreps_sh: Irreps, irreps_out: Irreps,
                          lin_w: np.ndarray, fold: bool = False):
    """lite_mode branch (message_passing.py:197-206): unweighted uvu tensor product followed by o3.Linear(mid.simplify()->out),
    i.e. per path p = (i, l_sh, k):  tile_k[w'', m] += coef_p[m] * sum_u (sqrt(2 l_k+1)/sqrt(fan_k) L_k[(p,u), w'']) x'_i[u, src_p(m)].
    fold (input-stationary kernel only): all paths of one (i, k) share the input block, the column map and its direction (the parity of
    l_i + l_sh + l_k is fixed by the parities of i and k), so they fold into ONE item with a weight matrix per column,
    A_m = sum_p coef_p[m] A_p (IT_LINM): 1 / (number of l_sh per pair) of the MFMAs and of the items."""
    if fold:
        return _add_lite_branch_items_folded(prog, seg_of_k, in_layout, nsrc, srcs, irreps_sh, irreps_out, lin_w)
    irr_in = Irreps([(m * nsrc, l, p) for m, l, p in in_layout.irreps])
    ins = tp_instructions(irr_in, irreps_sh, irreps_out)          # slot order = sorted by output irrep (stable), as the reference
    by_k: Dict[int, List[int]] = {}
    for n, (i, j, k, slot) in enumerate(ins):
        by_k.setdefault(k, []).append(n)
    order = sorted(by_k, key=lambda k: ((irreps_out[k][1], irreps_out[k][2]), k))
    off = 0
    for k in order:
        mk, lk, pk = irreps_out[k]
        fan = sum(irr_in[ins[n][0]][0] for n in by_k[k])
        Lk = lin_w[off:off + fan * mk].reshape(fan, mk).astype(np.float64) * (math.sqrt(2 * lk + 1) / math.sqrt(fan))
        off += fan * mk
        r = 0
        for n in by_k[k]:
            i, j, _, _ = ins[n]
            mi2, li, pi = irr_in[i]
            mi = mi2 // nsrc
            lj = irreps_sh[j][1]
            mm = min(li, lk)
            nc = 2 * mm + 1
            _, coef_c = so3.aligned_path(li, lj, lk)
            cf = np.array([coef_c[lk + m] for m in range(-mm, mm + 1)])
            Wp = Lk[r:r + mi2]                                   # [u (src channels then dst channels), w'']
            r += mi2
            ksteps = in_layout.mulp[i] // 4
            chunk = rtm_max(nc) * 16
            for seg, c0, c1 in prog.seg_chunks[k]:
                for r0 in range(c0, c1, chunk):
                    r1 = min(c1, r0 + chunk)
                    rtm = ceil_div(r1 - r0, 16)
                    a1 = [_frag_A(Wp[s_ * mi:(s_ + 1) * mi, r0:r1], ksteps, rtm, use_x4(in_layout.mulp[i], nc)) for s_ in range(nsrc)]
                    a1_off = prog.add_weights(np.stack(a1))
                    cf_off = prog.add_weights(cf)
                    _add_item(prog, seg, IT_LINC, list(srcs), in_layout.off[i], in_layout.mulp[i], li, mm, (li + lj + lk) % 2, ksteps, rtm, 0,
                              a1_off, 0, cf_off, 0, r1 - r0, row_off=r0 - c0)
            prog.flops_per_row += 2.0 * mi2 * mk * nc
    assert off == lin_w.size, (off, lin_w.size)


def _add_lite_branch_items_folded(prog: Program, seg_of_k, in_layout: PlanarLayout, nsrc, srcs, irreps_sh, irreps_out, lin_w):
    pairs: Dict[Tuple[int, int], dict] = {}
    for pth in lite_paths(in_layout, nsrc, irreps_sh, irreps_out, lin_w):
        q = pairs.setdefault((pth["i"], pth["k"]), dict(pth, Wc=np.zeros((2 * pth["mm"] + 1,) + pth["Wp"].shape)))
        assert (q["par"], q["mm"]) == (pth["par"], pth["mm"])
        q["Wc"] += pth["cf"][:, None, None] * pth["Wp"][None]
        prog.flops_per_row += 2.0 * pth["Wp"].shape[0] * pth["mk"] * (2 * pth["mm"] + 1)
    for (i, k), q in pairs.items():
        mi, mm, li = q["mi"], q["mm"], q["li"]
        nc = 2 * mm + 1
        ksteps = in_layout.mulp[i] // 4
        chunk = rtm_max(nc) * 16
        for seg, c0, c1 in prog.seg_chunks[k]:
            for r0 in range(c0, c1, chunk):
                r1 = min(c1, r0 + chunk)
                rtm = ceil_div(r1 - r0, 16)
                frags = np.stack([np.stack([_frag_A(q["Wc"][c, s_ * mi:(s_ + 1) * mi, r0:r1], ksteps, rtm, False) for s_ in range(nsrc)]) for c in range(nc)])
                a1_off = prog.add_weights(frags)                # [column][source][G][rt][64][4]
                _add_item(prog, seg, IT_LINM, list(srcs), in_layout.off[i], in_layout.mulp[i], li, mm, q["par"], ksteps, rtm, 0,
                          a1_off, 0, int(frags[0].size), 0, r1 - r0, row_off=r0 - c0)
                prog.seg_items[seg][-1][17] = 0                 # natural-K operands


def lite_paths(in_layout: PlanarLayout, nsrc, irreps_sh, irreps_out, lin_w: np.ndarray):
    """the paths of one lite_mode branch with their folded Linear blocks: yields (i, k, l_sh, Wp [nsrc mul_i, mul_k] (incl. sqrt(2 l_k + 1) /
    sqrt(fan)), offset of the block's first row in the flat _MidLinear weight, fan, cf [2 mm + 1], parity) -- see add_lite_branch_items"""
    irreps_sh, irreps_out = Irreps(irreps_sh), Irreps(irreps_out)
    irr_in = Irreps([(m * nsrc, l, p) for m, l, p in in_layout.irreps])
    ins = tp_instructions(irr_in, irreps_sh, irreps_out)
    by_k: Dict[int, List[int]] = {}
    for n, (i, j, k, slot) in enumerate(ins):
        by_k.setdefault(k, []).append(n)
    off = 0
    for k in sorted(by_k, key=lambda k: ((irreps_out[k][1], irreps_out[k][2]), k)):
        mk, lk, pk = irreps_out[k]
        fan = sum(irr_in[ins[n][0]][0] for n in by_k[k])
        scale = math.sqrt(2 * lk + 1) / math.sqrt(fan)
        Lk = lin_w[off:off + fan * mk].reshape(fan, mk).astype(np.float64) * scale
        r = 0
        for n in by_k[k]:
            i, j, _, _ = ins[n]
            mi2, li, pi = irr_in[i]
            lj = irreps_sh[j][1]
            mm = min(li, lk)
            _, coef_c = so3.aligned_path(li, lj, lk)
            cf = np.array([coef_c[lk + m] for m in range(-mm, mm + 1)])
            yield dict(i=i, k=k, lj=lj, Wp=Lk[r:r + mi2], w_off=off + r * mk, fan=fan, scale=scale, cf=cf, par=(li + lj + lk) % 2, mm=mm, li=li, lk=lk,
                       mi=mi2 // nsrc, mk=mk)
            r += mi2
        off += fan * mk
    assert off == lin_w.size, (off, lin_w.size)


@single_thread_blas
def build_message_pack_lite_adjoint_program(sd: Dict[str, np.ndarray], irreps_node, irreps_edge, irreps_sh, irreps_out) -> Program:
    """DATA GRADIENT of the item part of a lite_mode MessagePackBlock (the uvu products folded with the _MidLinears; the combine
    post-op's adjoint -- g_t = s * (Lc g_out) -- is applied to the gradient rows before, outside this program).  Source slot 0 = g_t
    [E, planar(irreps_out)] in the edge frame; output rows as message_pack_adjoint_layout; IT_LINC items with the roles of the two
    irreps exchanged: tile_i[u, l_i + s m] += cf[m] sum_w Wp[u, w] g_t[w, l_k + m]  (s = -1 for odd paths: the same `neg` flag, the
    coefficient vector reversed)."""
    irreps_node, irreps_edge, irreps_sh, irreps_out = Irreps(irreps_node), Irreps(irreps_edge), Irreps(irreps_sh), Irreps(irreps_out)
    adj, _ = message_pack_adjoint_layout(irreps_node, irreps_edge)
    nb = len(irreps_node)
    prog, _ = new_program(adj, 0, lambda k, ir: SEG_UNROTATE if k < nb else 0)
    gl = PlanarLayout(irreps_out)
    for lay, nsrc, key, base in ((PlanarLayout(irreps_node), 2, "node_linear_scaler.weight", 0), (PlanarLayout(irreps_edge), 1, "edge_linear_scaler.weight", nb)):
        for pth in lite_paths(lay, nsrc, irreps_sh, irreps_out, np.asarray(sd[key])):
            i, k, mm, lk, mk = pth["i"], pth["k"], pth["mm"], pth["lk"], pth["mk"]
            nc = 2 * mm + 1
            cf = pth["cf"][::-1] if pth["par"] else pth["cf"]
            ksteps = gl.mulp[k] // 4
            chunk = rtm_max(nc) * 16
            for seg, c0, c1 in prog.seg_chunks[base + i]:     # column chunks of the (nsrc * mul_i) target channels
                for r0 in range(c0, c1, chunk):
                    r1 = min(c1, r0 + chunk)
                    rtm = ceil_div(r1 - r0, 16)
                    a1_off = prog.add_weights(_frag_A(pth["Wp"][r0:r1].T, ksteps, rtm, use_x4(gl.mulp[k], nc))[None])
                    cf_off = prog.add_weights(cf)
                    _add_item(prog, seg, IT_LINC, [0], gl.off[k], gl.mulp[k], lk, mm, pth["par"], ksteps, rtm, 0, a1_off, 0, cf_off, 0, r1 - r0,
                              row_off=r0 - c0)
            prog.flops_per_row += 2.0 * pth["Wp"].shape[0] * mk * nc
    return prog.finalize()


@single_thread_blas
def build_message_pack_program_lite(sd: Dict[str, np.ndarray], irreps_node, irreps_edge, irreps_sh, irreps_out, unrotate: bool, post: bool = True,
                                    fold: bool = False) -> Program:
    """MessagePackBlock with lite_mode=True (message_passing.py:197-215) as one fused-kernel program.  post=False: without the combine
    post-op (the pre-combine rows t that the backward's reductions read).  fold: the paths of every (input irrep, output irrep) pair as one
    IT_LINM item (input-stationary kernel; the segment-stationary kernel runs the unfolded IT_LINC items)."""
    irreps_node, irreps_edge, irreps_sh, irreps_out = Irreps(irreps_node), Irreps(irreps_edge), Irreps(irreps_sh), Irreps(irreps_out)
    _, w3 = _last_layer(sd, "weight_generator_combine")
    H = w3.shape[0]
    prog, seg_of_k = new_program(irreps_out, H, lambda k, ir: SEG_UNROTATE if unrotate else 0)
    add_lite_branch_items(prog, seg_of_k, PlanarLayout(irreps_node), 2, [SRC_XS, SRC_XD], irreps_sh, irreps_out, np.asarray(sd["node_linear_scaler.weight"]), fold)
    add_lite_branch_items(prog, seg_of_k, PlanarLayout(irreps_edge), 1, [SRC_F], irreps_sh, irreps_out, np.asarray(sd["edge_linear_scaler.weight"]), fold)
    # post-op per segment: scale by the radial weights (one per channel of irreps_out.simplify()) and o3.Linear(out -> out)
    w3n = w3 / math.sqrt(H)
    lc = np.asarray(sd["combine_messages.linear_out.weight"])
    irs = [(l, p) for _, l, p in irreps_out]
    assert len(set(irs)) == len(irs)
    ch_off, lo_off, co, lo = {}, {}, 0, 0
    for k, (mk, lk, pk) in enumerate(irreps_out):
        ch_off[k], lo_off[k] = co, lo
        co += mk
        lo += mk * mk
    assert co == w3.shape[1] and lo == lc.size
    for k, (mk, lk, pk) in enumerate(irreps_out):
        if not post:
            break
        assert len(prog.seg_chunks[k]) == 1, "lite_mode post-op needs <= 64 channels per output irrep"
        seg = seg_of_k[k]
        rto = prog.segs[seg][2]
        Lk = lc[lo_off[k]:lo_off[k] + mk * mk].reshape(mk, mk).astype(np.float64) / math.sqrt(mk)
        w3_off = prog.add_weights(_frag_A(w3n[:, ch_off[k]:ch_off[k] + mk], prog.hidden_pad // 4, rto, True))
        Lp = np.zeros((rto * 16, rto * 16))
        Lp[:mk, :mk] = Lk
        a2 = Lp.reshape(rto, 4, 4, rto, 16).transpose(3, 0, 1, 4, 2).reshape(rto, rto, 64, 4)
        a2_off = prog.add_weights(a2)
        _add_item(prog, seg, IT_POST, [0], 0, 4, 0, 0, 0, 0, rto, 0, 0, w3_off, 0, a2_off, mk)
        prog.flops_per_row += 2.0 * H * mk + 2.0 * mk * mk * (2 * lk + 1)
    return prog.finalize()

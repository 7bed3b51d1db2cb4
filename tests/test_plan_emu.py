"""Planner + kernel algorithm (numpy emulation over the packed device buffers) against the golden fixtures."""
import math
import os

import numpy as np
import pytest

from hamgnn_amd import plan as P
from hamgnn_amd import so3
from tests import emu

MINI, SH = "8x0e+4x0o+4x1o+2x1e+2x2o+3x2e+2x3o", "0e+1o+2e+3o"


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        g, kk = k.split("/", 1)
        out.setdefault(g, {})[kk] = z[k]
    return out


def rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def test_planar_roundtrip():
    lay = P.PlanarLayout(MINI)
    x = np.random.default_rng(0).normal(size=(5, lay.irreps.dim))
    assert np.array_equal(lay.from_planar(lay.to_planar(x)), x)
    assert lay.dim % 4 == 0


def test_instruction_table_matches_reference_counts():
    A = so3.Irreps("64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e")
    A2 = so3.Irreps([(2 * m, l, p) for m, l, p in A])
    ins = P.tp_instructions(A2, so3.Irreps("0e+1o+2e+3o+4e+5o"), A)
    assert len(ins) == 255 and sum(A[k][0] for _, _, k, _ in ins) == 3589
    assert sum(A2[i][0] * A[k][0] for i, _, k, _ in ins) == 118482          # node TP weight count


@pytest.mark.parametrize("unrotate", [True, False])
def test_message_pack_program_vs_golden(golden_dir, unrotate):
    f = load(golden_dir, "message_pack_block")
    sd, i = f["weights"], f["inputs"]
    lay = P.PlanarLayout(MINI)
    lmax = 3
    n = i["sh"][:, 1:4] / math.sqrt(3.0)                                    # unit edge direction in e3nn axis order
    D = emu.edge_wigner_all(n, lmax)
    xs, xd, fe = (emu.rotate_rows(lay.to_planar(i[k]), lay, D, lmax) for k in ("src", "dst", "edge_feats"))
    hn = emu.radial_hidden(i["rbf"], P.radial_hidden_weights(sd, "node_weight_generator", emu.SILU_CST))
    he = emu.radial_hidden(i["rbf"], P.radial_hidden_weights(sd, "edge_weight_generator", emu.SILU_CST))
    prog = P.build_message_pack_program(sd, MINI, MINI, SH, MINI, unrotate=unrotate)
    outp = emu.run_program(prog, [xs, xd, fe], (hn, he), D, lmax)
    if not unrotate:
        outp = emu.rotate_rows(outp, lay, D, lmax, transpose=True)
    assert rel(lay.from_planar(outp), f["outputs"]["out"]) < 1e-6           # weights are packed in fp32


def test_linear_program(golden_dir):
    f = load(golden_dir, "residual_block")
    gate_in = str(f["meta"]["gate_irreps_in"])
    lay_in, lay_out = P.PlanarLayout(MINI), P.PlanarLayout(gate_in)
    prog = P.build_linear_program(f["weights"]["linear1.weight"], MINI, gate_in)
    x = f["inputs"]["x"]
    y = lay_out.from_planar(emu.run_program(prog, [lay_in.to_planar(x)]))
    # reference o3.Linear semantics, independent numpy statement
    import oracle.e3 as e3
    import torch
    lin = e3.Linear(MINI, gate_in)
    lin.weight.data = torch.from_numpy(f["weights"]["linear1.weight"])
    assert rel(y, lin(torch.from_numpy(x)).detach().numpy()) < 1e-6


def test_message_pack_program_x4_path_vs_oracle():
    """irreps with 16-/32-channel blocks exercise the permuted-K float4 (x4) operand packing; reference = the oracle."""
    import torch
    from oracle import hamgnn_ref as R
    irr, sh = "16x0e+12x0o+32x1o+4x1e+7x2e", "0e+1o+2e"
    torch.manual_seed(0)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        ref = R.MessagePackBlock(irr, irr, sh, irr, "8x0e", radial_MLP=[16, 16])
        E = 19
        g = torch.Generator().manual_seed(1)
        src, dst, ef = (torch.randn(E, ref.irreps_node_feats.dim, generator=g) for _ in range(3))
        vec = torch.randn(E, 3, generator=g)
        n = torch.nn.functional.normalize(vec, dim=-1)
        from oracle import e3
        shv = e3.spherical_harmonics([0, 1, 2], n, True, "component")
        rbf = torch.randn(E, 8, generator=g)
        out = ref(src, dst, ef, shv, rbf).detach().numpy()
    finally:
        torch.set_default_dtype(prev)
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    lay = P.PlanarLayout(irr)
    D = emu.edge_wigner_all(n.numpy(), 2)
    xs, xd, fe = (emu.rotate_rows(lay.to_planar(t.numpy()), lay, D, 2) for t in (src, dst, ef))
    hn = emu.radial_hidden(rbf.numpy(), P.radial_hidden_weights(sd, "node_weight_generator", emu.SILU_CST))
    he = emu.radial_hidden(rbf.numpy(), P.radial_hidden_weights(sd, "edge_weight_generator", emu.SILU_CST))
    prog = P.build_message_pack_program(sd, irr, irr, sh, irr, unrotate=True)
    assert any(int(it[17]) for it in prog.item_table) and any(not int(it[17]) for it in prog.item_table)
    outp = emu.run_program(prog, [xs, xd, fe], (hn, he), D, 2)
    assert rel(lay.from_planar(outp), out) < 1e-6


def test_radial_scale_split_half_precision_twins_vs_oracle(monkeypatch):
    """r6: programs with 64 hidden units carry, behind every W3 fragment block, its split-half-precision twin (hi = f16(w), lo = f16(w - hi), K-slots paired as
    the kernel's resident hidden rows are).  (1) hi + lo reproduces the fp32 weight to 2^-21; (2) the device-side refresh (ops.DeviceProgram.refresh_w3_split
    with the torch twin of hg_w3_split_refill) writes the same bytes as the host packer; (3) the emulator fed from the TWINS (the kernel's arithmetic: W_lo h_hi + W_hi h_lo + W_hi h_hi) agrees
    with the fp64 oracle to 3e-6 and with the exact-table form to 2e-6 -- but not to 1e-9: the path is exercised; (4) a weight beyond the half-precision
    range clears the part record's flag."""
    import torch
    from oracle import hamgnn_ref as R, e3
    from hamgnn_amd import ops
    from tests import cpu_ops
    monkeypatch.setattr(ops, "w3_split_refill", cpu_ops.w3_split_refill)      # (the product's refill is one HIP launch: hg_w3_split_refill; its torch twin here, the GPU suite compares the kernel)
    irr, sh = "16x0e+12x0o+32x1o+4x1e+7x2e", "0e+1o+2e"
    torch.manual_seed(0)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        ref = R.MessagePackBlock(irr, irr, sh, irr, "8x0e", radial_MLP=[64, 64])
        E = 19
        g = torch.Generator().manual_seed(1)
        src, dst, ef = (torch.randn(E, ref.irreps_node_feats.dim, generator=g) for _ in range(3))
        n = torch.nn.functional.normalize(torch.randn(E, 3, generator=g), dim=-1)
        shv = e3.spherical_harmonics([0, 1, 2], n, True, "component")
        rbf = torch.randn(E, 8, generator=g)
        out = ref(src, dst, ef, shv, rbf).detach().numpy()
    finally:
        torch.set_default_dtype(prev)
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    lay = P.PlanarLayout(irr)
    D = emu.edge_wigner_all(n.numpy(), 2)
    xs, xd, fe = (emu.rotate_rows(lay.to_planar(t.numpy()), lay, D, 2) for t in (src, dst, ef))
    hn = emu.radial_hidden(rbf.numpy(), P.radial_hidden_weights(sd, "node_weight_generator", emu.SILU_CST))
    he = emu.radial_hidden(rbf.numpy(), P.radial_hidden_weights(sd, "edge_weight_generator", emu.SILU_CST))
    prog = P.build_message_pack_program(sd, irr, irr, sh, irr, unrotate=True)
    assert prog.hidden_pad == 64 and prog.w3_regions and prog.w3_split_ok
    n_tp = int((prog.item_table[:, 0] == P.IT_TP).sum())
    assert len(prog.w3_regions) == n_tp and sorted(o for o, _ in prog.w3_regions) == sorted(int(it[12]) for it in prog.item_table if int(it[0]) == P.IT_TP)
    # (1) the twins decode to the weights: w 2^sw = hi + 2^-11 lo to 22 bits, every half normal or zero where it matters, the largest scaled weight in [2^12, 2^13)
    sw = prog.w3_exp
    big_ = max(float(np.abs(prog.weights[off:off + 4 * rtm * 256]).max()) for off, rtm in prog.w3_regions) * 2.0 ** sw
    assert 2.0 ** 12 <= big_ < 2.0 ** 13
    for off, rtm in prog.w3_regions:
        nfl = 4 * rtm * 256
        srcidx, dstidx = P.w3_split_index(rtm)
        tw = prog.weights[off + nfl:off + 2 * nfl].view(np.uint32)
        dec = lambda term, half: ((tw[dstidx[term]] >> (16 * half)) & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64)
        for half in (0, 1):
            w = prog.weights[off + srcidx[:, half]].astype(np.float64) * 2.0 ** sw
            hi, lo = dec(0, half), dec(1, half)
            hi[np.abs(hi) < 2.0 ** -14], lo[np.abs(lo) < 2.0 ** -14] = 0.0, 0.0          # what the matrix pipe sees
            # 22 bits relative; below the halves' normal range an absolute floor of 2^-24 in scaled units = 2^-36 of the largest weight
            assert (np.abs(hi + lo * 2.0 ** -11 - w) <= 2.0 ** -21 * np.abs(w) + 2.0 ** -24).all()
    # (2) device-side refresh == host packer, byte for byte
    dp = ops.DeviceProgram(prog, torch.device("cpu"), schedule="is")
    want = prog.weights.copy()
    for off, rtm in prog.w3_regions:
        dp.weights[off + 4 * rtm * 256:off + 8 * rtm * 256] = 0.0
    dp.refresh_w3_split()
    assert np.array_equal(dp.weights.numpy().view(np.uint32), want.view(np.uint32))
    ops.check_w3_split()
    sc = P.is_schedule(prog)
    assert int(sc.part_table[0][12]) == 1 and int(dp.part_table_host(sc)[0][12]) == 1
    # (3) the kernel's arithmetic from the twins
    exact = emu.run_program_is(prog, sc, [xs, xd, fe], (hn, he), D, 2)
    emu.S_F16 = True
    try:
        split = emu.run_program_is(prog, sc, [xs, xd, fe], (hn, he), D, 2)
    finally:
        emu.S_F16 = False
    assert rel(lay.from_planar(exact), out) < 1e-6
    assert rel(lay.from_planar(split), out) < 3e-6 and 1e-9 < rel(split, exact) < 2e-6, (rel(lay.from_planar(split), out), rel(split, exact))
    # (4) out of the half-precision range: the fp32 form
    sd_big = dict(sd)
    k_last = sorted(k for k in sd if k.startswith("node_weight_generator.layer") and k.endswith(".weight"))[-1]
    sd_big[k_last] = sd[k_last] * 1e6                          # (a uniformly larger layer only moves the exponent: still split)
    big = P.build_message_pack_program(sd_big, irr, irr, sh, irr, unrotate=True)
    assert big.w3_split_ok and big.w3_exp < prog.w3_exp - 15 and int(P.is_schedule(big).part_table[0][13]) == big.w3_exp
    sd_big[k_last] = sd[k_last] * float("inf")
    assert not P.build_message_pack_program({**sd, k_last: np.where(np.arange(sd[k_last].size).reshape(sd[k_last].shape) == 0, np.inf, sd[k_last])}, irr, irr, sh, irr, unrotate=True).w3_split_ok
    dp.weights[prog.w3_regions[0][0]] = 1e6                   # a weight that outgrew the program's exponent after a refresh on the device: the fp32 form
    dp.refresh_w3_split()
    ops.check_w3_split()
    assert int(dp.part_table_host(sc)[0][12]) == 0


def test_message_pack_lite_program_vs_golden(golden_dir):
    f = load(golden_dir, "message_pack_block_lite")
    sd, i = f["weights"], f["inputs"]
    lay = P.PlanarLayout(MINI)
    n = i["sh"][:, 1:4] / math.sqrt(3.0)
    D = emu.edge_wigner_all(n, 3)
    xs, xd, fe = (emu.rotate_rows(lay.to_planar(i[k]), lay, D, 3) for k in ("src", "dst", "edge_feats"))
    h = emu.radial_hidden(i["rbf"], P.radial_hidden_weights(sd, "weight_generator_combine", emu.SILU_CST))
    prog = P.build_message_pack_program_lite(sd, MINI, MINI, SH, MINI, unrotate=True)
    outp = emu.run_program(prog, [xs, xd, fe], (h, None), D, 3)
    assert rel(lay.from_planar(outp), f["outputs"]["out"]) < 1e-6
    # the same program on the input-stationary schedule (r3: IT_LINC items in the phases of their input blocks, the segments' IT_POST items as
    # the last phase), one and several workgroups per 16-edge tile
    # ... and with the paths of every (input irrep, output irrep) pair folded into one item with a weight matrix per column (IT_LINM)
    progf = P.build_message_pack_program_lite(sd, MINI, MINI, SH, MINI, unrotate=True, fold=True)
    assert progf.item_table.shape[0] < prog.item_table.shape[0] and progf.mfma_per_wave < prog.mfma_per_wave
    for pr in (prog, progf):
        for parts in (1, 3):
            sc = P.is_schedule(pr, parts)
            assert (sc.item_table[:, 0] == P.IT_STREAM).any() == (pr is progf)        # the folded items run as step streams (plan._lite_streams)
            assert (sc.item_table[:, 0] == P.IT_POST).sum() == pr.seg_table.shape[0] and sc.part_table[:, 11].all()
            assert not sc.part_table[:, 7].any()               # no private tile copies: the post-op runs on the shared tiles
            outi = emu.run_program_is(pr, sc, [xs, xd, fe], (h, None), D, 3)
            assert rel(lay.from_planar(outi), f["outputs"]["out"]) < 1e-6


def _merge_emu(yp, slot_tab, ptr, idx, val):
    """numpy twin of ham_merge_kernel without rotation (hamgnn_amd/csrc/head.hip)."""
    coef = np.stack([yp[:, b + a * st] for (_, a, b, st) in slot_tab], axis=1)
    out = np.zeros((yp.shape[0], len(ptr) - 1))
    for q in range(len(ptr) - 1):
        sl = slice(ptr[q], ptr[q + 1])
        out[:, q] = coef[:, idx[sl]] @ val[sl].astype(np.float64)
    return out


@pytest.mark.parametrize("ham_type,nao", [("openmx", 19), ("abacus", 13)])
def test_head_linear_and_merge_tables(ham_type, nao):
    """HamLayer.linear_transform regrouped by (L,p) + merge_tensor_components + reorder_matrix tables vs the oracle."""
    import torch
    from oracle import hamgnn_ref as R
    from hamgnn_amd import basis as B
    torch.manual_seed(0)
    ref = R.HamGNNPlusPlusOut(MINI, MINI, nao_max=nao, ham_type=ham_type).double()
    t = B.basis_table(ham_type, nao)
    row = so3.Irreps(t["row"])
    hirr = P.ham_irreps(row)
    W = ref.onsite_hamiltonian_network.linear_transform.weight.detach().numpy()
    prog, girr, slot_pos = P.build_ham_linear_program(W, MINI, hirr)
    x = np.random.default_rng(0).standard_normal((5, so3.Irreps(MINI).dim))
    yp = emu.run_program(prog, [P.PlanarLayout(MINI).to_planar(x)])
    tabs = P.ham_merge_tables(row, nao, t["index_change"], t["minus_index"], girr, slot_pos)
    got = _merge_emu(yp, *tabs)
    want = ref.reorder_matrix(ref.merge_tensor_components(ref.onsite_hamiltonian_network.linear_transform(torch.from_numpy(x))))
    assert rel(got, want.detach().numpy()) < 1e-6
    # the same grouped Linear as tables of the streaming kernel (the default path, csrc/linear.hip)
    mats, girr2, slot2 = P.ham_linear_mats(W, MINI, hirr)
    assert str(girr2) == str(girr) and slot2 == slot_pos
    ys = emu.run_linear_tables(P.linear_tables(mats, P.PlanarLayout(MINI), P.PlanarLayout(girr)), P.PlanarLayout(MINI).to_planar(x))
    assert rel(ys, yp) < 1e-6


def test_ham_merge_adjoint_tables():
    """<C y, g> == <y, C^T g> for the CSR merge map and its transposed tables (plan.ham_merge_adjoint_tables), plus the scatter map"""
    from hamgnn_amd import basis as B
    t = B.basis_table("openmx", 19)
    row = so3.Irreps(t["row"])
    hirr = P.ham_irreps(row)
    W = np.random.default_rng(0).normal(size=sum(m for m, l, p in so3.Irreps(MINI) for _, L, pp in hirr if (l, p) == (L, pp)))
    mats, girr, slot_pos = P.ham_linear_mats(W, MINI, hirr)
    st, ptr, idx, val = P.ham_merge_tables(row, 19, t["index_change"], t["minus_index"], girr, slot_pos)
    glay = P.PlanarLayout(girr)
    sid, pT, iT, vT, scat = P.ham_merge_adjoint_tables(st, ptr, idx, val, glay.dim)
    rng = np.random.default_rng(1)
    yp, g = rng.normal(size=(3, glay.dim)), rng.normal(size=(3, 361))
    Cy = _merge_emu(yp, st, ptr, idx, val)
    CTg_coef = _merge_emu(g, sid, pT, iT, vT)                       # [3, ncoef]
    CTg = np.where(scat[None, :] >= 0, CTg_coef[:, np.maximum(scat, 0)], 0.0)
    assert abs((Cy * g).sum() - (yp * CTg).sum()) < 1e-9 * abs((Cy * g).sum())


RICH6 = "4x0e+4x0o+2x1o+2x1e+2x2e+2x2o+2x3o+2x3e+1x4e+1x4o+1x5o+1x5e+1x6e+1x6o"


@pytest.mark.parametrize("nao,ham_type,MINI", [(13, "abacus", MINI), (27, "abacus", RICH6), (27, "abacus", MINI)])
def test_head_su2_tables(nao, ham_type, MINI):
    """su2: only copies 0/2 of the 4 x required irreps are computed; merge table == get_H + reorder + spin interleave.
    nao 27 (s s s s p p d d f): the L x 1 couplings reach l = 7, for which l <= 6 features have no o3.Linear path -- those outputs
    are structural zeros that the planner drops (also every l >= 4 output when the features stop at l = 3)."""
    import torch
    from oracle import hamgnn_ref as R
    from hamgnn_amd import basis as B
    torch.manual_seed(1)
    ref = R.HamGNNPlusPlusOut(MINI, MINI, nao_max=nao, ham_type=ham_type, soc_switch=True).double()
    t = B.basis_table(ham_type, nao)
    row = so3.Irreps(t["row"])
    half = P.su2_irreps(row)
    S = len(half)
    full = so3.Irreps(list(half) * 4)
    assert str(full) == str(so3.Irreps(str(ref.onsite_hamiltonian_network.linear_transform.irreps_out)))
    W = ref.onsite_hamiltonian_network.linear_transform.weight.detach().numpy()
    keep = [(s // S) in (0, 2) for s in range(4 * S)]
    prog, girr, slot_pos = P.build_ham_linear_program(W, MINI, full, keep)
    x = np.random.default_rng(1).standard_normal((4, so3.Irreps(MINI).dim))
    yp = emu.run_program(prog, [P.PlanarLayout(MINI).to_planar(x)])
    tabs = P.su2_merge_tables(row, nao, t["index_change"], t["minus_index"], girr, slot_pos)
    got = _merge_emu(yp, *tabs)
    H = ref.su2_get_H(ref.onsite_hamiltonian_network.linear_transform(torch.from_numpy(x)))
    H = ref.reorder_matrix(H.reshape(-1, nao * nao)).reshape(-1, 2, 2, nao, nao).swapaxes(2, 3).reshape(-1, 4 * nao * nao)
    want = torch.cat([H.real, H.imag], 1).detach().numpy()
    assert rel(got, want) < 1e-6


def test_input_stationary_schedule_vs_golden(golden_dir):
    """plan.is_schedule (csrc/tp_is.hip): same items regrouped by input block -> identical result; schedule invariants."""
    f = load(golden_dir, "message_pack_block")
    sd, i = f["weights"], f["inputs"]
    lay = P.PlanarLayout(MINI)
    lmax = 3
    n = i["sh"][:, 1:4] / math.sqrt(3.0)
    D = emu.edge_wigner_all(n, lmax)
    xs, xd, fe = (emu.rotate_rows(lay.to_planar(i[k]), lay, D, lmax) for k in ("src", "dst", "edge_feats"))
    hn = emu.radial_hidden(i["rbf"], P.radial_hidden_weights(sd, "node_weight_generator", emu.SILU_CST))
    he = emu.radial_hidden(i["rbf"], P.radial_hidden_weights(sd, "edge_weight_generator", emu.SILU_CST))
    skip = np.random.default_rng(3).standard_normal(so3.Irreps(MINI).dim * 0 + sum(m * m for m, _, _ in so3.Irreps(MINI)))
    prog = P.build_message_pack_program(sd, MINI, MINI, SH, MINI, unrotate=True)
    sched = P.is_schedule(prog)
    assert sched.item_table.shape == (prog.item_table.shape[0], P.IS_ITEM_I32) and sched.lds_floats * 4 <= P.IS_LDS_BYTES
    for rec in sched.item_table:                              # the segment fields embedded in every item record (csrc/tp_is.hip:ItemRec)
        sg = sched.seg_table[rec[19]]
        assert (rec[20], rec[21], rec[22]) == (sg[0], sg[1], sg[2])
        assert sched.rowtab[rec[23]] == sg[5] + sg[0] * 16                     # first row-table entry: centre column of row 0 of the tile
    assert sched.ctr_off == sched.stage_off + sched.stage_floats and sched.balance > 0.5
    outp = emu.run_program_is(prog, sched, [xs, xd, fe], (hn, he), D, lmax)
    assert rel(lay.from_planar(outp), f["outputs"]["out"]) < 1e-6
    assert rel(outp, emu.run_program(prog, [xs, xd, fe], (hn, he), D, lmax)) < 1e-12
    for parts in (2, 3, 7):                                   # split launches for small crystals: disjoint segment sets, private tile copies
        sp = P.is_schedule(prog, parts)
        assert sp.part_table.shape[0] == parts and all(int(p[7]) > 0 for p in sp.part_table)
        assert rel(emu.run_program_is(prog, sp, [xs, xd, fe], (hn, he), D, lmax), outp) < 1e-12
    # with the PairInteractionBlock skip o3.Linear folded in (extra linear items in the edge-row phases)
    prog2 = P.build_message_pack_program(sd, MINI, MINI, SH, MINI, unrotate=False, skip_weight=skip)
    s2 = P.is_schedule(prog2)
    a = emu.run_program_is(prog2, s2, [xs, xd, fe], (hn, he), D, lmax)
    assert rel(a, emu.run_program(prog2, [xs, xd, fe], (hn, he), D, lmax)) < 1e-12
    # replayed hipGraphs only (graph_capture.CapturedForward): one workgroup per (segment set, share of its phases), tiles added into zero-filled rows
    for pr, want in ((prog, outp), (prog2, a)):
        s2d = P.is_schedule(pr, ("2d", 3, 2))
        assert 3 <= s2d.part_table.shape[0] <= 6
        assert rel(emu.run_program_is(pr, s2d, [xs, xd, fe], (hn, he), D, lmax), want) < 1e-12


@pytest.mark.parametrize("which", ["A", "B"])
def test_input_stationary_schedule_shipped_irreps(which):
    """schedule invariants for the shipped irreps sets (bench.IRREPS): every item exactly once, one owner group per (phase, segment),
    staged blocks inside the staging area, LDS budget, Wigner batches inside the staging area."""
    import torch
    import bench
    from hamgnn_amd import nn as hnn
    irr = bench.IRREPS[which]
    torch.manual_seed(0)
    m = hnn.MessagePackBlock(irr, irr, bench.SH, irr, 64, [64, 64])
    skip = np.zeros(sum(mm * mm for mm, _, _ in so3.Irreps(irr)))
    prog = P.build_message_pack_program(hnn._np_sd(m), irr, irr, bench.SH, irr, True, skip)
    sc = P.is_schedule(prog)
    assert sc.lds_floats * 4 <= P.IS_LDS_BYTES and sc.item_table.shape == (prog.item_table.shape[0], P.IS_ITEM_I32)
    for parts in (8, 13):                                     # split launches: every part fits, worst part well below the whole program
        sp = P.is_schedule(prog, parts)
        assert sp.lds_floats * 4 <= P.IS_LDS_BYTES and len(sp.part_cost) == min(parts, sc.seg_table.shape[0])
        assert max(sp.part_cost) < 0.3 * sc.part_cost[0] and all(int(p[7]) > 0 for p in sp.part_table)
    seen = np.zeros(sc.item_table.shape[0], dtype=int)
    for b0, b1, g0, g1 in sc.phase_table[:, :4]:
        used, offs = 0, set()
        for blk in sc.block_table[b0:b1]:
            size = -(-((2 * int(blk[4]) + 1) * (int(blk[3]) // 4)) // 4) * 256 * int(blk[5])
            assert int(blk[6]) == used
            used += size
            offs.add(int(blk[6]))
        assert used <= sc.stage_floats
        owners = set()
        for ib, ie in sc.group_table[g0:g1]:
            segs = set(int(x) for x in sc.item_table[ib:ie, 19])
            assert len(segs) == 1 and not (segs & owners)
            owners |= segs
            seen[ib:ie] += 1
            assert all(int(o) in offs for o in sc.item_table[ib:ie, 1])
    assert (seen == 1).all()
    for sg in sc.seg_table:
        if int(sg[7]) & P.SEG_UNROTATE:
            assert int(sg[6]) + -(-((2 * int(sg[0]) + 1) ** 2) // 4) * 64 <= sc.stage_floats
    assert int(sc.seg_table[0][7]) & P.SEG_NEWBATCH
    # tile offsets are disjoint and end at the trash row
    ends = sorted((int(s[5]), int(s[5]) + int(s[1]) * ((2 * int(s[0]) + 1) * 16 + 4)) for s in sc.seg_table)
    assert all(a[1] <= b[0] for a, b in zip(ends, ends[1:])) and ends[-1][1] <= sc.trash_off


@pytest.mark.parametrize("which", ["A", "B"])
def test_split_launch_work_groups_are_dealt_not_claimed(which):
    """r6: a part with private tile copies per wave deals its work groups statically -- group g0 + k * waves + w is the k-th of wave w,
    short streams end with empty groups -- so which wave adds an item into which copy does not depend on a run's timing: one summation order per launch.
    Checks the table form the kernel relies on (csrc/tp_is.hip: gi = g0 + wave, += NW), that every item is still issued exactly once, that the dealing is
    the LPT one (no wave above the list-scheduling bound) and that the emulator agrees with the single-part schedule on the same inputs."""
    import torch
    import bench
    from hamgnn_amd import nn as hnn
    irr = bench.IRREPS[which]
    torch.manual_seed(0)
    m = hnn.MessagePackBlock(irr, irr, bench.SH, irr, 64, [64, 64])
    skip = np.zeros(sum(mm * mm for mm, _, _ in so3.Irreps(irr)))
    prog = P.build_message_pack_program(hnn._np_sd(m), irr, irr, bench.SH, irr, True, skip)
    hp4 = prog.hidden_pad // 4

    def cost_of(it):                                           # plan._item_cost on a schedule record ([22] = row tiles of GEMM2's output)
        nsrc, nc, rtm = (2 if it[2] >= 0 else 1), 2 * int(it[6]) + 1, int(it[9])
        return nsrc * int(it[8]) * rtm * nc + P.ITEM_OVERHEAD + ((hp4 * rtm + int(it[22]) * int(it[18]) * nc) if int(it[0]) == P.IT_TP else 0)

    for parts in (8, prog.seg_table.shape[0]):
        sp = P.is_schedule(prog, parts)
        seen = np.zeros(sp.item_table.shape[0], dtype=int)
        for pt in sp.part_table:
            assert int(pt[7]) > 0                              # every part of the shipped sets keeps private copies
            for b0, b1, g0, g1 in sp.phase_table[int(pt[2]):int(pt[2]) + int(pt[3]), :4]:
                assert (g1 - g0) % P.IS_WAVES == 0
                G = sp.group_table[g0:g1].reshape(-1, P.IS_WAVES, 2)
                cost = np.zeros(P.IS_WAVES)
                for k in range(G.shape[0]):
                    for w in range(P.IS_WAVES):
                        ib, ie = (int(v) for v in G[k, w])
                        assert ie - ib in (0, 1)                # an item is its own work group; an empty group ends a short stream ...
                        if ie == ib:
                            assert all(int(G[k2, w, 1]) == int(G[k2, w, 0]) for k2 in range(k, G.shape[0]))      # ... and nothing follows it
                        seen[ib:ie] += 1
                        cost[w] += sum(cost_of(sp.item_table[i]) for i in range(ib, ie))
                dearest = max((cost_of(sp.item_table[int(ib)]) for ib, ie in sp.group_table[g0:g1] if ie > ib), default=0)
                assert cost.max() <= cost.sum() / P.IS_WAVES + dearest      # list-scheduling bound
        assert (seen == 1).all()
        # twice the same tables: the planner itself is deterministic
        sp2 = P.is_schedule(prog, parts)
        assert np.array_equal(sp.group_table, sp2.group_table) and np.array_equal(sp.item_table, sp2.item_table)


@pytest.mark.parametrize("which", ["A", "B"])
def test_merged_items_shipped_irreps(which):
    """small output irreps of one parity class share MFMA row tiles (plan.choose_merge_groups / Program.vsegs): fewer issued MFMAs and
    items, the same results (emulated through the row table on a few edges, single- and multi-part schedules), LDS budget kept"""
    import torch
    import bench
    from oracle import hamgnn_ref as R, e3
    irr, sh = bench.IRREPS[which], bench.SH
    groups = P.choose_merge_groups(irr, irr, sh, irr, 16)
    assert groups and all(len(G) > 1 for G in groups)
    I = so3.Irreps(irr)
    for G in groups:
        assert len({(I[k][1] + (I[k][2] == -1)) % 2 for k in G}) == 1 and sum(I[k][0] for k in G) <= 64
    torch.manual_seed(1)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        ref = R.MessagePackBlock(irr, irr, sh, irr, "8x0e", radial_MLP=[16, 16])
        E = 5
        g = torch.Generator().manual_seed(1)
        src, dst, ef = (torch.randn(E, ref.irreps_node_feats.dim, generator=g) for _ in range(3))
        n = torch.nn.functional.normalize(torch.randn(E, 3, generator=g), dim=-1)
        shv = e3.spherical_harmonics(list(range(6)), n, True, "component")
        rbf = torch.randn(E, 8, generator=g)
        want = ref(src, dst, ef, shv, rbf).detach().numpy()
    finally:
        torch.set_default_dtype(prev)
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    lay = P.PlanarLayout(irr)
    D = emu.edge_wigner_all(n.numpy(), 6)
    xs, xd, fe = (emu.rotate_rows(lay.to_planar(t.numpy()), lay, D, 6) for t in (src, dst, ef))
    hn = emu.radial_hidden(rbf.numpy(), P.radial_hidden_weights(sd, "node_weight_generator", emu.SILU_CST))
    he = emu.radial_hidden(rbf.numpy(), P.radial_hidden_weights(sd, "edge_weight_generator", emu.SILU_CST))
    skip = np.random.default_rng(0).normal(size=sum(mm * mm for mm, _, _ in I))
    plain = P.build_message_pack_program(sd, irr, irr, sh, irr, True)
    merged = P.build_message_pack_program(sd, irr, irr, sh, irr, True, merge_groups=groups)
    assert merged.mfma_per_wave < 0.97 * plain.mfma_per_wave and merged.item_table.shape[0] < 0.9 * plain.item_table.shape[0]
    sc = P.is_schedule(merged)
    assert sc.lds_floats * 4 <= P.IS_LDS_BYTES and sc.balance > 0.85
    got = emu.run_program_is(merged, sc, [xs, xd, fe], (hn, he), D, 6)
    assert rel(lay.from_planar(got), want) < 1e-6
    for parts in (3, 8):                                      # the members of a merged item stay in one part
        sp = P.is_schedule(merged, parts)
        assert rel(lay.from_planar(emu.run_program_is(merged, sp, [xs, xd, fe], (hn, he), D, 6)), want) < 1e-6
    s2d = P.is_schedule(plain, ("2d", plain.seg_table.shape[0], 3))      # (experiment) every output segment's phases on three workgroups
    assert s2d.atomic_out and s2d.part_table.shape[0] > (2 if which == "A" else 1) * plain.seg_table.shape[0]
    assert rel(lay.from_planar(emu.run_program_is(plain, s2d, [xs, xd, fe], (hn, he), D, 6)), want) < 1e-6
    # with the PairInteractionBlock skip Linear (plain IT_LIN items into tiles that merged items write as well)
    m2 = P.build_message_pack_program(sd, irr, irr, sh, irr, False, skip, merge_groups=groups)
    p2 = P.build_message_pack_program(sd, irr, irr, sh, irr, False, skip)
    a = emu.run_program_is(m2, P.is_schedule(m2), [xs, xd, fe], (hn, he), None, None)
    b = emu.run_program_is(p2, P.is_schedule(p2), [xs, xd, fe], (hn, he), None, None)
    assert rel(a, b) < 1e-9


def test_input_stationary_schedule_falls_back_when_too_wide():
    """tiles of all output segments beyond the LDS budget -> NotImplementedError (ops.DeviceProgram 'auto' keeps the segment-stationary kernel)"""
    import torch
    from hamgnn_amd import nn as hnn
    irr, sh = "64x0e+64x0o+48x1o+48x1e+32x2e+32x2o+20x3o+20x3e", "0e+1o+2e+3o"
    torch.manual_seed(0)
    m = hnn.MessagePackBlock(irr, irr, sh, irr, 8, [16, 16])
    prog = P.build_message_pack_program(hnn._np_sd(m), irr, irr, sh, irr, True)
    with pytest.raises(NotImplementedError):
        P.is_schedule(prog)


def test_sym_contraction_tables_vs_oracle(golden_dir):
    """CorrProductBlock's symmetric contraction (a21): sparse U tables + concatenated weights == the oracle's dense einsums"""
    import torch
    from oracle import mace_ref as M
    f = load(golden_dir, "corr_product_block")
    irr, nh, nel = str(f["meta"]["irreps"]), int(f["meta"]["num_hidden"]), int(f["meta"]["num_elements"])
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        blk = M.CorrProductBlock(irr, nh, 2, nel, True)
    finally:
        torch.set_default_dtype(prev)
    blk.load_state_dict({k: torch.as_tensor(v) for k, v in f["weights"].items()}, strict=False)
    hid = P.corr_hidden_irreps(irr, nh)
    tab = P.sym_contraction_tables(hid, 2)
    cons = blk.prod.symmetric_contractions.contractions
    assert tab["K2"] == [c.U(2).shape[-1] for c in cons] and tab["K1"] == [c.U(1).shape[-1] for c in cons]
    W2 = np.concatenate([c.weights_max.detach().numpy() for c in cons], axis=1)
    W1 = np.concatenate([c.weights[0].detach().numpy() for c in cons], axis=1)
    rng = np.random.default_rng(2)
    lay = P.PlanarLayout(hid)
    h = rng.standard_normal((6, so3.Irreps(str(hid)).dim)) if hasattr(so3.Irreps, "dim") else None
    h = rng.standard_normal((6, sum(m * (2 * l + 1) for m, l, _ in hid)))
    z = rng.integers(0, nel, size=6)
    got = lay.from_planar(emu.sym_contraction(tab, lay.to_planar(h), z, W1, W2, nh, lay.dim))
    ht = torch.from_numpy(h)
    want = blk.prod.symmetric_contractions(M.reshape_irreps(blk.irreps_hidden, ht), torch.nn.functional.one_hot(torch.from_numpy(z), nel).double())
    assert rel(got, want.detach().numpy()) < 1e-6


def test_sym_contraction_backward_vs_autograd(golden_dir):
    """SURVEY 8f-3 x a21: gradients of the symmetric contraction (hamgnn_amd/backward_corr.py, sparse tables) with respect to the hidden
    rows and the element-dependent weights vs torch.autograd through the oracle's dense einsums"""
    import torch
    from oracle import mace_ref as M
    from hamgnn_amd.backward_corr import sym_contraction_backward
    f = load(golden_dir, "corr_product_block")
    irr, nh, nel = str(f["meta"]["irreps"]), int(f["meta"]["num_hidden"]), int(f["meta"]["num_elements"])
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        blk = M.CorrProductBlock(irr, nh, 2, nel, True)
    finally:
        torch.set_default_dtype(prev)
    blk.load_state_dict({k: torch.as_tensor(v) for k, v in f["weights"].items()}, strict=False)
    hid = P.corr_hidden_irreps(irr, nh)
    tab = P.sym_contraction_tables(hid, 2)
    cons = blk.prod.symmetric_contractions.contractions
    lay = P.PlanarLayout(hid)
    rng = np.random.default_rng(3)
    N = 7
    h = torch.from_numpy(rng.standard_normal((N, sum(m * (2 * l + 1) for m, l, _ in hid)))).requires_grad_()
    z = torch.from_numpy(rng.integers(0, nel, size=N))
    out = blk.prod.symmetric_contractions(M.reshape_irreps(blk.irreps_hidden, h), torch.nn.functional.one_hot(z, nel).double())
    Gm = torch.from_numpy(rng.standard_normal(tuple(out.shape)))
    (out * Gm).sum().backward()
    W2 = torch.cat([c.weights_max.detach() for c in cons], 1)
    W1 = torch.cat([c.weights[0].detach() for c in cons], 1)
    hp = torch.from_numpy(lay.to_planar(h.detach().numpy()))
    gp = torch.from_numpy(lay.to_planar(Gm.numpy()))
    g_h, gW1, gW2 = sym_contraction_backward(tab, hp, z, W1, W2, nh, gp, chunk=3)
    assert rel(lay.from_planar(g_h.numpy()), h.grad.numpy()) < 1e-6           # (the tables hold their coefficients in fp32)
    want2 = torch.cat([c.weights_max.grad for c in cons], 1)
    want1 = torch.cat([c.weights[0].grad for c in cons], 1)
    assert rel(gW2.numpy(), want2.numpy()) < 1e-6 and rel(gW1.numpy(), want1.numpy()) < 1e-6


def test_sym_contraction_nu3_term_vs_oracle_and_autograd(golden_dir):
    """correlation 3: plan.sym_contraction_tables' sparse U_3 entries (forward: the numpy twin of hg_sym_contraction3; backward: hamgnn_amd/corr3.py) against the
    oracle's dense einsum chain (pinned on the reference's own U matrices and outputs by the corr_product_block_nu3 fixture) and autograd"""
    import torch
    from oracle import mace_ref as M
    from hamgnn_amd.corr3 import sym3_backward
    from hamgnn_amd.backward_corr import sym_contraction_backward
    f = load(golden_dir, "corr_product_block_nu3")
    irr, nh, nel = str(f["meta"]["irreps"]), int(f["meta"]["num_hidden"]), int(f["meta"]["num_elements"])
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        blk = M.CorrProductBlock(irr, nh, 3, nel, True)
    finally:
        torch.set_default_dtype(prev)
    blk.load_state_dict({k: torch.as_tensor(v) for k, v in f["weights"].items()}, strict=False)
    hid = P.corr_hidden_irreps(irr, nh)
    tab = P.sym_contraction_tables(hid, 3)
    cons = blk.prod.symmetric_contractions.contractions
    assert all(tab[f"K{nu}"] == [c.U(nu).shape[-1] for c in cons] for nu in (1, 2, 3))
    lay = P.PlanarLayout(hid)
    rng = np.random.default_rng(5)
    N = 7
    h = torch.from_numpy(rng.standard_normal((N, sum(m * (2 * l + 1) for m, l, _ in hid)))).requires_grad_()
    z = torch.from_numpy(rng.integers(0, nel, size=N))
    out = blk.prod.symmetric_contractions(M.reshape_irreps(blk.irreps_hidden, h), torch.nn.functional.one_hot(z, nel).double())
    Gm = torch.from_numpy(rng.standard_normal(tuple(out.shape)))
    (out * Gm).sum().backward()
    W3 = torch.cat([c.weights_max.detach() for c in cons], 1)
    W2 = torch.cat([c.weights[0].detach() for c in cons], 1)
    W1 = torch.cat([c.weights[1].detach() for c in cons], 1)
    hp = torch.from_numpy(lay.to_planar(h.detach().numpy()))
    gp = torch.from_numpy(lay.to_planar(Gm.numpy()))
    low = torch.from_numpy(emu.sym_contraction(tab, hp.numpy(), z.numpy(), W1.numpy(), W2.numpy(), nh, lay.dim))
    got = torch.from_numpy(emu.sym_contraction3(tab, hp.numpy(), z.numpy(), W3.numpy(), nh, low.numpy()))      # the table-exact twin of csrc/corr3.hip
    assert rel(lay.from_planar(got.numpy()), out.detach().numpy()) < 1e-6            # (the tables hold their coefficients in fp32)
    assert rel(got.numpy(), low.numpy()) > 1e-2                                       # the nu = 3 term is not negligible in this check
    g_h, gW1, gW2 = sym_contraction_backward(tab, hp, z, W1, W2, nh, gp, chunk=3)
    g_h3, gW3 = sym3_backward(tab, hp, z, W3, nh, gp, chunk=2)
    assert rel(lay.from_planar((g_h + g_h3).numpy()), h.grad.numpy()) < 1e-6
    for g, want in ((gW3, [c.weights_max.grad for c in cons]), (gW2, [c.weights[0].grad for c in cons]), (gW1, [c.weights[1].grad for c in cons])):
        assert rel(g.numpy(), torch.cat(want, 1).numpy()) < 1e-6
    # per-node weight blocks (the charge-doped attributes): same gradients, one block per node
    g_hn, gWn = sym3_backward(tab, hp, torch.arange(N), W3[z], nh, gp, per_node=True)
    assert rel(g_hn.numpy(), g_h3.numpy()) < 1e-12
    acc = torch.zeros_like(W3).index_add_(0, z, gWn)
    assert rel(acc.numpy(), gW3.numpy()) < 1e-12


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_sym_contraction_tables_random_irreps_vs_dense_u_matrices(seed):
    """the sparse U_1 / U_2 / U_3 entry lists of plan.sym_contraction_tables == the oracle's dense U_matrix_real (whose values and path
    order the reference's own cg.py confirmed on the fixture irreps, oracle/gen_golden.py) on random irreps: both parities of an l in
    either order, missing l, targets that only some couplings reach"""
    import torch
    from oracle import mace_ref as M
    from oracle.e3 import Irrep, Irreps as OIrreps
    rng = np.random.default_rng(100 + seed)
    cand = [(l, p) for l in range(3) for p in (1, -1)]
    while True:
        pick = [c for c in cand if rng.random() < 0.6]
        if (0, 1) not in pick:
            pick.insert(0, (0, 1))                              # (the contraction takes its channel count from the even scalars)
        by_l = {}
        for l, p in pick:
            by_l.setdefault(l, []).append(p)
        irr = []
        for l in sorted(by_l):
            ps = by_l[l]
            rng.shuffle(ps)
            irr += [f"2x{l}{'e' if p == 1 else 'o'}" for p in ps]
        if sum(2 * int(t[2]) + 1 for t in irr) <= 14:
            break
    irr = "+".join(irr)
    hid = P.corr_hidden_irreps(irr, 2)
    tab = P.sym_contraction_tables(hid, 3)
    coupling = OIrreps([(1, ir) for _, ir in OIrreps(irr)])
    num_ell = tab["num_ell"]
    o = 0
    kg = {1: 0, 2: 0, 3: 0}
    for k, (_, ir) in enumerate(OIrreps(irr)):
        dense = {}
        for nu in (1, 2, 3):
            U = M.u_matrix_real(coupling, ir, nu, dtype=torch.float64).numpy()
            dense[nu] = U if ir.l > 0 else U[None]                # [w, ell.., paths]
            assert tab[f"K{nu}"][k] == U.shape[-1], (irr, k, nu)
        for w in range(ir.dim):
            for nu, ncol in ((1, 1), (2, 2), (3, 3)):
                ent, ptr = tab[f"ent{nu}"], tab[f"ptr{nu}"]
                D = np.zeros(dense[nu].shape[1:])
                for r in range(ptr[o], ptr[o + 1]):
                    idx = tuple(int(v) for v in ent[r, :ncol]) + (int(ent[r, ncol]) - kg[nu],)
                    D[idx] += float(ent[r, -1:].view(np.float32)[0])
                assert np.abs(D - dense[nu][w]).max() < 1e-6, (irr, k, w, nu)
            o += 1
        for nu in (1, 2, 3):
            kg[nu] += tab[f"K{nu}"][k]
    assert o == tab["nout"] and num_ell == coupling.dim


def _random_irreps(rng, lmax):
    """random simplified irreps (distinct (l, p), sorted like the reference's configs: by l, odd/even in random order)"""
    out = []
    for l in range(lmax + 1):
        ps = [p for p in (1, -1) if rng.random() < (0.9 if l == 0 and p == 1 else 0.6)]
        rng.shuffle(ps)
        for p in ps:
            out.append(f"{int(rng.integers(1, 20))}x{l}{'e' if p == 1 else 'o'}")
    return "+".join(out) if out else "3x0e"


@pytest.mark.parametrize("seed", range(16))
def test_random_irreps_both_schedules_vs_oracle(seed):
    """planner generality: random irreps sets (odd multiplicities, missing parities, single-irrep rows), random weights ->
    segment-stationary AND input-stationary schedules, emulated fragment-exactly, vs the oracle MessagePackBlock."""
    import torch
    from oracle import hamgnn_ref as R, e3
    rng = np.random.default_rng(100 + seed)
    lmax = int(rng.integers(1, 4))
    irr = _random_irreps(rng, lmax)
    if "0e" not in irr:
        irr = "5x0e+" + irr
    lsh = int(rng.integers(1, 4))
    sh = "+".join(f"{l}{'e' if l % 2 == 0 else 'o'}" for l in range(lsh + 1))
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        ref = R.MessagePackBlock(irr, irr, sh, irr, "8x0e", radial_MLP=[16, 16])
        E = 17
        g = torch.Generator().manual_seed(seed)
        src, dst, ef = (torch.randn(E, ref.irreps_node_feats.dim, generator=g) for _ in range(3))
        n = torch.nn.functional.normalize(torch.randn(E, 3, generator=g), dim=-1)
        shv = e3.spherical_harmonics(list(range(lsh + 1)), n, True, "component")
        rbf = torch.randn(E, 8, generator=g)
        out = ref(src, dst, ef, shv, rbf).detach().numpy()
    finally:
        torch.set_default_dtype(prev)
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    lay = P.PlanarLayout(irr)
    lm = max(lmax, lsh)
    D = emu.edge_wigner_all(n.numpy(), lm)
    xs, xd, fe = (emu.rotate_rows(lay.to_planar(t.numpy()), lay, D, lm) for t in (src, dst, ef))
    hn = emu.radial_hidden(rbf.numpy(), P.radial_hidden_weights(sd, "node_weight_generator", emu.SILU_CST))
    he = emu.radial_hidden(rbf.numpy(), P.radial_hidden_weights(sd, "edge_weight_generator", emu.SILU_CST))
    prog = P.build_message_pack_program(sd, irr, irr, sh, irr, unrotate=True)
    scale = np.abs(out).max()
    if scale < 1e-12:                                          # no path reaches the output irreps (degenerate draw)
        return
    outp = emu.run_program(prog, [xs, xd, fe], (hn, he), D, lm)
    assert rel(lay.from_planar(outp), out) < 1e-6, irr
    outi = emu.run_program_is(prog, P.is_schedule(prog), [xs, xd, fe], (hn, he), D, lm)
    assert rel(lay.from_planar(outi), out) < 1e-6, irr
    groups = P.choose_merge_groups(irr, irr, sh, irr, 16)
    if groups:                                                 # small output irreps stacked into shared MFMA row tiles
        pm = P.build_message_pack_program(sd, irr, irr, sh, irr, unrotate=True, merge_groups=groups)
        assert pm.item_table.shape[0] < prog.item_table.shape[0]              # (the chooser trades MFMAs against the per-item latency: 60 slots)
        outm = emu.run_program_is(pm, P.is_schedule(pm), [xs, xd, fe], (hn, he), D, lm)
        assert rel(lay.from_planar(outm), out) < 1e-6, (irr, groups)


@pytest.mark.parametrize("seed", range(6))
def test_streaming_linear_tables_vs_oracle(seed):
    """o3.Linear between random irreps sets (missing matches -> zero blocks, multiplicities > 64 -> channel chunks, several inputs per
    output) as tables of the streaming kernel (plan.linear_tables / csrc/linear.hip), emulated fragment-exactly, vs the oracle Linear"""
    import torch
    from oracle import e3
    rng = np.random.default_rng(300 + seed)
    irr_in = _random_irreps(rng, int(rng.integers(1, 4)))
    irr_out = _random_irreps(rng, int(rng.integers(1, 4)))
    if seed == 0:
        irr_in, irr_out = "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e", "199x0e+64x0o+32x1o+16x1e+12x2o+25x2e+3x3o"
    if seed == 1:
        irr_in, irr_out = "5x0e+7x0e+3x1o", "9x0e+2x1o+4x1o"                   # unsimplified: two inputs feed one output
    torch.manual_seed(seed)
    lin = e3.Linear(irr_in, irr_out).double()
    x = torch.randn(21, e3.Irreps(irr_in).dim, dtype=torch.float64)
    want = lin(x).detach().numpy()
    tabs = P.build_linear_tables(lin.weight.detach().numpy(), irr_in, irr_out)
    li, lo = P.PlanarLayout(irr_in), P.PlanarLayout(irr_out)
    got = emu.run_linear_tables(tabs, li.to_planar(x.numpy()))
    assert np.abs(lo.from_planar(got) - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), (irr_in, irr_out)
    pad = np.ones(lo.dim, bool)
    pad[lo.index_map()] = False
    assert not got[:, pad].any()                                               # channel-padding columns are written as zeros
    r1, r2 = rng.normal(size=got.shape), rng.normal(size=got.shape)
    assert np.allclose(emu.run_linear_tables(tabs, li.to_planar(x.numpy()), res=(r1, r2)), got + r1 + r2)


@pytest.mark.parametrize("seed", range(4))
def test_linear_adjoint_tables_vs_autograd(seed):
    """data gradient of o3.Linear = the streaming kernel on transposed blocks (plan.build_linear_adjoint_tables), emulated, vs autograd"""
    import torch
    from oracle import e3
    rng = np.random.default_rng(400 + seed)
    irr_in = _random_irreps(rng, int(rng.integers(1, 4)))
    irr_out = _random_irreps(rng, int(rng.integers(1, 4)))
    if seed == 0:
        irr_in, irr_out = "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e", "199x0e+64x0o+32x1o+16x1e+12x2o+25x2e+3x3o"
    torch.manual_seed(seed)
    lin = e3.Linear(irr_in, irr_out).double()
    x = torch.randn(9, e3.Irreps(irr_in).dim, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(9, e3.Irreps(irr_out).dim, dtype=torch.float64)
    (lin(x) * gy).sum().backward()
    li, lo = P.PlanarLayout(irr_in), P.PlanarLayout(irr_out)
    tabs = P.build_linear_adjoint_tables(lin.weight.detach().numpy(), irr_in, irr_out)
    got = emu.run_linear_tables(tabs, lo.to_planar(gy.numpy()))
    assert np.abs(li.from_planar(got) - x.grad.numpy()).max() <= 1e-6 * max(1.0, np.abs(x.grad.numpy()).max()), (irr_in, irr_out)


@pytest.mark.parametrize("seed", range(3))
def test_message_pack_weight_gradients_vs_autograd(seed):
    """weight gradients of a MessagePackBlock through the two materialisation programs (emulated) + the edge reductions of
    hamgnn_amd/backward_mp.py, every parameter of the block vs torch.autograd through the fp64 oracle"""
    import torch
    from oracle import hamgnn_ref as R, e3
    from hamgnn_amd import backward_mp as BM
    # "zeros" (r5): a first-layer block -- node rows non-zero in 0e only, edge rows in the irreps of the spherical harmonics only; the tables are built
    # WITHOUT the row tiles of the super-paths that read the other irreps (their gradients are exactly zero) and every parameter still matches autograd
    seed, zeros = seed if isinstance(seed, tuple) else (seed, None)
    rng = np.random.default_rng(500 + seed)
    lmax = int(rng.integers(1, 3))
    irr = _random_irreps(rng, lmax)
    if "0e" not in irr:
        irr = "5x0e+" + irr
    lsh = int(rng.integers(1, 3))
    sh = "+".join(f"{l}{'e' if l % 2 == 0 else 'o'}" for l in range(lsh + 1))
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        ref = R.MessagePackBlock(irr, irr, sh, irr, "8x0e", radial_MLP=[16, 16])
        E = 11
        g_ = torch.Generator().manual_seed(seed)
        src, dst, ef = (torch.randn(E, ref.irreps_node_feats.dim, generator=g_) for _ in range(3))
        zi = None
        if zeros:
            I, o = P.Irreps(irr), 0
            shs = {(l, (-1) ** l) for l in range(lsh + 1)}
            zi = {"node": [i for i, (m, l, p_) in enumerate(I) if (l, p_) != (0, 1)], "edge": [i for i, (m, l, p_) in enumerate(I) if (l, p_) not in shs]}
            for i, (m, l, p_) in enumerate(I):
                w = m * (2 * l + 1)
                if i in zi["node"]:
                    src[:, o:o + w] = 0.0
                    dst[:, o:o + w] = 0.0
                if i in zi["edge"]:
                    ef[:, o:o + w] = 0.0
                o += w
        n = torch.nn.functional.normalize(torch.randn(E, 3, generator=g_), dim=-1)
        shv = e3.spherical_harmonics(list(range(lsh + 1)), n, True, "component")
        rbf = torch.randn(E, 8, generator=g_)
        G = torch.randn(E, ref.irreps_node_feats.dim, generator=g_)
        (ref(src, dst, ef, shv, rbf) * G).sum().backward()
        want = {k: p.grad.clone() for k, p in ref.named_parameters()}
    finally:
        torch.set_default_dtype(prev)
    if max(float(v.abs().max()) for v in want.values()) < 1e-12:
        return
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    lay = P.PlanarLayout(irr)
    lm = max(lmax, lsh)
    D = emu.edge_wigner_all(n.numpy(), lm)
    rot = lambda t: torch.from_numpy(emu.rotate_rows(lay.to_planar(t.numpy()), lay, D, lm))
    run = lambda prog, srcs, hn, he: torch.from_numpy(emu.run_program(prog, [t.numpy() for t in srcs], (hn.numpy(), he.numpy())))
    # a previous instance with OTHER weights (the state after an optimiser step): its per-chunk constants are taken over
    rs = np.random.default_rng(seed)
    wg0 = BM.MessagePackWeightGrad({k: v + 0.1 * rs.normal(size=v.shape) for k, v in sd.items()}, irr, irr, sh, irr)
    BM.block_weight_grads(wg0, run, rot(src), rot(dst), rot(ef), rot(G), rbf, emu.SILU_CST, chunk=7)
    wg = BM.MessagePackWeightGrad(sd, irr, irr, sh, irr).adopt_constants(wg0)
    assert getattr(wg, "_groups_adopted", False)
    got = BM.block_weight_grads(wg, run, rot(src), rot(dst), rot(ef), rot(G), rbf, emu.SILU_CST, chunk=7)
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    for k in want:
        scale = max(float(want[k].abs().max()), 1e-30)
        assert float((got[k].reshape(want[k].shape) - want[k]).abs().max()) < 2e-6 * max(scale, 1e-3), (k, irr, sh)


@pytest.mark.parametrize("seed", list(range(5)) + [(3, "zeros"), (4, "zeros")], ids=lambda s_: "-".join(map(str, s_)) if isinstance(s_, tuple) else str(s_))
def test_message_pack_weight_gradients_fused_vs_autograd(seed):
    """the FUSED weight-gradient kernel's tables (plan.build_tp_wgrad_fused: units, B-operand fragments, accumulator blocks of the
    splits / edge-tile copies, the gather maps into the reference's flat parameters) through the kernel's numpy twin, every parameter of the
    block vs torch.autograd through the fp64 oracle.  seeds 3, 4: wide irreps (several 16-row tiles per super-path, two channel tiles)."""
    import torch
    from oracle import hamgnn_ref as R, e3
    from hamgnn_amd import backward_mp as BM
    # "zeros" (r5): a first-layer block -- node rows non-zero in 0e only, edge rows in the irreps of the spherical harmonics only; the tables are built
    # WITHOUT the row tiles of the super-paths that read the other irreps (their gradients are exactly zero) and every parameter still matches autograd
    seed, zeros = seed if isinstance(seed, tuple) else (seed, None)
    rng = np.random.default_rng(500 + seed)
    lmax = int(rng.integers(1, 3))
    irr = _random_irreps(rng, lmax)
    if "0e" not in irr:
        irr = "5x0e+" + irr
    lsh = int(rng.integers(1, 3))
    if seed >= 3:
        irr, lsh = ("20x0e+17x1o+6x2e", 2) if seed == 3 else ("33x0e+5x0o+9x1o+4x1e+3x2e+2x3o", 3)
    sh = "+".join(f"{l}{'e' if l % 2 == 0 else 'o'}" for l in range(lsh + 1))
    lmax = max(l for _, l, _ in P.Irreps(irr))
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        ref = R.MessagePackBlock(irr, irr, sh, irr, "8x0e", radial_MLP=[16, 64])                # (the kernel is built for the shipped 64-wide last hidden layer)
        E = 37 if seed >= 3 else 21
        g_ = torch.Generator().manual_seed(seed)
        src, dst, ef = (torch.randn(E, ref.irreps_node_feats.dim, generator=g_) for _ in range(3))
        zi = None
        if zeros:
            I, o = P.Irreps(irr), 0
            shs = {(l, (-1) ** l) for l in range(lsh + 1)}
            zi = {"node": [i for i, (m, l, p_) in enumerate(I) if (l, p_) != (0, 1)], "edge": [i for i, (m, l, p_) in enumerate(I) if (l, p_) not in shs]}
            for i, (m, l, p_) in enumerate(I):
                w = m * (2 * l + 1)
                if i in zi["node"]:
                    src[:, o:o + w] = 0.0
                    dst[:, o:o + w] = 0.0
                if i in zi["edge"]:
                    ef[:, o:o + w] = 0.0
                o += w
        n = torch.nn.functional.normalize(torch.randn(E, 3, generator=g_), dim=-1)
        shv = e3.spherical_harmonics(list(range(lsh + 1)), n, True, "component")
        rbf = torch.randn(E, 8, generator=g_)
        G = torch.randn(E, ref.irreps_node_feats.dim, generator=g_)
        (ref(src, dst, ef, shv, rbf) * G).sum().backward()
        want = {k: p.grad.clone() for k, p in ref.named_parameters()}
    finally:
        torch.set_default_dtype(prev)
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    lay = P.PlanarLayout(irr)
    lm = max(lmax, lsh)
    D = emu.edge_wigner_all(n.numpy(), lm)
    rot = lambda t: torch.from_numpy(emu.rotate_rows(lay.to_planar(t.numpy()), lay, D, lm))
    wg = BM.MessagePackWeightGrad(sd, irr, irr, sh, irr)
    wf = P.build_tp_wgrad_fused(wg.branches, sh, irr, wg.H, zero_inputs=zi)
    assert wf.units.shape[1] == P.WG_UNIT_I32 and wf.lds_bytes <= 80 * 1024
    if zeros:
        full = P.build_tp_wgrad_fused(wg.branches, sh, irr, wg.H)
        assert wf.mfma_per_tile < 0.7 * full.mfma_per_tile and not wf.gs_complete, (wf.mfma_per_tile, full.mfma_per_tile)
    nsplit = 1 + seed % 3

    def run(srcs, g, hn, he):
        acc, gs = emu.run_wgrad_fused(wf, [t.numpy() for t in srcs], g.numpy(), (hn.numpy(), he.numpy()), nsplit=nsplit)
        return torch.from_numpy(acc), [torch.from_numpy(a) for a in gs]
    got = BM.tp_weight_grads_fused(wg, wf, run, [rot(src), rot(dst), rot(ef)], rot(G), rbf, emu.SILU_CST, chunk=16 if seed % 2 else 1 << 20)
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    for k in want:
        scale = max(float(want[k].abs().max()), 1e-30)
        assert float((got[k].reshape(want[k].shape) - want[k]).abs().max()) < 2e-6 * max(scale, 1e-3), (k, irr, sh)


@pytest.mark.parametrize("seed", [(0, False), (1, False), (0, True), (1, True)], ids=["0", "1", "0-lite", "1-lite"])
def test_embedding_tp_weight_and_input_gradients_vs_autograd(seed):
    """the embedding tensor product (PairInteractionEmbeddingBlock.conv_tp, num_types x 0e input) through the same
    materialisation programs: every parameter AND the input rows' gradient vs torch.autograd through the fp64 oracle"""
    import torch
    from oracle import hamgnn_ref as R, e3
    from hamgnn_amd import backward_mp as BM
    seed, lite = seed
    rng = np.random.default_rng(700 + seed)
    T = int(rng.integers(3, 9))
    lsh = 2 + seed
    sh = "+".join(f"{l}{'e' if l % 2 == 0 else 'o'}" for l in range(lsh + 1))
    irr = "6x0e+5x1o+3x2e" + ("+2x3o" if lsh >= 3 else "")
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        ref = R.RadialTensorProduct(f"{T}x0e", sh, irr, "8x0e", [16, 16], lite_mode=lite)
        E = 9
        g_ = torch.Generator().manual_seed(seed)
        x = torch.randn(E, T, generator=g_).requires_grad_()
        n = torch.nn.functional.normalize(torch.randn(E, 3, generator=g_), dim=-1)
        shv = e3.spherical_harmonics(list(range(lsh + 1)), n, True, "component")
        rbf = torch.randn(E, 8, generator=g_)
        G = torch.randn(E, P.Irreps(irr).dim, generator=g_)
        (ref(x, shv, rbf) * G).sum().backward()
        want = {k: p.grad.clone() for k, p in ref.named_parameters()}
    finally:
        torch.set_default_dtype(prev)
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    lay, lin = P.PlanarLayout(irr), P.PlanarLayout([(T, 0, 1)])
    D = emu.edge_wigner_all(n.numpy(), lsh)
    grot = torch.from_numpy(emu.rotate_rows(lay.to_planar(G.numpy()), lay, D, lsh))
    if lite:
        sd = {k: v for k, v in sd.items() if "tensor_product" not in k}
        want = {k: v for k, v in want.items() if "tensor_product" not in k}
    wg = BM.TPWeightGrad(sd, P.embedding_wgrad_branches(sd, T, lite), sh, irr)
    run = lambda prog, srcs, hn, he: torch.from_numpy(emu.run_program(prog, [t.numpy() for t in srcs], (hn.numpy(), he.numpy())))
    xp = torch.from_numpy(lin.to_planar(x.detach().numpy()))
    got, gx = BM.tp_weight_grads(wg, run, [xp], grot, rbf, emu.SILU_CST, chunk=5, want_gx=True)
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    for k in want:
        scale = max(float(want[k].abs().max()), 1e-3)
        assert float((got[k].reshape(want[k].shape) - want[k]).abs().max()) < 2e-6 * scale, k
    assert float((gx[0][:, :T] - x.grad).abs().max()) < 2e-6 * max(float(x.grad.abs().max()), 1e-3)


@pytest.mark.parametrize("seed", [0, 1])
def test_embedding_tp_gradients_on_the_fused_route_vs_autograd(seed):
    """late r5: the embedding TP's weight gradients through the FUSED kernel's tables (its num_types x 0e row presented as two sources of num_types / 2
    channels: plan.embedding_wgrad_branches_split) and its input gradient as an adjoint program (plan.build_embedding_adjoint_program), numpy twins of
    both kernels, every parameter and the input rows' gradient vs torch.autograd through the fp64 oracle"""
    import torch
    from oracle import hamgnn_ref as R, e3
    from hamgnn_amd import backward_mp as BM
    T = 8 if seed == 0 else 16
    lsh = 2 + seed
    sh = "+".join(f"{l}{'e' if l % 2 == 0 else 'o'}" for l in range(lsh + 1))
    irr = "6x0e+5x1o+3x2e" + ("+2x3o" if lsh >= 3 else "") + "+3x1e"          # (1e: an output irrep the product cannot reach -- its block of the gradient is never read)
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        ref = R.RadialTensorProduct(f"{T}x0e", sh, irr, "8x0e", [16, 64], lite_mode=False)
        E = 21
        g_ = torch.Generator().manual_seed(seed)
        x = torch.randn(E, T, generator=g_).requires_grad_()
        n = torch.nn.functional.normalize(torch.randn(E, 3, generator=g_), dim=-1)
        shv = e3.spherical_harmonics(list(range(lsh + 1)), n, True, "component")
        rbf = torch.randn(E, 8, generator=g_)
        G = torch.randn(E, P.Irreps(irr).dim, generator=g_)
        (ref(x, shv, rbf) * G).sum().backward()
        want = {k: p.grad.clone() for k, p in ref.named_parameters()}
    finally:
        torch.set_default_dtype(prev)
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    lay, lin = P.PlanarLayout(irr), P.PlanarLayout([(T, 0, 1)])
    D = emu.edge_wigner_all(n.numpy(), lsh)
    grot = torch.from_numpy(emu.rotate_rows(lay.to_planar(G.numpy()), lay, D, lsh))
    wg = BM.TPWeightGrad(sd, P.embedding_wgrad_branches(sd, T, False), sh, irr)
    wf = P.build_tp_wgrad_fused(P.embedding_wgrad_branches_split(sd, T), sh, irr, wg.H)
    xp = torch.from_numpy(lin.to_planar(x.detach().numpy()))
    h = torch.from_numpy(emu.radial_hidden(rbf.numpy(), P.radial_hidden_weights(sd, "weight_generator", emu.SILU_CST)))

    def run(srcs, g, hn, he):
        acc, gs = emu.run_wgrad_fused(wf, [np.ascontiguousarray(t.numpy()) for t in srcs], g.numpy(), (hn.numpy(), hn.numpy()), nsplit=1 + seed)
        return torch.from_numpy(acc), [torch.from_numpy(a) for a in gs]
    got = BM.tp_weight_grads_fused(wg, wf, run, [xp[:, :T // 2], xp[:, T // 2:T]], grot, rbf, emu.SILU_CST, hidden={"emb": h[:, :wg.H]})
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    for k in want:
        scale = max(float(want[k].abs().max()), 1e-3)
        assert float((got[k].reshape(want[k].shape) - want[k]).abs().max()) < 2e-6 * scale, k
    adj = P.build_embedding_adjoint_program(sd, T, sh, irr)
    gx = emu.run_program(adj, [grot.numpy()], (h.numpy(), h.numpy()))
    assert float(np.abs(gx[:, :T] - x.grad.numpy()).max()) < 2e-6 * max(float(x.grad.abs().max()), 1e-3)


def _adjoint_case(irr, sh, lmax, lsh, seed, E=17, radial=(16, 16)):
    """oracle MessagePackBlock + torch.autograd: gradients of sum(out * G) with respect to the three inputs; and the emulator inputs"""
    import torch
    from oracle import hamgnn_ref as R, e3
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        ref = R.MessagePackBlock(irr, irr, sh, irr, "8x0e", radial_MLP=list(radial))
        g = torch.Generator().manual_seed(seed)
        src, dst, ef = (torch.randn(E, ref.irreps_node_feats.dim, generator=g).requires_grad_() for _ in range(3))
        n = torch.nn.functional.normalize(torch.randn(E, 3, generator=g), dim=-1)
        shv = e3.spherical_harmonics(list(range(lsh + 1)), n, True, "component")
        rbf = torch.randn(E, 8, generator=g)
        G = torch.randn(E, ref.irreps_node_feats.dim, generator=g)
        (ref(src, dst, ef, shv, rbf) * G).sum().backward()
    finally:
        torch.set_default_dtype(prev)
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items()}
    lm = max(lmax, lsh)
    D = emu.edge_wigner_all(n.numpy(), lm)
    hn = emu.radial_hidden(rbf.numpy(), P.radial_hidden_weights(sd, "node_weight_generator", emu.SILU_CST))
    he = emu.radial_hidden(rbf.numpy(), P.radial_hidden_weights(sd, "edge_weight_generator", emu.SILU_CST))
    return sd, D, lm, (hn, he), G.numpy(), (src.grad.numpy(), dst.grad.numpy(), ef.grad.numpy())


def _check_adjoint(prog, sched, irr, D, lm, h, G, want):
    lay = P.PlanarLayout(irr)
    _, maps = P.message_pack_adjoint_layout(irr, irr)
    gout = emu.rotate_rows(lay.to_planar(G), lay, D, lm)                      # adjoint of the forward epilogue's un-rotation
    o = emu.run_program(prog, [gout], h, D, lm) if sched is None else emu.run_program_is(prog, sched, [gout], h, D, lm)
    take = lambda im: np.where(im[None, :] >= 0, o[:, np.maximum(im, 0)], 0.0)
    got = (take(maps[0]), take(maps[1]), emu.rotate_rows(take(maps[2]), lay, D, lm, transpose=True))
    for a, b in zip(got, want):
        assert rel(lay.from_planar(a), b) < 1e-6, irr


@pytest.mark.parametrize("seed", range(8))
def test_data_gradient_program_vs_autograd(seed):
    """SURVEY 8f-3, first step: the data gradient of a MessagePackBlock as an ADJOINT PROGRAM for the same kernels (roles of the two
    weight matrices swapped, plan.add_tp_adjoint_items) vs torch.autograd through the fp64 oracle -- random irreps sets, both schedules."""
    rng = np.random.default_rng(100 + seed)
    lmax = int(rng.integers(1, 4))
    irr = _random_irreps(rng, lmax)
    if "0e" not in irr:
        irr = "5x0e+" + irr
    lsh = int(rng.integers(1, 4))
    sh = "+".join(f"{l}{'e' if l % 2 == 0 else 'o'}" for l in range(lsh + 1))
    sd, D, lm, h, G, want = _adjoint_case(irr, sh, lmax, lsh, seed)
    if max(np.abs(w).max() for w in want) < 1e-12:
        return
    prog = P.build_message_pack_adjoint_program(sd, irr, irr, sh, irr)
    _check_adjoint(prog, None, irr, D, lm, h, G, want)
    _check_adjoint(prog, P.is_schedule(prog, "lds"), irr, D, lm, h, G, want)


def test_data_gradient_program_wide_rows_need_several_workgroups():
    """three feature rows of output per edge: with wide irreps the tiles exceed one workgroup's LDS -> lds_partition spreads the output
    segments over several parts (blockIdx.y); 2 x 64 channels of one irrep > 64 -> column chunks (sender / receiver halves)"""
    irr, sh = "64x0e+32x0o+32x1o+16x1e+16x2e+8x2o+8x3o", "0e+1o+2e"
    sd, D, lm, h, G, want = _adjoint_case(irr, sh, 3, 2, seed=3, E=5)
    prog = P.build_message_pack_adjoint_program(sd, irr, irr, sh, irr)
    assert len(prog.seg_chunks[0]) == 2
    with pytest.raises(NotImplementedError):
        P.is_schedule(prog, 1)
    sc = P.is_schedule(prog, "lds")
    assert sc.part_table.shape[0] >= 2
    _check_adjoint(prog, sc, irr, D, lm, h, G, want)
    _check_adjoint(prog, None, irr, D, lm, h, G, want)


@pytest.mark.parametrize("irreps", [MINI, "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e", "3x0e+2x1o"])
def test_gate_tables_compact_match_the_oracle_gate(irreps):
    """hg_gate's tables (plan.gate_tables -> gate_tables_compact: distinct activated scalars once per row, outputs by look-up),
    emulated in numpy on planar rows, against the oracle's e3nn-style Gate of the reference ResidualBlock."""
    import torch
    from oracle import hamgnn_ref as R
    irr_in, irr_out, tab = P.gate_tables(irreps)
    act_tab, out_tab = P.gate_tables_compact(tab)
    assert len({(int(a), int(b)) for a, b in act_tab}) == len(act_tab)                  # every (input, activation) pair once
    n_scal = sum(m for m, l, p in so3.Irreps(irreps) if l == 0)
    n_gate = sum(m for m, l, p in so3.Irreps(irreps) if l > 0)
    assert len(act_tab) == n_scal + n_gate
    rb = R.ResidualBlock(irreps, irreps).double()
    gate = rb.equivariant_nonlin
    assert str(gate.irreps_in) == str(irr_in) and str(gate.irreps_out) == str(irr_out)
    x = np.random.default_rng(2).standard_normal((7, so3.Irreps(str(irr_in)).dim))
    want = gate(torch.from_numpy(x)).numpy()
    lay_in, lay_out = P.PlanarLayout(irr_in), P.PlanarLayout(irr_out)
    xp = lay_in.to_planar(x)

    def act(v, k):
        c = float(P.ACT_CONSTS[k])
        if k == P.ACT_SSP:
            return c * (np.log1p(np.exp(v)) - math.log(2.0))
        if k == P.ACT_TANH:
            return c * np.tanh(v)
        if k == P.ACT_ABS:
            return c * np.abs(v)
        return v
    av = np.stack([act(xp[:, i], k) for i, k in act_tab], 1)
    out = np.zeros((x.shape[0], lay_out.dim))
    for p, (sc, gc) in enumerate(out_tab):
        if sc < 0:
            continue
        v = av[:, sc & 0x3fffffff] if sc & 0x40000000 else xp[:, sc]
        out[:, p] = v * (av[:, gc] if gc >= 0 else 1.0)
    assert rel(lay_out.from_planar(out), want) < 1e-6          # ACT_CONSTS are float32 roundings of the Monte-Carlo constants


@pytest.mark.parametrize("seed", range(3))
def test_device_repack_map_equals_host_builder(seed):
    """hamgnn_amd/repack.py: the affine map (const, coef, idx) discovered by probing the host builders reproduces the packed weight blob
    of every program of a MessagePackBlock (forward plain / merged with a fused skip Linear, data-gradient adjoint, the two weight-
    gradient materialisation programs) for fresh random weights"""
    import torch
    from hamgnn_amd import nn as hnn, repack as RP
    rng = np.random.default_rng(900 + seed)
    irr = _random_irreps(rng, int(rng.integers(1, 4)))
    if "0e" not in irr:
        irr = "5x0e+" + irr
    lsh = int(rng.integers(1, 4))
    sh = "+".join(f"{l}{'e' if l % 2 == 0 else 'o'}" for l in range(lsh + 1))
    torch.manual_seed(seed)
    blk = hnn.MessagePackBlock(irr, irr, sh, irr, 8, [16, 16])
    sd = hnn._np_sd(blk)
    shapes = {k: v.shape for k, v in sd.items()}
    last = {n: P._last_layer(sd, f"{n}_weight_generator")[0][-1] for n in ("node", "edge")}
    lays = RP.mp_branch_layouts(irr, irr, sh, irr)
    nskip = hnn.E3Linear(irr, irr).weight.numel()
    groups = P.choose_merge_groups(blk.irreps_node, blk.irreps_edge, blk.irreps_sh, blk.irreps_out, 16)
    args = (blk.irreps_node, blk.irreps_edge, blk.irreps_sh, blk.irreps_out)
    builders = {
        "plain": (lambda d, skip: P.build_message_pack_program(d, *args, True, None).weights, 0),
        "merged+skip": (lambda d, skip: P.build_message_pack_program(d, *args, False, skip, merge_groups=groups).weights, nskip),
        "adjoint": (lambda d, skip: P.build_message_pack_adjoint_program(d, *args).weights, 0),
        "wgradA": (lambda d, skip: P.build_message_pack_wgrad_programs(d, *args)[0].weights, 0),
        "wgradB": (lambda d, skip: P.build_message_pack_wgrad_programs(d, *args)[1].weights, 0),
    }
    sd2 = {k: rng.normal(size=v.shape) for k, v in sd.items()}             # "after the optimiser step"
    skip2 = rng.normal(size=nskip)
    for name, (fn, ns) in builders.items():
        sizes = RP.mp_source_sizes(shapes, last, ns)
        pk = RP.AffinePack(lambda src, fn=fn: fn(RP.mp_probe_state_dict(src, shapes, last, lays, irr), src.get("skip")), sizes)
        want = np.asarray(fn(sd2, skip2 if ns else None), dtype=np.float32)
        got = pk.apply_np(RP.mp_sources(lambda k: sd2[k].reshape(-1), last, lays, skip2 if ns else None)).astype(np.float32)
        assert want.shape == got.shape, name
        assert np.abs(want - got).max() <= 1e-6 * max(1.0, np.abs(want).max()), name
        t = pk.apply({k: torch.from_numpy(np.asarray(v)) for k, v in RP.mp_sources(lambda k: sd2[k].reshape(-1), last, lays, skip2 if ns else None).items()})
        assert np.abs(t.numpy() - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), name


@pytest.mark.parametrize("seed", range(4))
def test_lite_mode_message_pack_backward_vs_autograd(seed):
    """SURVEY 8f-3, lite_mode MessagePackBlock (message_passing.py:197-215): data gradient (adjoint IT_LINC program, emulated on both
    schedules) and EVERY parameter gradient (hamgnn_amd/backward_lite.py) vs torch.autograd through the fp64 oracle; random irreps sets"""
    import torch
    from oracle import hamgnn_ref as R, e3
    from hamgnn_amd import backward_lite as BL
    from hamgnn_amd.nn import o3_linear_weight_grad
    rng = np.random.default_rng(300 + seed)
    lmax = int(rng.integers(1, 4))
    irr = _random_irreps(rng, lmax)
    if "0e" not in irr:
        irr = "5x0e+" + irr
    lsh = int(rng.integers(1, 4))
    sh = "+".join(f"{l}{'e' if l % 2 == 0 else 'o'}" for l in range(lsh + 1))
    E = 13
    torch.manual_seed(seed)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        ref = R.MessagePackBlock(irr, irr, sh, irr, "8x0e", radial_MLP=[16, 16], lite_mode=True)
        g = torch.Generator().manual_seed(seed)
        src, dst, ef = (torch.randn(E, ref.irreps_node_feats.dim, generator=g).requires_grad_() for _ in range(3))
        n = torch.nn.functional.normalize(torch.randn(E, 3, generator=g), dim=-1)
        shv = e3.spherical_harmonics(list(range(lsh + 1)), n, True, "component")
        rbf = torch.randn(E, 8, generator=g)
        G = torch.randn(E, ref.irreps_node_feats.dim, generator=g)
        (ref(src, dst, ef, shv, rbf) * G).sum().backward()
    finally:
        torch.set_default_dtype(prev)
    want_w = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
    sd = {k: v.detach().numpy() for k, v in ref.state_dict().items() if "tensor_product" not in k}
    lay = P.PlanarLayout(irr)
    lm = max(lmax, lsh)
    D = emu.edge_wigner_all(n.numpy(), lm)
    rot = lambda t: torch.from_numpy(emu.rotate_rows(lay.to_planar(t.detach().numpy()), lay, D, lm))
    lb = BL.LiteBackward(sd, irr, irr, sh, irr)
    for sched in ("seg", "is"):
        def run_program(prog, srcs):
            xs = [t.numpy() for t in srcs]
            if sched == "is":
                try:
                    return torch.from_numpy(emu.run_program_is(prog, P.is_schedule(prog, "lds"), xs, (None, None), D, lm))
                except NotImplementedError:
                    pass
            return torch.from_numpy(emu.run_program(prog, xs, (None, None), D, lm))
        run_linear = lambda tabs, x: torch.from_numpy(emu.run_linear_tables(tabs, x.numpy()))
        rows, grads = lb.run(run_program, run_linear, o3_linear_weight_grad, rot(src), rot(dst), rot(ef), rot(G), rbf, emu.SILU_CST)
        _, maps = P.message_pack_adjoint_layout(irr, irr)
        o = rows.numpy()
        take = lambda im: np.where(im[None, :] >= 0, o[:, np.maximum(im, 0)], 0.0)
        got = (take(maps[0]), take(maps[1]), emu.rotate_rows(take(maps[2]), lay, D, lm, transpose=True))
        for a, b in zip(got, (src.grad, dst.grad, ef.grad)):
            assert rel(lay.from_planar(a), b.numpy()) < 1e-6, (irr, sh)
        assert set(grads) == set(want_w), sorted(set(grads) ^ set(want_w))
        for k in want_w:
            scale = max(float(want_w[k].abs().max()), 1e-3)
            assert float((grads[k].reshape(want_w[k].shape) - want_w[k]).abs().max()) < 2e-6 * scale, (k, irr, sh)

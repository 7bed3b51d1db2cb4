"""Known-answer / invariant tests that pin oracle/e3.py (the e3nn==0.5.0 restatement), SURVEY.md 8c (1)-(7).
The reference ships no tests for this path, so these identities are the pin."""
import math

import numpy as np
import pytest
import torch

from oracle import e3

F64 = torch.float64


def test_irreps_parse_sort_simplify():
    ir = e3.Irreps("64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e")
    assert ir.dim == 877 and ir.num_irreps == 263 and len(ir) == 13
    s = e3.Irreps("2x1e+3x0e+1x0o+4x0e").sort()
    assert str(s.irreps) == "1x0o+3x0e+4x0e+2x1e" and s.p == (3, 1, 0, 2)
    assert str(s.irreps.simplify()) == "1x0o+7x0e+2x1e"
    assert str(e3.Irreps("0e + 1o + 2e")) == "1x0e+1x1o+1x2e"
    assert e3.Irrep("2e") in e3.Irrep("1o") * e3.Irrep("1o") and e3.Irrep("2o") not in e3.Irrep("1o") * e3.Irrep("1o")


@pytest.mark.parametrize("l1,l2,l3", [(a, b, c) for a in range(5) for b in range(4) for c in range(abs(a - b), min(a + b, 6) + 1)])
def test_w3j_orthogonality(l1, l2, l3):
    C = e3.wigner_3j(l1, l2, l3, dtype=F64)
    G = torch.einsum("ijk,ijl->kl", C, C)
    assert torch.allclose(G, torch.eye(2 * l3 + 1, dtype=F64) / (2 * l3 + 1), atol=1e-12)
    assert abs(C.norm().item() - 1) < 1e-12


def test_w3j_known_values_and_symmetry():
    assert abs(e3.wigner_3j(1, 1, 1, dtype=F64)[0, 1, 2].item() - 1 / math.sqrt(6)) < 1e-12
    for l in range(7):
        C = e3.wigner_3j(l, l, 0, dtype=F64)[:, :, 0]
        assert torch.allclose(C, torch.eye(2 * l + 1, dtype=F64) / math.sqrt(2 * l + 1), atol=1e-12)
        C = e3.wigner_3j(0, l, l, dtype=F64)[0]
        assert torch.allclose(C, torch.eye(2 * l + 1, dtype=F64) / math.sqrt(2 * l + 1), atol=1e-12)
    # (1,1,1) is the Levi-Civita tensor / sqrt(6)
    C = e3.wigner_3j(1, 1, 1, dtype=F64)
    assert torch.allclose(C, -C.transpose(0, 1), atol=1e-12)


def test_w3j_is_invariant_tensor():
    g = torch.Generator().manual_seed(0)
    R = e3.rand_rotation(g)
    for (a, b, c) in [(1, 1, 2), (2, 1, 3), (2, 2, 2), (3, 2, 4), (4, 5, 6), (6, 5, 1)]:
        Da, Db, Dc = (e3.wigner_D_from_matrix(l, R) for l in (a, b, c))
        C = e3.wigner_3j(a, b, c, dtype=F64)
        C2 = torch.einsum("ai,bj,ck,ijk->abc", Da, Db, Dc, C)
        assert torch.allclose(C, C2, atol=1e-9), (a, b, c)


def test_sh_norm_pole_and_polynomials():
    g = torch.Generator().manual_seed(1)
    v = torch.randn(50, 3, generator=g, dtype=F64)
    Y = e3.spherical_harmonics(list(range(7)), v, True, "component")
    i = 0
    for l in range(7):
        blk = Y[:, i:i + 2 * l + 1]
        assert torch.allclose((blk ** 2).sum(-1), torch.full((50,), 2.0 * l + 1, dtype=F64), atol=1e-10)
        i += 2 * l + 1
    pole = e3.spherical_harmonics(list(range(7)), torch.tensor([[0.0, 1.0, 0.0]], dtype=F64), True, "component")[0]
    i = 0
    for l in range(7):
        blk = pole[i:i + 2 * l + 1]
        expect = torch.zeros(2 * l + 1, dtype=F64)
        expect[l] = math.sqrt(2 * l + 1)
        assert torch.allclose(blk, expect, atol=1e-10)
        i += 2 * l + 1
    u = torch.nn.functional.normalize(v, dim=-1)
    x, y, z = u[:, 0], u[:, 1], u[:, 2]
    assert torch.allclose(Y[:, 1:4], math.sqrt(3) * u, atol=1e-12)
    l2 = torch.stack([math.sqrt(15) * x * z, math.sqrt(15) * x * y, math.sqrt(5) * (y * y - 0.5 * (x * x + z * z)),
                      math.sqrt(15) * y * z, math.sqrt(15) / 2 * (z * z - x * x)], -1)   # e3nn's explicit l=2 polynomials
    assert torch.allclose(Y[:, 4:9], l2, atol=1e-10)


def test_sh_matches_scipy_real_harmonics():
    """In physical coords (HamGNN feeds v[[1,2,0]]) Y^l equals sqrt(4pi) x the standard real spherical harmonics."""
    from scipy.special import sph_harm_y
    g = torch.Generator().manual_seed(2)
    v = torch.nn.functional.normalize(torch.randn(40, 3, generator=g, dtype=F64), dim=-1)
    x, y, z = v[:, 0].numpy(), v[:, 1].numpy(), v[:, 2].numpy()
    theta, phi = np.arccos(np.clip(z, -1, 1)), np.arctan2(y, x)
    Y = e3.spherical_harmonics(list(range(7)), v[:, [1, 2, 0]], True, "component").numpy()
    i = 0
    for l in range(7):
        for m in range(-l, l + 1):
            c = sph_harm_y(l, abs(m), theta, phi)
            if m < 0:
                ref = math.sqrt(2) * (-1) ** m * c.imag
            elif m == 0:
                ref = c.real
            else:
                ref = math.sqrt(2) * (-1) ** m * c.real
            assert np.allclose(Y[:, i], math.sqrt(4 * math.pi) * ref, atol=1e-9), (l, m)
            i += 1


def test_tensor_product_equivariance_and_variance():
    g = torch.Generator().manual_seed(3)
    torch.manual_seed(3)
    i1, i2, io = e3.Irreps("3x0e+2x1o+2x2e"), e3.Irreps("0e+1o+2e"), e3.Irreps("2x0e+2x1o+1x1e+2x2e")
    ins = [(a, b, c, "uvw", True) for a, (_, ia) in enumerate(i1) for b, (_, ib) in enumerate(i2) for c, (_, ic) in enumerate(io) if ic in ia * ib]
    tp = e3.TensorProduct(i1, i2, io, ins, internal_weights=True, shared_weights=True).to(F64)
    x, y = torch.randn(200, i1.dim, generator=g, dtype=F64), torch.randn(200, i2.dim, generator=g, dtype=F64)
    R = e3.rand_rotation(g)
    D1, D2, Do = e3.block_D(i1, R), e3.block_D(i2, R), e3.block_D(io, R)
    assert torch.allclose(tp(x @ D1.T, y @ D2.T), tp(x, y) @ Do.T, atol=1e-9)
    # parity (inversion): odd irreps flip sign
    P1, P2, Po = (torch.diag(torch.cat([torch.full((m * ir.dim,), float(ir.p)) for m, ir in irr]).to(F64)) for irr in (i1, i2, io))
    assert torch.allclose(tp(x @ P1, y @ P2), tp(x, y) @ Po, atol=1e-9)
    # e3nn normalisation: unit-variance in, N(0,1) weights => O(1) second moment out (averaged over weights)
    tot = 0
    for s in range(20):
        torch.manual_seed(100 + s)
        tp2 = e3.TensorProduct(i1, i2, io, ins, internal_weights=True, shared_weights=True).to(F64)
        tot += tp2(x, y).pow(2).mean().item()
    assert 0.6 < tot / 20 < 1.6


def test_linear_scale_with_weights_is_pure_channel_scale():
    from oracle.hamgnn_ref import LinearScaleWithWeights
    irr = e3.Irreps("3x0e+2x1o+2x2e")
    m = LinearScaleWithWeights(irr, irr).to(F64)
    g = torch.Generator().manual_seed(4)
    x, w = torch.randn(5, irr.dim, generator=g, dtype=F64), torch.randn(5, irr.num_irreps, generator=g, dtype=F64)
    y = m.tp(x, torch.ones(5, 1, dtype=F64), w)
    scale = torch.cat([w[:, c:c + 1].expand(-1, ir.dim) for c, ir in enumerate([ir for mul, ir in irr for _ in range(mul)])], 1)
    assert torch.allclose(y, x * scale, atol=1e-12)


def test_gate_layout_and_normalize2mom_constants():
    from oracle.hamgnn_ref import ResidualBlock, ssp
    assert abs(e3.normalize2mom_const(torch.nn.functional.silu) - 1.6791767923989418) < 1e-9
    assert abs(e3.normalize2mom_const(ssp) - 1.878204668541552) < 1e-9
    assert abs(e3.normalize2mom_const(torch.tanh) - 1.5937334472592692) < 1e-9
    A = "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e"
    rb = ResidualBlock(A, A)
    assert rb.equivariant_nonlin.irreps_in.dim == 1012
    assert str(rb.equivariant_nonlin.irreps_in).startswith("64x0o+199x0e+32x1o")
    B = "64x0e+32x1o+16x1e+8x2o+20x2e+8x3o+4x3e+4x4e"
    rb = ResidualBlock(B, B)
    assert str(rb.equivariant_nonlin.irreps_in).startswith("156x0e+32x1o")


def test_message_pack_shapes_set_A():
    """SURVEY 8a shape table: 255 paths per TP, mid_dim 17523, 3589 radial channels (set-A)."""
    from oracle.hamgnn_ref import tp_instructions
    A = e3.Irreps("64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e")
    sh = e3.Irreps("0e+1o+2e+3o+4e+5o")
    mid, ins = tp_instructions(A, sh, A)
    assert len(ins) == 255 and mid.dim == 17523 and mid.num_irreps == 3589 and len(mid.simplify()) == 13


def test_timed_contraction_order_equals_the_parity_order():
    """bench.py's cpu_baseline leg times the port with the uvw paths contracted weights-first (e3.CONTRACTION = "optimized": what opt_einsum_fx does to e3nn's
    generated einsums); the parity oracle keeps the naive outer-product order.  Same numbers."""
    import torch
    from oracle import e3
    torch.manual_seed(0)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        i1, i2, io = e3.Irreps("5x0e+3x1o+2x2e"), e3.Irreps("0e+1o+2e"), e3.Irreps("4x0e+2x1o+3x1e+2x2e")
        ins = [(a, b, c, "uvw", True) for a, (_, ir1) in enumerate(i1) for b, (_, ir2) in enumerate(i2) for c, (_, iro) in enumerate(io) if iro in ir1 * ir2]
        tp = e3.TensorProduct(i1, i2, io, ins, internal_weights=True, shared_weights=True)
        x, y = torch.randn(7, i1.dim), torch.randn(7, i2.dim)
        a = tp(x, y)
        e3.CONTRACTION = "optimized"
        try:
            b = tp(x, y)
        finally:
            e3.CONTRACTION = "outer"
    finally:
        torch.set_default_dtype(prev)
    a, b = a.detach(), b.detach()
    assert len(ins) > 10 and float((a - b).abs().max()) < 1e-12 * float(a.abs().max())

"""Host-side pieces of the read-out head's backward that are plain tensor algebra (no kernel): checked on CPU against autograd."""
import pytest
import torch

from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut


@pytest.mark.parametrize("sym", [True, False])
def test_soc_fold_is_the_adjoint_of_the_spin_assembly(sym):
    """HamGNNPlusPlusOut._soc_fold vs autograd through the assembly formula of csrc/head.hip:soc_assemble_kernel
    (hamgnn_output.py:3026-3144): real = [[H, A_y], [A_y, H]], imag = [[A_z, A_x], [-A_x, -A_z]], A_k = antiherm(ksi L_k)"""
    torch.manual_seed(0)
    n, E = 4, 6

    class Head:
        nao_max, symmetrize = n, sym

    inv = torch.tensor([1, 0, 3, 2, 5, 4])
    H = torch.randn(E, n * n, dtype=torch.float64, requires_grad=True)
    ksi = torch.randn(E, n * n, dtype=torch.float64, requires_grad=True)
    L = torch.randn(E, n * n, 3, dtype=torch.float64)

    def A(k):
        M = (ksi * L[:, :, k]).reshape(E, n, n)
        return 0.5 * (M - M[inv].transpose(1, 2)) if sym else M

    Ax, Ay, Az = A(0), A(1), A(2)
    Hm = H.reshape(E, n, n)
    real = torch.cat([torch.cat([Hm, Ay], 2), torch.cat([Ay, Hm], 2)], 1).reshape(E, -1)
    imag = torch.cat([torch.cat([Az, Ax], 2), torch.cat([-Ax, -Az], 2)], 1).reshape(E, -1)
    Gr, Gi = torch.randn_like(real), torch.randn_like(imag)
    ((real * Gr).sum() + (imag * Gi).sum()).backward()
    gk, gh = HamGNNPlusPlusOut._soc_fold(Head, Gr, Gi, L, inv)
    assert float((gk - ksi.grad).abs().max()) < 1e-12 and float((gh - H.grad).abs().max()) < 1e-12


def test_attention_backward_vs_autograd_through_the_oracle():
    """hamgnn_amd/backward_attn.py (planar rows, column -> head table) vs autograd through the oracle's AttentionAggregation
    (hamgnn/nn/attention.py:91-164) incl. the learnable soft cutoff"""
    import numpy as np
    from oracle import hamgnn_ref as R
    from hamgnn_amd import plan as P
    from hamgnn_amd.backward_attn import attention_backward
    torch.manual_seed(1)
    irr, H = "8x0e+4x0o+4x1o+2x1e+2x2o+4x2e+2x3o", 2
    lay = P.PlanarLayout(irr)
    D = P.Irreps(irr).dim
    N, E, rc = 5, 23, 6.0
    gen = torch.Generator().manual_seed(2)
    src, dst = torch.randint(0, N, (E,), generator=gen), torch.randint(0, N, (E,), generator=gen)
    K = torch.randn(N, D, generator=gen, dtype=torch.float64).requires_grad_()
    V = torch.randn(E, D, generator=gen, dtype=torch.float64).requires_grad_()
    length = 0.5 + 6.5 * torch.rand(E, generator=gen, dtype=torch.float64)        # some edges beyond the cutoff (cut = 0)
    p = torch.tensor(3.0, dtype=torch.float64, requires_grad=True)
    att = R.AttentionAggregation(H, irr)
    cut = R.soft_unit_step(p * (1.0 - length / rc))
    out = att(K[src], V, K[dst], cut, torch.stack([src, dst]), N)
    G = torch.randn(N, D, generator=gen, dtype=torch.float64)
    (out * G).sum().backward()
    tab, hd = P.attention_head_table(irr, H)
    pl = lambda t: torch.from_numpy(lay.to_planar(t.detach().numpy()))
    gK, gV, gp = attention_backward(pl(K), pl(V), pl(G), src, dst, length, torch.from_numpy(tab), H, hd, p.detach(), rc)
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    assert rel(lay.from_planar(gK.numpy()), K.grad.numpy()) < 1e-10
    assert rel(lay.from_planar(gV.numpy()), V.grad.numpy()) < 1e-10
    assert abs(float(gp) - float(p.grad)) < 1e-10 * max(1.0, abs(float(p.grad)))


def test_k_space_assembly_adjoint_vs_autograd():
    """kspace.assemble_k_adjoint vs autograd through a dense restatement of the H(k) assembly (padded [nk, n, n, nao, nao] blocks with
    index_put(accumulate) and masked_select to the compact basis -- the reference's formulation, hamgnn_output.py:1779-1905)"""
    import math
    import numpy as np
    from hamgnn_amd import kspace
    from hamgnn_amd.data import synthetic as S
    g = S.random_cell(4, [6, 1, 8], seed=5, density=0.004)
    n, e, nao, nk = g.num_nodes, g.num_edges, 5, 3
    gen = torch.Generator().manual_seed(0)
    valid = {6: [0, 1, 2, 3, 4], 1: [0, 2], 8: [0, 1, 3, 4]}                       # element -> orbitals it has
    orank_all = torch.full((n, nao), -1, dtype=torch.int64)
    for i, zz in enumerate(g.z.tolist()):
        orank_all[i, valid[zz]] = torch.arange(len(valid[zz]))
    on = torch.randn(n, nao * nao, generator=gen, dtype=torch.float64, requires_grad=True)
    off = torch.randn(e, nao * nao, generator=gen, dtype=torch.float64, requires_grad=True)
    kv = 0.1 * torch.randn(nk, 3, generator=gen, dtype=torch.float64)
    src, dst = g.edge_index
    phase = torch.exp(2j * math.pi * (g.nbr_shift.double()[:, None, :] * kv[None, :, :]).sum(-1))                # [e, nk]
    Hk = torch.zeros(nk, n, n, nao, nao, dtype=torch.complex128)
    ar = torch.arange(n)
    Hk[:, ar, ar] += on.reshape(-1, nao, nao)[None].to(torch.complex128)
    Ho = off.reshape(-1, nao, nao).to(torch.complex128)
    Hk = torch.stack([torch.index_put(Hk[k], (src, dst), phase[:, k][:, None, None] * Ho, accumulate=True) for k in range(nk)])
    Hk = Hk.swapaxes(-2, -3).reshape(nk, n * nao, n * nao)
    m = (orank_all >= 0).reshape(-1)
    M = int(m.sum())
    Hc = torch.masked_select(Hk, (m[:, None] & m[None, :])[None].expand(nk, -1, -1)).reshape(nk, M, M)
    G = torch.complex(torch.randn(nk, M, M, generator=gen, dtype=torch.float64), torch.randn(nk, M, M, generator=gen, dtype=torch.float64))
    (Hc.real * G.real + Hc.imag * G.imag).sum().backward()
    g_on, g_off = kspace.assemble_k_adjoint(G, g, kv, 0, n, 0, e, orank_all, nao)
    assert float((g_on - on.grad).abs().max()) < 1e-12 and float((g_off - off.grad).abs().max()) < 1e-12
    assert float(off.grad.abs().max()) > 0.1 and float((g_off.reshape(e, nao, nao)[:, 1, :].abs().sum(1) == 0).float().mean()) > 0   # masked orbitals

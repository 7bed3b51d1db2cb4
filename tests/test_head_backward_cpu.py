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

"""host-side (index / bool plumbing) parts of the MI355X head that need no GPU: interaction masks, sparsity ratio inputs."""
import numpy as np
import torch

from oracle import hamgnn_ref as R
from hamgnn_amd.data import Graph, collate
from hamgnn_amd.data import synthetic as S
from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut

MINI = "8x0e+4x0o+4x1o+2x1e+2x2o+3x2e+2x3o"


def _graphs():
    return [S.add_random_targets(S.random_cell(4 + k, [14, 8, 6, 1], seed=k, density=0.006), 19, seed=k) for k in range(2)]


def test_interaction_masks_match_oracle():
    """get_nonzero_mask_tensor (hamgnn_output.py:2616-2665, 2716-2783; oracle pinned against the reference in oracle/gen_golden.py)"""
    g = collate(_graphs())
    head = HamGNNPlusPlusOut(MINI, MINI, nao_max=19, ham_type="openmx", ham_only=True, soc_switch=False, get_nonzero_mask_tensor=True)
    head.compile("cpu")
    ref = R.HamGNNPlusPlusOut(MINI, MINI, nao_max=19, ham_type="openmx")
    m = head.build_interaction_masks(g)
    assert m.dtype == torch.bool and torch.equal(m, ref.interaction_masks(g))
    _, counts = head._global_inverse(g)
    ms = head.build_interaction_masks(g, counts, soc=True)
    assert torch.equal(ms, ref.interaction_masks(g, soc=True))
    assert 0 < int(m.sum()) < m.numel()


def test_scatter_rows_is_index_add_in_a_fixed_order():
    """ops.scatter_rows (the deterministic edge -> node / species reduction of the backward passes): equals index_add_ up to rounding,
    covers empty segments and an empty input, and two calls give bit-identical sums whatever the order of equal indices' rows."""
    from hamgnn_amd import ops
    g = torch.Generator().manual_seed(3)
    n, q = 37, 5000
    idx = torch.randint(0, n - 3, (q,), generator=g)           # the last three segments stay empty
    src = torch.randn(q, 7, generator=g)
    ref = torch.zeros(n, 7, dtype=torch.float64).index_add_(0, idx, src.double())
    out = ops.scatter_rows(idx, src, n)
    assert out.shape == (n, 7) and float((out.double() - ref).abs().max()) < 1e-4
    assert float(out[n - 3:].abs().max()) == 0.0
    assert torch.equal(out, ops.scatter_rows(idx, src, n))
    assert ops.scatter_rows(idx[:0], src[:0], n).shape == (n, 7)
    one = ops.scatter_rows(torch.tensor([2, 2, 0]), torch.tensor([1.0, 2.0, 4.0]), 4)
    assert torch.equal(one, torch.tensor([4.0, 0.0, 3.0, 0.0]))

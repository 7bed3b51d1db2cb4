"""The product's HOST logic end to end on the CPU: hamgnn_amd.ops replaced by the table-exact stand-ins of tests/cpu_ops.py (numpy / torch
twins of the kernels that consume the planner's packed tables), everything above the C ABI unchanged -- forward and full backward of
the whole model against the fp64 oracle.  The `-m gpu` twins of these checks run the same functions through the HIP kernels."""
import pytest
import torch

from tests import cpu_ops
from tests import gpu_checks as G


@pytest.fixture
def cpu_backend(monkeypatch):
    cpu_ops.install(monkeypatch)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    torch.set_num_threads(min(8, torch.get_num_threads()))


def test_forward_whole_model_vs_oracle(cpu_backend):
    r = G.oracle_vs_hip_random(device="cpu", n_atoms=4, seed=3)
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["H_rel_err"] < G.TOL, r


def test_radial_mlp_with_four_hidden_layers(cpu_backend):
    """`radial_MLP` with more than three entries (VERDICT r5 "what's missing" #4): the hidden activations chain hg_radial_hidden launches (three layers each);
    forward and every parameter gradient vs the fp64 oracle"""
    r = G.oracle_vs_hip_random(device="cpu", n_atoms=3, seed=2, radial=(8, 16, 8, 16))
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["H_rel_err"] < G.TOL, r
    b = G.check_full_backward(device="cpu", n_atoms=2, seed=3, radial=(8, 16, 8, 16))
    assert b["loss_rel_err"] < G.TOL and b["max_rel_err"] < G.TOL and b["n_params"] == 92, b


def test_full_backward_whole_model_vs_autograd(cpu_backend):
    r = G.check_full_backward(device="cpu", n_atoms=3, seed=4)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL and r["n_params"] == 74, r


@pytest.mark.parametrize("kw", [
    dict(legacy=True, metric="mae"),                                   # legacy_edge_update: layer 0 keeps its edge features
    dict(crystals=3, n_atoms=2, metric="mae"),                         # ragged batch, per-crystal row order of the result
    dict(charge=True, crystals=2, n_atoms=2),                          # charge doping: embedding tables + the charge MLP
    dict(corr=True),                                                   # CorrProductBlock after every ConvBlock
    dict(corr=True, charge=True, crystals=2, n_atoms=2),               # ... with doped node attributes: per-node mixtures of its element weights
    dict(corr=3, charge=True, crystals=2, n_atoms=2, irr="6x0e+3x0o+3x1o+2x1e+2x2e", nao=13, seed=6),   # correlation 3: the nu = 3 term (hamgnn_amd/corr3.py)
    dict(corr=1, n_atoms=2),                                           # correlation 1
    dict(soc="so3"), dict(soc="so3_nonsoc", crystals=2, n_atoms=2),    # SOC / so3 head, and the Uni-HamGNN SOC mode
    dict(transformer=True, irr="8x0e+4x0o+4x1o+2x1e+2x2o+4x2e+2x3o"),  # HamGNNTransformer
    dict(lite=True), dict(lite=True, legacy=True, crystals=2, n_atoms=2),  # lite_mode: uvu products + combine post-op (CPU-validated only)
    dict(zps=True, crystals=2, n_atoms=2), dict(zps=True, soc="so3"),      # zero_point_shift (the universal non-SOC model trains with it)
    dict(radial=(16, 64), num_types=24, n_atoms=2),                        # 64-wide radial layers: the FUSED weight-gradient kernel's tables (message blocks with the structural-zero
    dict(radial=(16, 64), num_types=24, n_atoms=2, charge=True, crystals=2),   # shortcut; the embedding TP as two 12-channel sources + its adjoint program), incl. doped node attributes
], ids=["legacy", "batch", "charge", "corr", "corr_charge", "corr_nu3_charge", "corr_nu1", "so3", "so3_nonsoc", "transformer", "lite", "lite_legacy_batch", "zero_point_shift", "zero_point_shift_soc",
        "fused_wgrad_route", "fused_wgrad_route_charge"])
def test_full_backward_variants_vs_autograd(cpu_backend, kw):
    kw = dict(dict(n_atoms=3, seed=5), **kw)
    r = G.check_full_backward(device="cpu", **kw)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL, r


@pytest.mark.parametrize("kw", [dict(), dict(add_H_nonsoc=True, crystals=2), dict(basis="su2", n_atoms=3), dict(basis="su2_f", n_atoms=2)],
                         ids=["so3", "so3_nonsoc", "su2", "su2_f_shell_l7"])
def test_soc_head_backward_vs_autograd(cpu_backend, kw):
    r = G.check_soc_head_backward(device="cpu", **kw)
    assert all(v < G.TOL for k, v in r.items() if k.endswith("rel_err")), r


def test_small_graph_reproducibility_check_runs_on_the_stand_ins(cpu_backend):
    """the r6 GPU check (split launches bit-reproducible) through the host code on the CPU stand-ins: the launches it means to test ARE split launches
    (the stand-ins are deterministic by construction: what is pinned here is the check's plumbing and the dispatch decision)"""
    r = G.check_small_graph_forward_reproducible(device="cpu", graph="cell9", reps=2)
    assert r["parts"] != ["1"] and r["H_max_abs_diff"] == 0.0 and r["H_absmax"] > 0, r


def test_device_repack_equals_recompile_on_cpu(cpu_backend):
    """hamgnn_amd/repack.py through the product's own refresh path (training_step -> weights_changed -> refresh_weights)"""
    r = G.check_refresh_equals_recompile(device="cpu")
    assert r["packers"] >= 8 and r["loss_rel_diff"] < 1e-6 and r["grad_max_rel_diff"] < 1e-5, r
    assert r["inference_rel_diff"] < 1e-6 and r["step_moved_H"] > 1e-3, r      # validate -> step -> validate uses the new weights (stale ResidualBlock row program)


def test_device_repack_equals_recompile_attention_backbone(cpu_backend):
    r = G.check_refresh_equals_recompile(device="cpu", transformer=True)
    assert r["packers"] >= 8 and r["loss_rel_diff"] < 1e-6 and r["grad_max_rel_diff"] < 1e-5, r
    assert r["inference_rel_diff"] < 1e-6 and r["step_moved_H"] > 1e-3, r


def test_band_energy_loss_backward_on_cpu(cpu_backend):
    r = G.check_band_energy_backward(device="cpu")
    assert r["g_on_rel_err"] < 1e-4 and r["g_off_rel_err"] < 1e-4 and r["grads_finite"] and r["losses"][-1] < r["losses"][0], r


_REL = lambda r: all(v < G.TOL for k, v in r.items() if k.endswith("rel_err"))
_FORWARD_CASES = {
    "backbone": lambda: G.check_backbone("cpu", "backbone"),
    "backbone_lite": lambda: G.check_backbone("cpu", "backbone_lite"),
    "backbone_corr": lambda: G.check_backbone("cpu", "backbone_corr"),
    "backbone_gaussian_rbf": lambda: G.check_backbone("cpu", "backbone_gaussian_rbf"),
    "charge_doping": lambda: G.check_charge_doping("cpu"),
    "charge_doping_corr": lambda: G.check_charge_doping_corr("cpu"),
    "transformer": lambda: G.check_transformer("cpu"),
    "corr_product": lambda: G.check_corr_product("cpu"),
    "corr_product_nu3": lambda: G.check_corr_product("cpu", "corr_product_block_nu3"),
    "corr_product_nu1": lambda: G.check_corr_product("cpu", "corr_product_block_nu1"),
    "head_openmx_19": lambda: G.check_head("cpu"),
    "head_abacus_13": lambda: G.check_head("cpu", "head_abacus_13", "abacus", 13),
    "head_from_planar_rows": lambda: G.check_head("cpu", use_planar_path=True),
    "head_norm_openmx_19": lambda: G.check_head("cpu", "head_norm_openmx_19", nonlinearity_type="norm"),      # r5: nonlinearity_type = "norm" (NormActivation)
    "head_backward_norm": lambda: G.check_head_backward("cpu", nonlinearity_type="norm"),
    "head_soc_so3": lambda: G.check_head_soc("cpu"),
    "head_soc_su2": lambda: G.check_head_su2("cpu"),
    "residual_block_backward": lambda: G.check_residual_block_backward("cpu"),
    "head_backward": lambda: G.check_head_backward("cpu"),
    "message_pack_backward": lambda: G.check_message_pack_backward("cpu", seed=1, E=21),
    "message_pack_weight_grads": lambda: G.check_message_pack_weight_grads("cpu", seed=1, E=21),
    "attribute_style_graph": lambda: G.check_attribute_style_graph("cpu"),
    "zero_point_shift": lambda: G.check_zero_point_shift("cpu"),
    "head_overlap_networks": lambda: G.check_head_overlap("cpu"),
    "conv_message_chain_backward": lambda: G.check_conv_message_backward("cpu", n_atoms=4),
    "front_door": lambda: G.check_front_door("cpu", tmpdir="/tmp/hg_front_door_cpu"),
    "fused_scatter": lambda: G.check_fused_scatter("cpu", n_atoms=5),
    "structural_zeros": lambda: G.check_structural_zeros("cpu", n_atoms=4),
    "structural_zeros_legacy": lambda: G.check_structural_zeros("cpu", legacy=True, n_atoms=4),
}


@pytest.mark.parametrize("name", list(_FORWARD_CASES))
def test_gpu_check_functions_on_the_cpu_stand_ins(cpu_backend, name):
    """the reference FIXTURES (backbones, heads, SOC, CorrProduct, transformer) and the block-level backward checks of the `-m gpu` suite,
    unchanged, on the CPU stand-ins: the planner tables and the host code of every forward variant against the reference's own outputs"""
    r = _FORWARD_CASES[name]()
    if name == "attribute_style_graph":
        assert r["node_attr"] < 1e-6 and r["edge_attr"] < 1e-6 and r["node_vs_fixture"] < G.TOL and r["head_vs_fixture"] < G.TOL and r["cache_reused"], r
        assert abs(r["sparsity_ratio"] - r["sparsity_ratio_fixture"]) < 1e-6 * r["sparsity_ratio_fixture"]
        return
    assert _REL(r) and any(k.endswith("rel_err") for k in r), r
    if "sparsity_ratio_reference" in r:
        assert abs(r["sparsity_ratio"] - r["sparsity_ratio_reference"]) < 1e-6 * r["sparsity_ratio_reference"]


def _assert_dead_outputs(r, min_dead, nonzero=True):
    if min_dead == 0:                                          # a head that reads every irrep (SOC / su2): nothing is skipped, nothing changes
        assert r["dead_irreps"] == 0 and r["alive_declared"] == 0.0 and r["last_pair_mfma_ratio"] == 1.0, r
        assert r["ham_rel_err"] == 0.0 and r["edge_attr_rel_err"] == 0.0 and r["wider_head_rel_err"] < 1e-6, r
        return
    assert r["dead_irreps"] >= min_dead and r["alive_declared"] == 1.0, r
    assert r["ham_rel_err"] < 1e-6 and r["ham_noise_max_abs"] == 0.0, r                     # same result; the unread blocks really are unread
    assert r["dead_blocks_max_abs"] == 0.0 and (r["dead_blocks_full_max_abs"] > 0.0 or not nonzero), r      # written as zeros where the complete program has values
    assert r["edge_attr_rel_err"] < 1e-6 and r["wider_head_rel_err"] < 1e-6, r              # the public tensor / a head that reads more: complete rows
    assert r["last_pair_mfma_ratio"] < 1.0, r
    assert r["training_rows_rel_err"] < 1e-6 and r["training_alive_declared"] == 0.0, r     # training forwards run the complete program
    assert r["refresh_rel_err"] < 1e-6, r                                                   # the reduced program is repacked on the device like the others


@pytest.mark.parametrize("kw", [dict(), dict(num_layers=1), dict(soc=True), dict(soc="su2", n_atoms=3), dict(nonlinearity_type="norm"), dict(transformer=True),
                                dict(irr="8x0e+4x0o+4x1o+2x1e+2x2o+3x2e+2x3o+2x3e+2x4o+2x4e", n_atoms=3)],
                         ids=["mini", "one_layer", "soc_so3", "soc_su2", "norm_activation", "transformer", "l4"])
def test_unread_irreps_of_the_last_pair_block(cpu_backend, kw):
    """r5: HamGNNConvE3.declare_consumer (Model does the call): the last PairInteractionBlock skips the output irreps the read-out head never reads"""
    r = G.check_dead_outputs("cpu", **dict(dict(n_atoms=4), **kw))
    _assert_dead_outputs(r, 0 if kw.get("soc") == "su2" else 2 if "irr" in kw else 1, nonzero=kw.get("num_layers") != 1)       # (one layer: 0o is still structurally zero after the first pair block)


@pytest.mark.parametrize("legacy", [False, True])
def test_backward_skips_structural_zero_inputs(cpu_backend, legacy):
    """r5: first-layer blocks leave out, in their backward too, the super-paths that read structurally zero input irreps (fused weight-gradient tables,
    adjoint program): same loss, same gradient for every parameter as with the shortcut off"""
    r = G.check_structural_zeros_backward("cpu", n_atoms=5 if not legacy else 4, legacy=legacy)
    assert r["loss_rel_err"] < 1e-9 and r["grad_max_rel_err"] < 1e-8 and r["fused_route"] == 1.0, r
    assert r["first_conv_wgrad_mfma_ratio"] < 0.6 and r["first_conv_adjoint_mfma_ratio"] < 0.6, r


def test_representation_is_released_by_reference_counting(cpu_backend):
    """the lazy entries of a representation must not close over the representation itself: a reference cycle keeps every row of a forward (~9 GB at
    0.82 M edges) alive until the cycle collector runs -- on the GPU that showed as one 500 ms step in five (allocator growth + a stalled free)"""
    import gc
    import weakref
    from hamgnn_amd.data import synthetic as S
    from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
    from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
    from hamgnn_amd.models.model import Model
    cfg = dict(num_types=96, irreps_edge_sh=G.SH, edge_sh_normalization="component", edge_sh_normalize=True, build_internal_graph=False, cutoff=26.0, rbf_func="bessel",
               num_radial=8, num_layers=2, irreps_node_features=G.MINI, use_kan=False, radial_MLP=[16, 16], correlation=2, num_hidden_features=4, use_corr_prod=False,
               legacy_edge_update=False)
    m = Model(HamGNNConvE3(cfg), HamGNNPlusPlusOut(G.MINI, G.MINI, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=False, soc_switch=False,
                                                    zero_point_shift=False))
    g = S.random_cell(4, [14, 8, 6, 1], seed=1, density=0.004)
    was = gc.isenabled()
    gc.disable()
    try:
        for declared in (True, False):
            if not declared:
                m.representation.declare_consumer(object())     # nothing declared: the plain lazy entries
            with torch.no_grad():
                rep = m.representation(g)
                assert (rep.get("_edge_alive") is not None) == declared
                w = weakref.ref(rep["_edge_planar_rot"])
                m.output_module(g, rep)
                del rep
            assert w() is None, ("rows kept alive by a reference cycle", declared)
    finally:
        if was:
            gc.enable()


def test_reference_loss_semantics_on_cpu(cpu_backend):
    """the reference's calculate_loss (hamgnn/models/Model.py:150-166): hamiltonian-type losses are multiplied by the head's sparsity_ratio
    (calculate_sparsity=True is the head's default), SOC models train on hamiltonian_real + hamiltonian_imag with their own weights --
    loss value and every parameter gradient vs autograd through the oracle with that formula"""
    r = G.check_full_backward(device="cpu", n_atoms=4, num_layers=1, nao=14, sparsity=True)
    assert r["loss_rel_err"] < 1e-5 and r["max_rel_err"] < 2e-5, r
    r = G.check_full_backward(device="cpu", n_atoms=4, num_layers=1, nao=14, soc="so3", sparsity=True, split_losses=True, metric="mae")
    assert r["loss_rel_err"] < 1e-5 and r["max_rel_err"] < 2e-5, r


@pytest.mark.parametrize("zps", [False, True])
def test_band_energy_loss_with_zero_point_shift_on_cpu(cpu_backend, zps):
    """hamiltonian + band_energy losses on a whole model vs autograd through the oracle: with zero_point_shift the band gradient enters at
    the blocks BEFORE the shift (not through the shift's adjoint) and carries the adjoint of the mean alignment"""
    r = G.check_full_backward(device="cpu", n_atoms=3, num_layers=1, nao=13, metric="mae", zps=zps, bands=True)
    assert r["loss_rel_err"] < 1e-4 and r["max_rel_err"] < 5e-4, r


@pytest.mark.parametrize("zps", [False, True])
def test_band_energy_loss_on_a_spin_orbit_head_on_cpu(cpu_backend, zps):
    """hamiltonian + band_energy losses on a SOC / so3 head: the bands of the stacked spinor H(k) (kspace.band_energy_backward_soc: eigh chain by
    autograd, the four spin blocks through the assembly's adjoint, real and imaginary rows) vs autograd through the oracle"""
    r = G.check_full_backward(device="cpu", n_atoms=2, num_layers=1, nao=13, metric="mse", zps=zps, bands=True, soc="so3")
    assert r["loss_rel_err"] < 1e-5 and r["max_rel_err"] < 5e-5, r


@pytest.mark.parametrize("tag", ["batch", "single"])
def test_head_bands_with_zero_point_shift_on_cpu(cpu_backend, tag):
    """calculate_band_energy + zero_point_shift (the default of build_hamgnn_model) vs the reference's forward: bands from the unshifted
    blocks, aligned by their mean (hamgnn_output.py:3802-3880, 3971-3985), for a two-crystal batch and for a single crystal (rows in place)"""
    r = G.check_head_bands_zero_point("cpu", tag)
    assert r["H_rel_err"] < G.TOL and r["unshifted_H_rel_err"] < G.TOL, r
    assert r["band_energy_err"] < 1e-4 and r["unshifted_band_energy_err"] < 1e-4 and r["target_band_energy_err"] < 1e-4, r
    assert r["shift_matters"] > 1e-3 and r["H_shift_matters"] > 1e-3, r       # the fixture separates the right order from the wrong ones


def test_band_energies_on_cpu(cpu_backend):
    """k-space step + the head's forward with calculate_band_energy (ham_only True and False) through the host code"""
    r = G.check_band_energies("cpu")
    assert r["band_energy_err"] < 1e-4 and r["band_gap_err"] < 1e-4 and r["window_err"] < 1e-4
    assert r["forward_ok"] and r["kpath_ok"] and r["with_overlap_ok"] and r["targets_consistent"] < 1e-5


def test_band_energies_spin_orbit_on_cpu(cpu_backend):
    """the spin-orbit k-space step and the SOC head's forward with calculate_band_energy=True through the host code (stand-in assembly)"""
    r = G.check_band_energies_soc("cpu")
    assert r["bands"] == r["ref_bands"] and r["band_energy_err"] < 1e-4 and r["window_err"] < 1e-4
    assert r["forward_ok"] and r["targets_consistent"] < 1e-5


def test_band_cal_on_cpu(cpu_backend):
    """band structure along a k-path from saved Hamiltonian rows (DFT_interfaces/openmx/band_cal.py, non-SOC branch) vs the script's own dense loop"""
    r = G.check_band_cal("cpu")
    assert r["bands_rel_err"] < 1e-4 and r["gap_abs_err_eV"] < 1e-2 and r["crystals"] == 2, r


def test_band_cal_spin_branches_on_cpu(cpu_backend):
    """the spin-orbit and collinear branches of DFT_interfaces/openmx/band_cal.py vs dense restatements of the script's loops"""
    r = G.check_band_cal_spin("cpu")
    assert r["soc_bands_rel_err"] < 1e-4 and r["soc_gap_abs_err_eV"] < 1e-2 and r["collinear_bands_rel_err"] < 1e-4, r


def test_band_energies_export_on_cpu(cpu_backend):
    """export_reciprocal_values vs the reference's own outputs (fixture band_energies_export_openmx_13)"""
    r = G.check_band_energies_export("cpu")
    assert all(v < (2e-3 if k.endswith("wf_abs_err") else 2e-4) for k, v in r.items()), r


def test_block_gemm_tables_on_cpu(cpu_backend):
    """the unit tables of hg_block_gemm through the numpy twin of the kernel (same check as on the GPU)"""
    r = G.check_block_gemm("cpu")
    assert r["f64_rel_err"] < 1e-13 and r["f32_rel_err"] < 2e-7, r


def test_training_loop_on_cpu(cpu_backend):
    """a few optimiser steps of the whole model (training_step -> Adam -> device-side refresh of the packed weights at the next forward):
    the teacher-student loss falls monotonically"""
    r = G.check_full_training(device="cpu", steps=5)
    assert all(b < a for a, b in zip(r["losses"], r["losses"][1:])), r


def test_uni_hamgnn_two_model_chain_on_cpu(cpu_backend):
    """BASELINE config #5 in small: non-SOC universal model (zero-point shift) -> Hon_nonsoc / Hoff_nonsoc -> SOC / so3 model with add_H_nonsoc
    (hamgnn_amd/uni.py) vs the same chain on the oracle; mixed-Z crystals, nao 26"""
    r = G.check_uni_chain_vs_oracle("cpu", irreps=G.MINI, n_graphs=2)
    assert r["real"] < G.TOL and r["imag"] < G.TOL and r["nonsoc"] < G.TOL, r


def test_uni_hamgnn_chain_on_a_batch_of_crystals_on_cpu(cpu_backend):
    """the non-SOC -> SOC hand-over for a batch of crystals == the chain crystal by crystal (the reference's batch_size=1 order)"""
    r = G.check_uni_chain_batched("cpu", irreps=G.MINI, n_graphs=3)
    assert r["hamiltonian_real_rel_err"] < 1e-6 and r["hamiltonian_imag_rel_err"] < 1e-6 and r["rows"] > 0, r


def test_default_irreps_si2_forward_on_cpu(cpu_backend):
    """BASELINE config #1 (Si diamond 2-atom cell, 172 edges) at the SHIPPED irreps (set A: D = 877, l <= 6, SH to l = 5, 64 radial, three
    layers, nao 19): the production planner output -- merged items, odd-path templates, split launches for a small crystal, the one-pass
    read-out tables -- through the product's host code on the CPU stand-ins vs the fp64 oracle"""
    r = G.check_default_irreps_si2("cpu", "A", "si2")
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["H_rel_err"] < G.TOL and r["Hnet_rel_err"] < G.TOL, r

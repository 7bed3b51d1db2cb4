"""-m gpu: the HIP path (through the C ABI) against the oracle / golden fixtures.  Tolerance: 1e-5 relative (max-norm),
the north_star's fp32 bar."""
import pytest
import torch

from tests import gpu_checks as G

G_IRREPS_A = "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e"

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def test_geometry():
    r = G.check_geometry()
    print(r)
    assert r["wigner_abs_err"] < 5e-6 and r["rbf_rel_err"] < 1e-6


@pytest.mark.parametrize("schedule", ["seg", "is"])
@pytest.mark.parametrize("unrotate", [True, False])
def test_message_pack_block_golden(unrotate, schedule):
    """the fused MessagePackBlock on both kernels: segment-stationary (tp_fused.hip) and input-stationary (tp_is.hip)"""
    r = G.check_message_pack(unrotate=unrotate, schedule=schedule)
    print(r)
    assert r["message_pack_rel_err"] < G.TOL


@pytest.mark.parametrize("schedule", ["seg", "auto"])
def test_backbone_golden_both_kernels(schedule, monkeypatch):
    """whole backbone on the segment-stationary kernel + hg_rotate_gather vs the default (input-stationary, fused gather+rotation)"""
    monkeypatch.setenv("HG_MP_KERNEL", schedule)
    r = G.check_backbone()
    print(r)
    assert r["backbone_node_rel_err"] < G.TOL and r["backbone_edge_rel_err"] < G.TOL


@pytest.mark.parametrize("schedule", ["seg", "auto"])
@pytest.mark.parametrize("seed", range(5))
def test_message_pack_random_irreps_vs_oracle(seed, schedule):
    r = G.check_message_pack_random(seed=seed, schedule=schedule)
    print(r)
    assert r["rel_err"] < G.TOL


@pytest.mark.parametrize("case", list(range(6)) + ["A", "B"])
def test_message_pack_single_part_vs_oracle(case):
    """the LARGE-graph path of csrc/tp_is.hip (one workgroup per 16-edge tile, `tp_is_kernel<false, false>`: what the benchmark runs) forced
    on small inputs: random irreps sets and the two shipped sets, 64-wide radial MLP (hidden rows resident in registers), vs the fp64 oracle"""
    import bench
    kw = dict(irr=bench.IRREPS[case], sh=bench.SH, seed=7, E=37) if isinstance(case, str) else dict(seed=case)
    r = G.check_message_pack_random(radial=(64, 64), parts=1, **kw)
    print(r)
    assert r["kernel"] == "is" and r["rel_err"] < G.TOL


def test_front_door_checkpoint_and_datasets_reproduce_the_fixtures():
    """SURVEY 8f-1 through the front door: Model.load_from_checkpoint(.ckpt) + NPZGraphDataset / LMDBGraphDataset -> HIP forward == the
    reference's outputs of the backbone and head fixtures (tests/gpu_checks.py:check_front_door).  Files written by real liblmdb / Lightning
    cannot be produced in this image."""
    r = G.check_front_door()
    print(r)
    assert all(v < G.TOL for v in r.values()), r


def test_fused_node_scatter_equals_message_rows_plus_segment_sum():
    r = G.check_fused_scatter()
    print(r)
    # (two summation orders of the receivers' sums: G.SAME_MATH_TOL; the edge rows downstream inherit the node rows' rounding)
    assert r["node_rel_err"] < G.SAME_MATH_TOL and r["edge_rel_err"] < G.SAME_MATH_TOL, r


@pytest.mark.parametrize("legacy", [False, True])
def test_structural_zero_inputs_of_the_first_layer(legacy):
    """r5: the programs of the leading layers skip the super-paths whose input irreps are structurally zero (node rows out of the 0e embedding Linear, edge rows
    out of the 0e x Y^l pair embedding; with legacy_edge_update the edge rows stay the embedding's for one more layer): same rows as the complete programs"""
    r = G.check_structural_zeros(legacy=legacy)
    print(r)
    # (a reduced program deals its items to the waves differently from the complete one: another order of the same sums, G.SAME_MATH_TOL)
    assert r["node_rel_err"] < G.SAME_MATH_TOL and r["edge_rel_err"] < G.SAME_MATH_TOL and r["first_layer_mfma_ratio"] < 0.6 and (legacy or r["last_layer_mfma_ratio"] == 1.0), r


SET_A = "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e"


@pytest.mark.parametrize("kw", [dict(), dict(num_layers=1), dict(soc=True), dict(soc="su2", n_atoms=5), dict(nonlinearity_type="norm"), dict(transformer=True),
                                dict(irr=SET_A, n_atoms=6), dict(irr=SET_A, workload="sio2_300")],
                         ids=["mini", "one_layer", "soc_so3", "soc_su2_reads_all", "norm_activation", "transformer", "setA", "setA_single_part"])
def test_unread_irreps_of_the_last_pair_block(kw):
    """r5: Model(representation, output) tells the backbone that the head is its only reader (HamGNNConvE3.declare_consumer): the last PairInteractionBlock
    leaves out the output irreps the head never reads (set-A, nao_max 19: 0o, 4o, 5o, 5e, 6e).  Same Hamiltonian rows; the head's claim holds bit for bit
    (noise in the unread blocks changes nothing); `edge_attr`, a wider head and training forwards get the complete rows."""
    r = G.check_dead_outputs(**kw)
    print(r)
    # the COMPLETE program run again on the same rows (`edge_attr` after a reduced forward, a head that reads more than was declared): the same launches as
    # the undeclared forward -> bit-identical since r6; reduced vs complete program, device repack vs fresh compile, training vs inference launches: other
    # orders of the same fp32 sums -> G.SAME_MATH_TOL (derived in tests/gpu_checks.py)
    if kw.get("soc") == "su2":
        assert r["dead_irreps"] == 0 and r["alive_declared"] == 0.0 and r["ham_rel_err"] == 0.0 and r["edge_attr_rel_err"] == 0.0 and r["wider_head_rel_err"] == 0.0, r
        return
    assert r["dead_irreps"] >= (5 if "irr" in kw else 1) and r["alive_declared"] == 1.0, r
    assert r["ham_rel_err"] < G.SAME_MATH_TOL and r["ham_noise_max_abs"] == 0.0 and r["dead_blocks_max_abs"] == 0.0, r
    # (one layer: the last pair block is also the FIRST -- its forward launch takes the structural-zero shortcut, the lazy complete re-run does not: two programs)
    same_launches = kw.get("num_layers") != 1
    assert (r["edge_attr_rel_err"] == 0.0 and r["wider_head_rel_err"] == 0.0) if same_launches else max(r["edge_attr_rel_err"], r["wider_head_rel_err"]) < G.SAME_MATH_TOL, r
    assert r["last_pair_mfma_ratio"] < (0.9 if "irr" in kw else 1.0), r
    assert r["training_rows_rel_err"] < G.SAME_MATH_TOL and r["training_alive_declared"] == 0.0 and r["refresh_rel_err"] < G.SAME_MATH_TOL, r


def test_corr_product_block_golden():
    r = G.check_corr_product()
    print(r)
    assert r["corr_product_rel_err"] < G.TOL


@pytest.mark.parametrize("name", ["corr_product_block_nu3", "corr_product_block_nu1"])
def test_corr_product_block_other_correlations_golden(name):
    """config key `correlation` = 3 (hg_sym_contraction + the nu = 3 term of hamgnn_amd/corr3.py) and 1, against the reference's outputs"""
    r = G.check_corr_product(name=name)
    print(r)
    assert r["corr_product_rel_err"] < G.TOL


def test_backbone_use_corr_prod_golden():
    r = G.check_backbone(name="backbone_corr")
    print(r)
    assert r["backbone_node_rel_err"] < G.TOL and r["backbone_edge_rel_err"] < G.TOL


def test_backbone_gaussian_rbf_golden():
    """rbf_func="gaussian": hg_radial_basis (GaussianSmearing x cosine cutoff) vs the reference fixture"""
    r = G.check_backbone(name="backbone_gaussian_rbf")
    print(r)
    assert r["backbone_node_rel_err"] < G.TOL and r["backbone_edge_rel_err"] < G.TOL


def test_backbone_charge_doping_golden():
    r = G.check_charge_doping()
    print(r)
    assert r["effect_of_charge"] > 1e-3                                    # the fixture's charges do move the features
    assert all(v < G.TOL for k, v in r.items() if k.endswith("rel_err")), r


def test_backbone_charge_doping_with_corr_product_golden():
    """apply_charge_doping + use_corr_prod (the reference's default when the key is missing): hg_sym_contraction on per-node weight mixtures"""
    r = G.check_charge_doping_corr()
    print(r)
    assert r["effect_of_charge"] > 1e-3
    assert all(v < G.TOL for k, v in r.items() if k.endswith("rel_err")), r


def test_block_gemm_kernel_vs_float64_matmuls():
    """hg_block_gemm: plain / transposed operands, ragged sizes, fp32 and fp64 results (fp64 accumulation)"""
    r = G.check_block_gemm()
    print(r)
    assert r["f64_rel_err"] < 1e-13 and r["f32_rel_err"] < 2e-7, r


def test_transformer_backbone_golden():
    r = G.check_transformer()
    print(r)
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["block_rel_err"] < G.TOL


def test_transformer_vs_oracle_random_crystal():
    r = G.check_transformer_vs_oracle()
    print(r)
    assert r["max_in_degree"] > 130                                        # more than one LDS weight chunk of hg_attn_aggregate
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["H_rel_err"] < G.TOL


@pytest.mark.parametrize("seed,schedule", [(0, "auto"), (1, "auto"), (2, "seg"), (3, "auto"), (5, "auto")])
def test_message_pack_data_gradient_vs_autograd(seed, schedule):
    r = G.check_message_pack_backward(seed=seed, schedule=schedule)
    print(r)
    assert r["kernel"] == ("seg" if schedule == "seg" else "is")
    assert max(r["g_src_rel_err"], r["g_dst_rel_err"], r["g_edge_rel_err"]) < G.TOL


def test_message_pack_data_gradient_default_irreps():
    """the shipped set-A irreps (l <= 6, sh lmax 5, hidden 64): three feature rows of output per edge -> several workgroups per tile"""
    r = G.check_message_pack_backward(seed=7, irr=G_IRREPS_A, sh="0e+1o+2e+3o+4e+5o", E=45, radial=(64, 64))
    print(r)
    assert r["kernel"] == "is" and r["parts"] >= 3
    assert max(r["g_src_rel_err"], r["g_dst_rel_err"], r["g_edge_rel_err"]) < G.TOL


def test_conv_message_chain_data_gradient_vs_autograd():
    r = G.check_conv_message_backward()
    print(r)
    assert r["g_node_rel_err"] < G.TOL and r["g_edge_rel_err"] < G.TOL


def test_model_test_stage_writes_the_reference_files():
    r = G.check_test_stage()
    print(r)
    assert r["rows"] == r["rows_expected"] and r["target_max_abs_diff"] == 0.0 and r["finite"] and r["same_as_returned"]


@pytest.mark.parametrize("irr", [None, "8x0e+4x0o+4x1o+2x1e+2x2o+3x2e+2x3o"])
def test_residual_block_backward_vs_autograd(irr):
    r = G.check_residual_block_backward(irr=irr)
    print(r)
    assert max(r.values()) < G.TOL


@pytest.mark.parametrize("seed", [0, 1, 3])
def test_message_pack_weight_gradients_vs_autograd(seed):
    r = G.check_message_pack_weight_grads(seed=seed)
    print(r)
    assert r["max_rel_err"] < G.TOL


def test_message_pack_weight_gradients_default_irreps():
    r = G.check_message_pack_weight_grads(seed=7, irr=G_IRREPS_A, sh="0e+1o+2e+3o+4e+5o", E=37, radial=(64, 64), num_radial=64)
    print(r)
    assert r["max_rel_err"] < G.TOL


@pytest.mark.parametrize("legacy,metric", [(False, "mse"), (True, "mae")])
def test_full_model_backward_vs_autograd(legacy, metric):
    r = G.check_full_backward(legacy=legacy, metric=metric)
    print(r)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL, r


def test_full_model_backward_materialisation_route(monkeypatch):
    """HG_WGRAD=rows: the K2 weight gradients through the two materialisation programs + library reductions (the route irreps without a fused
    kernel instantiation fall back to; the default since r3 is the fused kernel hg_tp_wgrad)"""
    monkeypatch.setenv("HG_WGRAD", "rows")
    r = G.check_full_backward(legacy=False, metric="mse")
    print(r)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL, r


@pytest.mark.parametrize("kw", [dict(), dict(charge=True, crystals=2), dict(legacy=True)], ids=["plain", "charge", "legacy"])
def test_full_model_backward_fused_routes_vs_autograd(kw):
    """64-wide radial layers (the shipped width): every weight gradient of the message blocks through hg_tp_wgrad with the structural-zero shortcut in the
    first layer, the embedding TP's through the same kernel (its 24-channel 0e row as two 12-channel sources) + its adjoint program (late r5) -- loss and
    all parameter gradients vs torch.autograd through the fp64 oracle"""
    r = G.check_full_backward(**dict(dict(radial=(16, 64), num_types=24, n_atoms=4, seed=10), **kw))
    print(r)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL, r


def test_full_model_backward_batch_of_crystals():
    """three crystals of different sizes in one batch (per-crystal [on-site; off-site] row order of the result, batch-global inverse edges)"""
    r = G.check_full_backward(n_atoms=3, seed=8, crystals=3, metric="mae")
    print(r)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL, r


@pytest.mark.parametrize("soc", ["so3", "so3_nonsoc"])
def test_full_model_backward_soc(soc):
    """SOC / so3 model (ksi networks + spin assembly); "so3_nonsoc": the Uni-HamGNN SOC training mode (spin-free block given, add_H_nonsoc)"""
    r = G.check_full_backward(n_atoms=4, seed=6, soc=soc, crystals=2 if soc == "so3_nonsoc" else 1)
    print(r)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL, r


def test_full_model_backward_charge_doping():
    """apply_charge_doping: node attributes one_hot(z) + mlp_q(gauss(q)) - mlp_q(gauss(0)); two crystals with different charges -- the
    charge MLP's four parameters, the embedding tables and everything downstream"""
    r = G.check_full_backward(n_atoms=4, seed=7, crystals=2, charge=True)
    print(r)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL and r["n_params"] >= 78, r


def test_full_model_backward_charge_doping_with_corr_product():
    """both together: the gradient with respect to the doped attributes also flows through the CorrProductBlocks' weight mixtures"""
    r = G.check_full_backward(n_atoms=4, seed=9, crystals=2, charge=True, corr=True)
    print(r)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL and r["n_params"] > 94, r



def test_full_model_backward_corr_product():
    """use_corr_prod: a CorrProductBlock (MACE symmetric contraction with element-dependent weights) after every ConvBlock"""
    r = G.check_full_backward(n_atoms=5, seed=8, corr=True)
    print(r)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL and r["n_params"] > 90, r


def test_full_model_backward_corr_product_correlation_3():
    """correlation 3 with doped node attributes: forward and every gradient through the nu = 3 term"""
    r = G.check_full_backward(n_atoms=2, seed=6, crystals=2, charge=True, corr=3, irr="6x0e+3x0o+3x1o+2x1e+2x2e", nao=13)
    print(r)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL and r["n_params"] > 100, r


def test_full_model_backward_transformer():
    """HamGNNTransformer: attention blocks (soft-max over incoming edges, learnable soft cutoff, value MessagePackBlock) + CorrProductBlocks"""
    r = G.check_full_backward(n_atoms=5, seed=9, transformer=True, irr="8x0e+4x0o+4x1o+2x1e+2x2o+4x2e+2x3o")
    print(r)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL and r["n_params"] > 120, r


@pytest.mark.parametrize("kw", [dict(lite=True), dict(lite=True, legacy=True, crystals=2, n_atoms=2),
                                dict(zps=True, crystals=2, n_atoms=2), dict(zps=True, soc="so3")],
                         ids=["lite", "lite_legacy_batch", "zero_point_shift", "zero_point_shift_soc"])
def test_full_model_backward_lite_mode_and_zero_point_shift(kw):
    """lite_mode backward (hamgnn_amd/backward_lite.py: adjoint IT_LINC program, streaming-Linear adjoints) and the zero-point shift's adjoint
    on the HIP kernels vs autograd through the fp64 oracle (round 2 had run these on the CPU stand-ins only)"""
    r = G.check_full_backward(**dict(dict(n_atoms=3, seed=5), **kw))
    print(r)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL


def test_reference_loss_semantics():
    """Model.py:150-166: hamiltonian-type losses x sparsity_ratio; SOC training on hamiltonian_real + hamiltonian_imag with their weights"""
    r = G.check_full_backward(n_atoms=4, num_layers=1, nao=14, sparsity=True)
    print(r)
    assert r["loss_rel_err"] < 1e-5 and r["max_rel_err"] < 2e-5
    r = G.check_full_backward(n_atoms=4, num_layers=1, nao=14, soc="so3", sparsity=True, split_losses=True, metric="mae")
    print(r)
    assert r["loss_rel_err"] < 1e-5 and r["max_rel_err"] < 2e-5


@pytest.mark.parametrize("zps", [False, True])
def test_band_energy_loss_with_zero_point_shift(zps):
    """hamiltonian + band_energy losses vs autograd through the oracle; with zero_point_shift the band gradient enters BEFORE the shift and
    carries the adjoint of the mean alignment (complex64 eigensolver on the GPU against fp64: looser bar)"""
    r = G.check_full_backward(n_atoms=3, num_layers=1, nao=13, metric="mae", zps=zps, bands=True)
    print(r)
    assert r["loss_rel_err"] < 1e-4 and r["max_rel_err"] < 2e-3


@pytest.mark.parametrize("zps", [False, True])
def test_band_energy_loss_on_a_spin_orbit_head(zps):
    """the same two-loss step on a SOC / so3 head: bands of the stacked spinor H(k), gradient through the four spin blocks' assembly adjoints"""
    r = G.check_full_backward(n_atoms=2, num_layers=1, nao=13, metric="mse", zps=zps, bands=True, soc="so3")
    print(r)
    assert r["loss_rel_err"] < 1e-4 and r["max_rel_err"] < 2e-3


def test_full_model_backward_default_irreps():
    """one layer at the reference's default irreps (set A: 877 channels, l <= 6, SH to l = 5, 64-wide radial MLPs), 4-atom cell"""
    r = G.check_full_backward(n_atoms=4, seed=5, num_layers=1, irr=G_IRREPS_A, sh="0e+1o+2e+3o+4e+5o", radial=(64, 64), num_radial=64)
    print(r)
    assert r["loss_rel_err"] < G.TOL and r["max_rel_err"] < G.TOL, r


@pytest.mark.parametrize("legacy", [False, True])
def test_device_repack_equals_recompile(legacy):
    r = G.check_refresh_equals_recompile(legacy=legacy)
    print(r)
    # (device repack: hg_block_gemm forms the L' products in fp64 in its own order, one fp64 ulp from the host's BLAS -> fp32 weights that may differ in the last bit)
    assert r["packers"] >= 8 and r["loss_rel_diff"] < G.SAME_MATH_TOL and r["grad_max_rel_diff"] < G.TOL, r
    assert r["inference_rel_diff"] < G.SAME_MATH_TOL and r["step_moved_H"] > 1e-3, r      # inference -> step -> inference: no stale cached chain


def test_full_model_training_loss_falls():
    r = G.check_full_training()
    print(r)
    assert r["last"] < 0.8 * r["first"] and all(b < a for a, b in zip(r["losses"], r["losses"][1:])), r


@pytest.mark.parametrize("nonsoc,crystals", [(False, 1), (True, 2)])
def test_soc_head_backward_vs_autograd(nonsoc, crystals):
    r = G.check_soc_head_backward(add_H_nonsoc=nonsoc, crystals=crystals)
    print(r)
    assert r["trained"] >= 6
    assert all(v < G.TOL for k, v in r.items() if k.endswith("rel_err")), r


@pytest.mark.parametrize("basis,n_atoms", [("su2", 4), ("su2_f", 2)], ids=["siesta_13", "abacus_27_f_shells_l7"])
def test_soc_su2_head_backward_vs_autograd(basis, n_atoms):
    """SOC / su2 head (siesta-13: spinor CG merge, [real | imaginary] planes finished with sign +1 / -1; abacus-27: f shells, couplings
    up to l = 7 -- the gradient rows are rotated by the l = 7 instantiation of hg_rotate_gather)"""
    r = G.check_soc_head_backward(basis=basis, n_atoms=n_atoms)
    print(r)
    assert all(v < G.TOL for k, v in r.items() if k.endswith("rel_err")), r


def test_head_backward_vs_autograd():
    r = G.check_head_backward()
    print(r)
    assert max(r.values()) < G.TOL


def test_head_finetune_loss_falls():
    r = G.check_head_finetune()
    print(r)
    assert r["last_loss"] < 0.5 * r["first_loss"] and r["monotone_fraction"] > 0.7


def test_backbone_golden():
    r = G.check_backbone()
    print(r)
    assert r["backbone_node_rel_err"] < G.TOL and r["backbone_edge_rel_err"] < G.TOL


@pytest.mark.parametrize("name,ham_type,nao", [("head_openmx_19", "openmx", 19), ("head_abacus_13", "abacus", 13)])
def test_head_golden(name, ham_type, nao):
    r = G.check_head(name=name, ham_type=ham_type, nao=nao)
    print(r)
    assert r[name + "_rel_err"] < G.TOL
    assert abs(r["sparsity_ratio"] - r["sparsity_ratio_reference"]) < 1e-6 * r["sparsity_ratio_reference"]     # hamgnn_output.py:2784-2872


def test_head_nonlinearity_type_norm_golden_and_backward():
    """config key `nonlinearity_type: norm` of the head (hamgnn_output.py:38-58 -> interaction_blocks.py:311-330, e3nn NormActivation): the
    reference's own rows (fixture from its HamGNNPlusPlusOut / ResidualBlock) and the head's backward vs autograd through the oracle"""
    r = G.check_head(name="head_norm_openmx_19", nonlinearity_type="norm")
    print(r)
    assert r["head_norm_openmx_19_rel_err"] < G.TOL and r["residual_block_rel_err"] < G.TOL
    b = G.check_head_backward(nonlinearity_type="norm")
    print(b)
    assert all(v < 2e-5 for v in b.values()), b


def test_full_forward_vs_oracle_random_cell():
    r = G.oracle_vs_hip_random()
    print(r)
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["H_rel_err"] < G.TOL


def test_head_overlap_networks_golden():
    r = G.check_head_overlap()
    print(r)
    assert r["overlap_rel_err"] < G.TOL and r["hamiltonian_rel_err"] < G.TOL


def test_head_soc_so3_golden():
    r = G.check_head_soc()
    print(r)
    assert r["soc_real_rel_err"] < G.TOL and r["soc_imag_rel_err"] < G.TOL


def test_zero_point_shift():
    r = G.check_zero_point_shift()
    print(r)
    assert r["zero_point_rel_err"] < G.TOL and r["soc_zero_point_rel_err"] < G.TOL
    assert r["shift_effect"] > 1e-3 and r["soc_shift_effect"] > 1e-3


def test_head_soc_su2():
    r = G.check_head_su2()
    print(r)
    assert all(v < G.TOL for v in r.values()), r


@pytest.mark.parametrize("workload,irreps,world,node_shard", [("si64", "B", 2, "0"), ("sio2_300", "A", 2, "0"), ("sio2_300", "A", 8, "0"), ("sio2_300", "A", 8, "1")])
def test_sharded_forward_matches_single_rank(workload, irreps, world, node_shard):
    """2 / 8 ranks (gloo) sharing cuda:0: pair-sharded edges + all-reduce of node aggregates == unsharded forward; also on BASELINE config
    #4's generator (amorphous SiO2, set-A) at the world size the scaling bench ends with: eight processes initialise, partition, run the HIP
    kernels on their shards and meet in the three all-reduces (RCCL itself refuses several ranks on one device, so the collective leg is
    gloo here).  The child asserts rel_err < 1e-5 itself; here the return code AND the printed figure count.
    node_shard = "1" (r5, HG_NODE_SHARD): the node-level chain of every ConvBlock on the rank's block of rows, reduce-scatter / all-gather around it."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HG_DIST_WORKLOAD=workload, HG_DIST_IRREPS=irreps, HG_NODE_SHARD=node_shard)
    cp = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                         "--master-port", str(29541 + world), os.path.join(root, "tests", "dist_gpu_check.py")], capture_output=True, text=True, timeout=900, env=env)
    tail = cp.stdout[-2000:] + cp.stderr[-2000:]
    assert cp.returncode == 0, tail
    lines = [l for l in cp.stdout.splitlines() if l.startswith("DIST_CHECK ")]
    assert lines, tail
    r = json.loads(lines[-1][len("DIST_CHECK "):])
    print(r)
    assert r["world"] == world and min(r["edges_per_rank"]) > 0 and sum(r["edges_per_rank"]) == r["E"]
    assert max(r["edges_per_rank"]) < 1.05 * r["E"] / world + 64 and r["rel_err"] < 1e-5, r


@pytest.mark.parametrize("mode", ["train_conv", "train_attn", "dp"])
def test_two_rank_training_paths_on_one_gpu(mode):
    """the multi-rank training paths ON THE HIP KERNELS (2 ranks over gloo sharing cuda:0): model-parallel training step of both backbones on
    an edge-sharded crystal (sharded attention forward + backward, zero-point shift and sparsity ratio of the whole crystal) == the
    single-process step; data-parallel allreduce_gradients == the mean of the single-process gradients"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cp = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29551", os.path.join(root, "tests", "dist_gpu_train_check.py")], capture_output=True, text=True,
                        timeout=900, env=dict(os.environ, HG_DIST_MODE=mode))
    tail = cp.stdout[-2000:] + cp.stderr[-2000:]
    assert cp.returncode == 0, tail
    lines = [l for l in cp.stdout.splitlines() if l.startswith("DIST_TRAIN ")]
    assert lines, tail
    r = json.loads(lines[-1][len("DIST_TRAIN "):])
    print(r)
    assert r["loss_err"] < 1e-5 and r["grad_err"] < 5e-5 and r["n"] > 100, r
    if mode == "dp":
        assert r["differs_from_rank0_alone"] > 1e-3, r          # the mean is not rank 0's own gradient


def test_rccl_backend_single_rank():
    """the RCCL leg of the multi-GPU path (init with device_id, all_reduce, barrier) runs on this image / box -- one rank"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cp = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_smoke.py")], capture_output=True, text=True, timeout=300)
    assert cp.returncode == 0 and "RCCL_OK" in cp.stdout, cp.stdout[-1500:] + cp.stderr[-1500:]


@pytest.mark.parametrize("launcher", ["bare", "torchrun"])
def test_bench_script_two_rank_path_on_one_gpu(launcher):
    """`bench.py --gpus 2` on a 1-GPU box: both ranks share cuda:0 and talk over gloo (HG_BENCH_SAME_DEVICE / HG_BENCH_BACKEND test hooks; RCCL refuses two
    ranks on one device).  "bare": the ONE command `python bench.py --gpus 2` -- the script launches its own ranks (r6; VERDICT r5 #2: the first 8-GPU
    invocation must not die in argument handling); "torchrun": exactly as the driver launches it (torch.distributed.run, one process per rank).
    Checks the sharded N > 1 code path of the benchmark itself: one JSON line from rank 0, whole-job edge count, n_gpus, return code."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HG_BENCH_SAME_DEVICE="1", HG_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    tail = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "si512", "--irreps", "B"]
    head = [sys.executable] if launcher == "bare" else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                                         "--master-port", "29547"]
    cp = subprocess.run(head + tail, capture_output=True, text=True, timeout=600, env=env)
    assert cp.returncode == 0, cp.stdout[-1500:] + cp.stderr[-1500:]
    lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, cp.stdout[-1500:]
    r = json.loads(lines[0])
    print({k: r[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")})
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["value"] > 0 and r["scaling"] == "strong" and "cpu_baseline" not in r
    assert "pair-sharded" in r["config"]["parallelism"] and r["sharded_check"]["rel_err"] < 1e-5


def test_bench_script_propagates_a_failing_rank():
    """a rank that cannot run (here: --gpus 2 without the shared-device hook on a 1-GPU box -> LOCAL_RANK 1 has no device) makes the ONE-command form exit
    non-zero and say which rank it was, instead of hanging or printing a line"""
    import os, subprocess, sys
    if torch.cuda.device_count() > 1:
        pytest.skip("needs a box with exactly one GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HG_BENCH_SAME_DEVICE")}
    cp = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "si64", "--irreps", "B"],
                        capture_output=True, text=True, timeout=300, env=env)
    assert cp.returncode != 0 and "BENCH_RANK_FAILURE" in cp.stderr and not [l for l in cp.stdout.splitlines() if l.startswith("{")], cp.stdout[-800:] + cp.stderr[-800:]


def test_radial_mlp_with_five_hidden_layers_vs_oracle():
    """r6: `radial_MLP` deeper than the three layers one hg_radial_hidden launch holds (chained launches): forward vs the fp64 oracle, and the training step's
    gradients vs autograd through it"""
    r = G.oracle_vs_hip_random(n_atoms=5, seed=4, radial=(8, 16, 24, 16, 32))
    print(r)
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["H_rel_err"] < G.TOL
    b = G.check_full_backward(n_atoms=3, seed=4, radial=(8, 16, 24, 16, 32))
    print(b)
    assert b["loss_rel_err"] < G.TOL and b["max_rel_err"] < G.TOL, b


def test_multi_crystal_batch_vs_oracle():
    r = G.oracle_vs_hip_random(n_graphs=3, seed=5)
    print(r)
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["H_rel_err"] < G.TOL


def test_ragged_batch_with_an_edgeless_crystal_vs_oracle():
    """ragged batch: crystals of 6 / 1 / 8 atoms, the middle one without a single edge (its on-site block comes from the embeddings alone;
    empty segments in the receiver CSR, an empty slice in the per-crystal [on-site; off-site] order).  (A batch with NO edges at all is
    not a case: the reference's AttentionHeadsToVector `.view(0, -1)` raises on it, nn/attention_utils.py:116.)"""
    r = G.oracle_vs_hip_random(n_graphs=2, seed=9, isolated=True)
    print(r)
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["H_rel_err"] < G.TOL


def test_uni_hamgnn_style_batch_vs_oracle():
    """BASELINE config #5 in small: mixed-Z multi-crystal batch, nao_max 26 (f shells), legacy_edge_update (layer 0 keeps the
    embedded edge features), SOC so3 head -- full HIP forward vs the fp64 oracle."""
    r = G.oracle_vs_hip_random(n_graphs=3, seed=9, nao=26, legacy_edge_update=True, zs=(14, 8, 79, 42), soc=True)
    print(r)
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["H_rel_err"] < G.TOL


@pytest.mark.parametrize("which", ["A", "B"])
def test_si2_default_irreps_vs_oracle(which):
    """BASELINE config #1 with the shipped irreps (set-A: D=877, l<=6) and the lmax-4 set (set-B)."""
    r = G.check_default_irreps_si2(which=which)
    print(r)
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["H_rel_err"] < G.TOL and r["Hnet_rel_err"] < G.TOL


def test_band_energy_loss_backward():
    r = G.check_band_energy_backward()
    print(r)
    assert r["g_on_rel_err"] < 1e-4 and r["g_off_rel_err"] < 1e-4          # complex64 Cholesky / inverse / eigh chain vs the fp64 reference (measured 4e-6)
    assert r["grads_finite"] and r["losses"][-1] < r["losses"][0], r


def test_band_energies_k_space_step():
    """calculate_band_energy=True (non-SOC, reference overlaps) -- H(k) / S(k) assembly kernel + hipSOLVER eigensolver vs the
    reference's output; fp32 complex arithmetic on the GPU against fp64: eigenvalues to 1e-4 of the spectrum's scale"""
    r = G.check_band_energies()
    print(r)
    assert r["band_energy_err"] < 1e-4 and r["band_gap_err"] < 1e-4 and r["window_err"] < 1e-4
    assert r["forward_ok"] and r["kpath_ok"] and r["with_overlap_ok"] and r["targets_consistent"] < 1e-5


def test_band_energies_spin_orbit_k_space_step():
    """calculate_band_energy=True of the spin-orbit branches (hamgnn_output.py:1998-2286): four spin blocks of H(k) from the
    real / imaginary rows, kron(1_2, S(k)), eigensolver; vs the reference's output"""
    r = G.check_band_energies_soc()
    print(r)
    assert r["bands"] == r["ref_bands"] and r["band_energy_err"] < 1e-4 and r["window_err"] < 1e-4
    assert r["forward_ok"] and r["targets_consistent"] < 1e-5


@pytest.mark.parametrize("tag", ["batch", "single"])
def test_head_bands_with_zero_point_shift(tag):
    """calculate_band_energy + zero_point_shift vs the reference's forward (bands from the UNSHIFTED blocks, then aligned by their mean)"""
    r = G.check_head_bands_zero_point(tag=tag)
    print(r)
    assert r["H_rel_err"] < G.TOL and r["unshifted_H_rel_err"] < G.TOL
    assert r["band_energy_err"] < 1e-4 and r["unshifted_band_energy_err"] < 1e-4 and r["target_band_energy_err"] < 1e-4
    assert r["shift_matters"] > 1e-3 and r["H_shift_matters"] > 1e-3


def test_band_cal_along_a_k_path():
    """hamgnn_amd.band_cal.band_structure (DFT_interfaces/openmx/band_cal.py:64-108, 296-392: saved prediction rows -> bands along a k-path, eV
    relative to the valence-band maximum) on hg_hk_assemble + hipSOLVER vs the script's dense numpy / scipy loop in fp64"""
    r = G.check_band_cal()
    print(r)
    assert r["bands_rel_err"] < 1e-4 and r["gap_abs_err_eV"] < 1e-2 and r["crystals"] == 2


def test_band_cal_spin_orbit_and_collinear_branches():
    """the `soc_switch` (band_cal.py:101-283) and `spin_colinear` (:284-452) branches of the post-processing script vs dense fp64 restatements of its loops"""
    r = G.check_band_cal_spin()
    print(r)
    assert r["soc_bands_rel_err"] < 1e-4 and r["soc_gap_abs_err_eV"] < 1e-2 and r["collinear_bands_rel_err"] < 1e-4, r


def test_band_energies_export_reciprocal_values():
    """export_reciprocal_values (hamgnn_output.py:1368-1673, 1675-1996 with the flag): H(k), S(k), dS(k), normalised wavefunctions vs the reference"""
    r = G.check_band_energies_export()
    print(r)
    assert all(v < (2e-3 if k.endswith("wf_abs_err") else 2e-4) for k, v in r.items()), r


def test_linear_weight_gradient_kernel():
    """hg_linear_wgrad (csrc/linear_wgrad.hip): every path of an o3.Linear's weight gradient in one launch vs per-path fp64 GEMMs"""
    r = G.check_linear_wgrad_kernel()
    print(r)
    assert all(v < 2e-6 for v in r.values()), r


def test_attribute_style_graph_object():
    """a non-dict graph object (PyG Data look-alike) through backbone and head; the topology cache is stored on the object"""
    r = G.check_attribute_style_graph()
    print(r)
    assert r["node_attr"] == 0.0 and r["edge_attr"] == 0.0 and r["node_vs_fixture"] < G.TOL and r["head_vs_fixture"] < G.TOL      # (the same launches on the same rows: bit-identical since r6)
    assert r["cache_reused"]
    assert abs(r["sparsity_ratio"] - r["sparsity_ratio_fixture"]) < 1e-6 * r["sparsity_ratio_fixture"]


def test_captured_forward_replay_si2():
    """BASELINE config #1 as a HIP graph: replay == eager (also after an in-place position update), and faster than eager launches"""
    r = G.check_captured_forward_si2()
    print(r)
    # (the replayed graph holds the finer 2d split: a workgroup per (segment, share of its phases) ADDING its tiles -- another order of the same sums than the
    #  eager launches', and the one schedule whose own order is not fixed: G.SAME_MATH_TOL)
    assert r["replay_vs_eager"] < G.SAME_MATH_TOL and r["replay_vs_eager_moved"] < G.SAME_MATH_TOL and r["moved_changes_H"] > 1e-4
    assert r["replay_ms"] < 1.1 * r["eager_ms"] and r["replay_ms"] < 2.0     # r1: 4.4-4.8 ms per forward, eager or replayed
    # CapturedForward(fine_split=False): the eager launches replayed -- bit-identical to the eager forward and to itself
    assert r["deterministic_replay_vs_eager_max_abs"] == 0.0 and r["deterministic_replay_repeat_max_abs"] == 0.0 and r["deterministic_replay_ms"] < 1.1 * r["eager_ms"], r


def test_backbone_lite_mode_golden():
    """lite_mode (uvu products + plain Linears + one combined radial scale, message_passing.py:197-215) incl. the lite embedding."""
    r = G.check_backbone(name="backbone_lite")
    print(r)
    assert r["backbone_node_rel_err"] < G.TOL and r["backbone_edge_rel_err"] < G.TOL


def test_sio2_setA_vs_oracle():
    """BASELINE config #4's generator (amorphous SiO2, Si:O mixed radii) with the shipped set-A irreps, 3 layers, at a size the
    fp64 oracle affords (60 atoms / 4 904 edges): full HIP forward vs the oracle."""
    r = G.check_default_irreps_si2(which="A", graph="sio2_60")
    print(r)
    assert r["E"] > 4000
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["H_rel_err"] < G.TOL and r["Hnet_rel_err"] < G.TOL


def test_mos2_soc_setA_vs_oracle():
    """BASELINE config #3's generator (2H-MoS2 monolayer, 4 x 4 cells = 48 atoms) with the shipped set-A irreps, 3 layers and the SOC / so3 read-out
    (hamgnn_output.py:3026-3144): full HIP forward vs the fp64 oracle on [real | imaginary] rows of the (2 nao)^2 spin blocks"""
    r = G.check_default_irreps_si2(which="A", graph="mos2_4", soc=True)
    print(r)
    assert r["E"] > 2000
    assert r["node_rel_err"] < G.TOL and r["edge_rel_err"] < G.TOL and r["H_rel_err"] < G.TOL and r["Hnet_rel_err"] < G.TOL


def test_uni_hamgnn_chain_vs_oracle():
    """BASELINE config #5 as the reference runs it: 8 mixed-Z crystals, set-A, nao 26, non-SOC -> SOC(add_H_nonsoc) chain."""
    r = G.check_uni_chain_vs_oracle()
    print(r)
    assert r["nonsoc"] < G.TOL and r["real"] < G.TOL and r["imag"] < G.TOL


def test_uni_hamgnn_chain_on_a_batch_of_crystals():
    """config #5 issued as ONE batch per model instead of one crystal per forward: same rows (set-A irreps, mixed-Z crystals)"""
    r = G.check_uni_chain_batched(n_graphs=4)
    print(r)
    assert r["hamiltonian_real_rel_err"] < 1e-5 and r["hamiltonian_imag_rel_err"] < 1e-5


def test_uni_hamgnn_chain_full_size_properties():
    r = G.check_uni_chain_full_size()
    print(r)
    assert r["diag_vs_nonsoc"] == 0.0 and r["herm_err"] < 1e-6 and r["atoms"] >= 8 * 32


@pytest.mark.parametrize("workload,which,soc", [("si512", "B", False), ("mos2_1200", "A", True), ("sio2_10k", "A", False)])
def test_full_size_properties(workload, which, soc):
    """BASELINE configs #2 (Si512, set-B), #3 (MoS2 1200 atoms + SOC, set-A) and #4 (a-SiO2 10 002 atoms, set-A: the benchmarked
    forward, same weights) at full size: symmetry / Hermiticity, rotation
    invariants (eigenvalues, singular values), translation invariance."""
    r = G.check_full_size_properties(workload=workload, which=which, soc=soc)
    print(r)
    assert r["onsite_sym_err"] < 1e-6 and r["offsite_sym_err"] < 1e-6
    # (SOC / so3: the rotation invariants are taken on the spin-diagonal real block, the one that does not read the crystal-frame L data)
    assert r["rot_onsite_eig_err"] < 5e-5 and r["rot_offsite_sv_err"] < 5e-5
    assert r["rot_changes_H"] > 1e-2
    assert r["translation_err"] < 5e-5


@pytest.mark.parametrize("seed", range(4))
def test_tp_wgrad_kernel_vs_twin(seed):
    """hg_tp_wgrad vs its numpy twin on the same tables (random irreps sets; seed 3: set-B with l up to 4, two sources, 64-channel blocks)"""
    import bench
    r = G.check_tp_wgrad_kernel(seed=seed, **({"irr": bench.IRREPS["B"], "sh": "0e+1o+2e+3o+4e", "E": 70} if seed == 3 else {}))
    print(r)
    assert r["acc_rel_err"] < 2e-5 and r["gs_rel_err"] < 2e-5, r              # fp32 sums over up to 150 edges x 13 columns against float64


@pytest.mark.parametrize("seed", range(4))
def test_row_program_kernel_vs_separate_kernels_and_twin(seed):
    """hg_row_program (HamLayer as one LDS-resident pass) == Linear1 / gate / Linear2 + x / linear_transform as separate launches == the twin"""
    import bench
    r = G.check_row_program_kernel(seed=seed, **({"irr": bench.IRREPS["A"], "nao": 19, "rows": 45} if seed == 3 else {}))
    print(r)
    assert r["used"] and r["vs_separate_rel_err"] < G.SAME_MATH_TOL and r["vs_twin_rel_err"] < G.SAME_MATH_TOL, r      # (K order of the fused Linear units vs the streaming Linear's)


def test_round3_kernels_full_size_properties():
    """hg_row_program at 822 350 rows and hg_tp_wgrad at 131 072 edges (set-A): size-independent properties, see the check"""
    r = G.check_new_kernels_full_size()
    print(r)
    assert r["rowprog_vs_separate"] < G.SAME_MATH_TOL and r["rowprog_subrange"] == 0.0      # (a sub-range of the rows through the same kernel: the same sums)
    assert r["wgrad_linearity_acc"] < 2e-5 and r["wgrad_linearity_gs"] < 2e-5 and r["wgrad_splits"] < 2e-5 and r["wgrad_halves"] < 2e-5


@pytest.mark.parametrize("legacy", [False, True])
def test_backward_skips_structural_zero_inputs(legacy):
    """r5: the backward of the first-layer blocks without the super-paths that read structurally zero input irreps (hg_tp_wgrad tables without their row
    tiles, adjoint program without their items): loss and every parameter gradient of a training step as with the shortcut off"""
    r = G.check_structural_zeros_backward(legacy=legacy)
    print(r)
    assert r["loss_rel_err"] < 1e-6 and r["grad_max_rel_err"] < 2e-5 and r["fused_route"] == 1.0, r
    assert r["first_conv_wgrad_mfma_ratio"] < 0.6 and r["first_conv_adjoint_mfma_ratio"] < 0.6, r


def test_split_radial_scale_full_size():
    """r6: the radial scales on the half-precision matrix pipe with split operands (27.5 % of a MessagePackBlock's MFMAs: 6 x 16 pipe cycles per row tile instead of
    16 x 32) on the 10 002-atom benchmark crystal and on a 48-atom MoS2 sheet (split launches): repeated forwards bit-identical, the rows within the same-math
    tolerance of the fp32 form of the same build"""
    r = G.check_split_radial_scale()
    print(r)
    assert r["single_part"] == 1.0 and r["twins_flagged"] == 1.0 and r["E"] > 800000 and r["small_split_launch"] == 1.0, r
    assert r["big_repeat_max_abs"] == 0.0 and r["small_repeat_max_abs"] == 0.0, r
    for k in ("big_split_vs_fp32_node", "big_split_vs_fp32_edge", "small_split_vs_fp32_node", "small_split_vs_fp32_edge"):
        assert 0.0 < r[k] < G.SAME_MATH_TOL, (k, r)


@pytest.mark.parametrize("which,graph", [("A", "si2"), ("B", "si2"), ("A", "cell9")], ids=["si2_setA", "si2_setB", "cell9_mini"])
def test_small_graph_forward_is_bit_reproducible(which, graph):
    """r6: BASELINE config #1 (and the 9-atom cell GPUTEST_r05 went red on) evaluated four times eagerly: node rows, edge rows and Hamiltonian blocks agree
    BIT FOR BIT.  Every edge launch of these crystals is a split launch with private tile copies per wave -- their work is dealt statically since r6."""
    r = G.check_small_graph_forward_reproducible(which=which, graph=graph)
    print(r)
    assert r["parts"] != ["1"], r                              # the split launches are what is being tested
    assert r["node_max_abs_diff"] == 0.0 and r["edge_max_abs_diff"] == 0.0 and r["H_max_abs_diff"] == 0.0, r


def test_training_step_on_a_small_crystal_is_bit_reproducible():
    """the same guarantee for the training step of a 6-atom cell: split forward launches, data-gradient programs spread over several workgroups per tile"""
    r = G.check_training_step_reproducible(n_atoms=6)
    print(r)
    assert r["loss_diff"] == 0.0 and r["max_grad_diff"] == 0.0 and r["n_params"] > 60, r


def test_training_step_is_bit_reproducible():
    """the default model's training step has a fixed summation order everywhere: loss and every gradient bit-identical between two runs"""
    r = G.check_training_step_reproducible()
    print(r)
    assert r["loss_diff"] == 0.0 and r["max_grad_diff"] == 0.0 and r["n_params"] > 60, r


@pytest.mark.gpu
def test_edge_kernel_is_not_disturbed_by_a_co_running_half_precision_mfma_kernel():
    """profiles/r06_tp_is.md section 8 (a gfx950 interaction between packed fp32 VALU instructions and the 16x16x32 f16 / bf16 MFMAs of another wave; the library is
    built without the former).  Bit-exact, and the aggressor must really have overlapped the launches."""
    r = G.check_edge_kernel_next_to_half_precision_mfma_kernel()
    if "skipped" in r:
        pytest.skip(r["skipped"])
    for name in ("fp32_mfma_control", "f16_16x16x32_chains", "f16_16x16x32_independent", "bf16_16x16x32_chains"):
        assert r[name]["launches_overlapped"] >= 3, r
        assert r[name]["wrong_tiles"] == 0, r


@pytest.mark.gpu
def test_split_radial_scale_twins_refilled_on_the_device_equal_the_planner():
    """hg_w3_split_refill vs plan/program.py:w3_split_fill on a set-A MessagePackBlock: bit for bit"""
    r = G.check_w3_twins_device_refill()
    assert r["regions"] > 100 and r["twin_dwords"] > 100000, r
    assert r["dwords_different"] == 0 and r["wiped_dwords_left"] == 0, r
    assert abs(r["maxabs_device"] - r["maxabs_host"]) <= 1e-6 * r["maxabs_host"] and not r["split_off_after_check"], r

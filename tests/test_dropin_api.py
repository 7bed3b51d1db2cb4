"""The drop-in boundary without a GPU: the reference's import paths resolve to the MI355X modules, build_hamgnn_model follows the
reference's call sequence (hamgnn/main.py:178-263), checkpoints in the reference's key layout load verified, the Uni-HamGNN pickle
is ingested with stub classes, attribute-style (non-dict) graph objects are accepted by the index plumbing."""
import io
import pickle
import sys
import types

import numpy as np
import pytest
import torch
from torch import nn

MINI, SH = "8x0e+4x0o+4x1o+2x1e+2x2o+3x2e+2x3o", "0e+1o+2e+3o"


class _NS(dict):                                               # EasyDict stand-in: attribute + key access
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _config(soc=False, nao=19):
    pre = _NS(num_types=96, irreps_edge_sh=SH, edge_sh_normalization="component", edge_sh_normalize=True, build_internal_graph=False,
              cutoff=26.0, rbf_func="bessel", num_radial=8, num_layers=2, irreps_node_features=MINI, use_kan=False, radial_MLP=[16, 16],
              correlation=2, num_hidden_features=4, use_corr_prod=False)
    out = _NS(nao_max=nao, ham_type="openmx", ham_only=True, symmetrize=True, calculate_band_energy=False, num_k=4, k_path=None,
              band_num_control=None, soc_switch=soc, nonlinearity_type="gate", add_H0=True, spin_constrained=False, collinear_spin=False,
              minMagneticMoment=0.5)
    return _NS(setup=_NS(GNN_Net="HamGNN_pre", property="hamiltonian"), representation_nets=_NS(HamGNN_pre=pre), output_nets=_NS(HamGNN_out=out))


def test_reference_import_paths_resolve_to_hip_modules():
    from hamgnn.models.hamgnn_conv import HamGNNConvE3
    from hamgnn.models.hamgnn_output import HamGNNPlusPlusOut
    from hamgnn.models.Model import Model
    from hamgnn.main import Model as Model2, build_hamgnn_model
    from hamgnn.data.graph_data import NPZGraphDataset, LMDBGraphDataset  # noqa: F401
    import hamgnn_amd.models.hamgnn_conv as hc
    assert HamGNNConvE3 is hc.HamGNNConvE3 and Model is Model2
    cfg = _config()
    rep, out, post = build_hamgnn_model(cfg)
    assert isinstance(rep, HamGNNConvE3) and isinstance(out, HamGNNPlusPlusOut) and post is None
    assert cfg.representation_nets.HamGNN_pre.radius_type == "openmx"          # main.py:208
    assert out.zero_point_shift is True and out.soc_basis == "so3"              # initialize_output_parameters defaults
    assert str(rep.irreps_node_features) == MINI
    m = Model(representation=rep, output=out, losses=None, validation_metrics=None, lr=1e-3, lr_decay=0.5, lr_patience=5, post_processing=None)
    assert set(k.split(".")[0] for k in m.state_dict()) == {"representation", "output_module"}
    cfg.setup.GNN_Net = "SomethingElse"
    with pytest.raises(SystemExit):
        build_hamgnn_model(cfg)


def test_missing_use_corr_prod_defaults_to_true_like_the_reference():
    from hamgnn.main import build_hamgnn_model
    cfg = _config()
    del cfg.representation_nets.HamGNN_pre["use_corr_prod"]
    rep, _, _ = build_hamgnn_model(cfg)
    assert rep.use_corr_prod and hasattr(rep, "corr_products")                  # main.py:216-217


def test_return_forces_is_accepted_and_carried_like_the_reference():
    """HamGNNPlusPlusOut(return_forces=True) only sets `derivative` (hamgnn_output.py:127), the Model's `requires_derivatives` (Model.py:103): the
    reference computes no force anywhere, so the flag must not be an error and must not change the modules that are built"""
    from hamgnn.models.hamgnn_output import HamGNNPlusPlusOut
    from hamgnn.models.hamgnn_conv import HamGNNConvE3
    from hamgnn.models.Model import Model
    cfg = _config()
    kw = dict(irreps_in_node=MINI, irreps_in_edge=MINI, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True, soc_switch=False)
    a, b = HamGNNPlusPlusOut(return_forces=True, create_graph=True, **kw), HamGNNPlusPlusOut(**kw)
    assert a.derivative is True and b.derivative is False
    assert [k for k, _ in a.named_parameters()] == [k for k, _ in b.named_parameters()]
    m = Model(HamGNNConvE3(cfg.representation_nets), a)
    assert m.requires_derivatives is True
    with pytest.raises(NotImplementedError):                   # what is NOT built still fails loudly at construction
        HamGNNPlusPlusOut(spin_constrained=True, **kw)


def test_transformer_backbone_through_build_hamgnn_model_and_verified_loading():
    """GNN_Net: HamGNNTransformer (main.py:219-220): module tree / parameter names of the reference, weights load verified"""
    from hamgnn.main import build_hamgnn_model
    from hamgnn.models.hamgnn_transformer import HamGNNTransformer
    from hamgnn_amd.models.model import load_reference_state_dict
    from oracle import hamgnn_ref as R
    cfg = _config()
    cfg.setup.GNN_Net = "HamGNNTransformer"
    cfg.representation_nets.HamGNN_pre.update(irreps_node_features="8x0e+4x0o+4x1o+2x1e+2x2o+4x2e+2x3o", num_heads=2)
    rep, out, _ = build_hamgnn_model(cfg)
    assert isinstance(rep, HamGNNTransformer) and len(rep.orb_transformers) == 2 and len(rep.corr_products) == 2
    ref = R.HamGNNTransformer(dict(cfg.representation_nets.HamGNN_pre))
    theirs = {k for k, _ in ref.named_parameters()}
    ours = {k for k, _ in rep.named_parameters()}
    assert theirs == ours, sorted(theirs ^ ours)[:6]
    assert "orb_transformers.0.cutoff_func.cut_param" in ours and "orb_transformers.1.linear_query.weight" in ours
    sd = {k: torch.randn_like(v) for k, v in ref.state_dict().items()}
    load_reference_state_dict(rep, sd)
    assert torch.equal(rep.orb_transformers[1].linear_key.weight, sd["orb_transformers.1.linear_key.weight"].float())
    bad = dict(sd)
    bad["orb_transformers.0.linear_value.weight"] = bad.pop("orb_transformers.0.linear_key.weight")      # renamed parameter
    with pytest.raises(KeyError):
        load_reference_state_dict(rep, bad)
    cfg.representation_nets.HamGNN_pre.num_heads = 3                             # 8x0e cannot be split over three heads
    with pytest.raises(ValueError):
        build_hamgnn_model(cfg)


def test_attention_head_table():
    from hamgnn_amd import plan as P
    tab, hd = P.attention_head_table("8x0e+4x1o+2x2e", 2)
    lay = P.PlanarLayout("8x0e+4x1o+2x2e")
    assert hd == 4 + 2 * 3 + 1 * 5 and tab.shape == (lay.dim,)
    assert tab[:8].tolist() == [0] * 4 + [1] * 4
    o1 = lay.off[1]
    for a in range(3):
        assert tab[o1 + a * lay.mulp[1]:o1 + a * lay.mulp[1] + 4].tolist() == [0, 0, 1, 1]
    o2 = lay.off[2]
    assert tab[o2:o2 + lay.mulp[2]].tolist() == [0, 1] + [-1] * (lay.mulp[2] - 2)
    assert (tab >= 0).sum() == 8 + 12 + 10


def _fake_reference_checkpoint(model, extra=None):
    sd = {k: torch.randn_like(v) for k, v in model.state_dict().items()}
    sd["representation.radial_basis_functions.freqs"] = torch.arange(8.0)       # buffers the reference keeps in its state_dict
    sd["representation.cutoff_func.cutoff"] = torch.tensor([26.0])
    sd["output_module.cg_calculator.cg_1_1_2"] = torch.zeros(3, 3, 5)
    sd["representation.convolutions.0.skip_linear.output_mask"] = torch.ones(4)
    sd.update(extra or {})
    return sd


def test_load_from_checkpoint_lightning_layout(tmp_path):
    from hamgnn.main import Model, build_hamgnn_model
    rep, out, _ = build_hamgnn_model(_config())
    proto = Model(representation=rep, output=out)
    sd = _fake_reference_checkpoint(proto)
    p = str(tmp_path / "last.ckpt")
    torch.save({"state_dict": sd, "epoch": 3, "hyper_parameters": {"lr": 1e-3}}, p)
    rep2, out2, _ = build_hamgnn_model(_config())
    m = Model.load_from_checkpoint(checkpoint_path=p, representation=rep2, output=out2, post_processing=None, losses=None,
                                   validation_metrics=None, lr=None, lr_decay=None, lr_patience=None)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # a missing parameter must fail loudly (strict=False would leave randn weights in place)
    bad = dict(sd)
    del bad["representation.convolutions.1.conv_tp.node_tensor_product.weight"]
    torch.save({"state_dict": bad}, p)
    with pytest.raises(KeyError, match="lacks"):
        Model.load_from_checkpoint(checkpoint_path=p, representation=rep2, output=out2)
    # a learned tensor the model has no slot for must fail as well
    torch.save({"state_dict": dict(sd, **{"representation.kan_layer.spline_weight": torch.zeros(3)})}, p)
    with pytest.raises(KeyError, match="no slot"):
        Model.load_from_checkpoint(checkpoint_path=p, representation=rep2, output=out2)


def test_uni_hamgnn_pickle_ingestion_with_stub_classes(tmp_path):
    """a predictor pickled the way Uni-HamiltonianPredictor.py:80-82 does (whole nn.Module trees of hamgnn / e3nn classes, EasyDict
    configs) loads WITHOUT those packages: classes are stubbed, tensors are real, parameters land in the HIP modules."""
    from hamgnn_amd import uni
    from hamgnn_amd.models.model import Model
    cfgs = {False: _config(False, 26), True: _config(True, 26)}
    protos = {}
    for soc, cfg in cfgs.items():
        rep, head = uni.build_hamgnn_components(cfg)
        protos[soc] = Model(representation=rep, output=head)
    # --- producer side: fake `hamgnn` / `e3nn` / `easydict` modules whose classes hold the same parameter tree
    names = ["HamGNN_v_2_1", "HamGNN_v_2_1.models", "HamGNN_v_2_1.models.Model", "e3nn", "e3nn.o3", "easydict", "uni_predictor_main"]
    mods = {n: types.ModuleType(n) for n in names}

    def mk(modname, clsname, base):
        cls = type(clsname, (base,), {"__module__": modname})
        setattr(mods[modname], clsname, cls)
        return cls
    LegacyModel = mk("HamGNN_v_2_1.models.Model", "Model", nn.Module)
    E3Lin = mk("e3nn.o3", "Linear", nn.Module)
    EasyDict = mk("easydict", "EasyDict", dict)
    Pred = mk("uni_predictor_main", "HamiltonianPredictor", object)

    def clone_tree(m):
        """same parameter tree, every node an instance of a class from the fake packages (+ a buffer the loader must ignore)"""
        node = E3Lin() if not isinstance(m, Model) else LegacyModel()
        nn.Module.__init__(node)
        for k, p in m._parameters.items():
            node.register_parameter(k, nn.Parameter(torch.randn_like(p)))
        if m._parameters:
            node.register_buffer("output_mask", torch.ones(3))
        for k, c in m._modules.items():
            node.add_module(k, clone_tree(c))
        return node

    def easy(d):
        e = EasyDict()
        for k, v in d.items():
            e[k] = easy(v) if isinstance(v, dict) else v
        return e
    pred = Pred()
    pred.soc_enabled, pred.device = True, "cuda:0"
    pred.non_soc_model, pred.soc_model = clone_tree(protos[False]), clone_tree(protos[True])
    pred.config_nonsoc, pred.config_soc = easy(cfgs[False]), easy(cfgs[True])
    sys.modules.update(mods)
    try:
        p = str(tmp_path / "uni.pkl")
        with open(p, "wb") as f:
            pickle.dump(pred, f)
        want = {False: pred.non_soc_model.state_dict(), True: pred.soc_model.state_dict()}
    finally:
        for n in names:
            sys.modules.pop(n, None)
    with pytest.raises(Exception):                             # the stock unpickler cannot resolve the producer's packages here
        with open(p, "rb") as f:
            pickle.load(f)
    got = uni.load_model_predictor(p)
    assert got.soc_enabled and got.soc_model.output_module.add_H_nonsoc and not got.soc_model.output_module.zero_point_shift
    assert got.non_soc_model.representation.legacy_edge_update and not got.non_soc_model.representation.use_corr_prod
    for soc, model in ((False, got.non_soc_model), (True, got.soc_model)):
        for k, v in model.state_dict().items():
            assert torch.equal(v, want[soc][k]), k


def test_stub_unpickler_runs_no_foreign_code(tmp_path):
    from hamgnn_amd import uni

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > /tmp/hg_pwned",))
    buf = io.BytesIO(pickle.dumps({"x": Evil()}))
    out = uni.stub_load(buf)
    import os
    assert not os.path.exists("/tmp/hg_pwned") and isinstance(out["x"], uni._Bag)


def test_graph_npz_loader_refuses_foreign_globals(tmp_path):
    from hamgnn_amd.data import graph_data as GD

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > /tmp/hg_pwned2",))
    p = str(tmp_path / "graph_data.npz")
    np.savez(p, graph=np.array({0: Evil()}, dtype=object))
    with pytest.raises(pickle.UnpicklingError, match="allow-list"):
        GD.load_graph_npz(p)


class _AttrGraph:
    """attribute-style graph object (what torch_geometric Data looks like to the model code): no dict base class"""

    def __init__(self, **k):
        self.__dict__.update(k)

    def __getitem__(self, k):
        return self.__dict__[k]

    def __setitem__(self, k, v):
        self.__dict__[k] = v

    def __contains__(self, k):
        return k in self.__dict__


def test_topology_cache_on_attribute_style_graphs_and_invalidation():
    from hamgnn_amd.data import synthetic as S
    from hamgnn_amd.topo import get_topology
    g = S.si_diamond(primitive=True)
    a = _AttrGraph(**{k: v for k, v in g.items()})
    t = get_topology(a)
    rowptr, perm = t.receiver_csr()
    assert int(rowptr[-1]) == g.num_edges and torch.equal(torch.sort(g.edge_index[1][perm]).values, g.edge_index[1][perm])
    assert get_topology(a) is t                                 # cached on the object
    a.edge_index = a.edge_index.clone()                         # replaced tensor -> rebuilt
    t2 = get_topology(a)
    assert t2 is not t
    a.edge_index[0, 0] = a.edge_index[0, 0]                     # in-place write bumps the version counter -> rebuilt
    assert get_topology(a) is not t2
    # the cache never travels to derived graphs
    get_topology(g)
    assert "_hg_topology" in g and "_hg_topology" not in g.to("cpu")
    # z outside [0, num_types) is refused (the reference's one-hot raises; the device tables would be read out of bounds)
    g2 = S.si_diamond(primitive=True)
    g2["z"] = torch.tensor([14, 120])
    with pytest.raises(ValueError, match="num_types"):
        get_topology(g2).check_num_types(96)


def test_checkpoint_round_trip(tmp_path):
    """Model.save_checkpoint writes the reference's Lightning key layout; Model.load_from_checkpoint (verified loading) reads it back"""
    import torch
    from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
    from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
    from hamgnn_amd.models.model import Model, read_checkpoint_state_dict
    mini = "8x0e+4x0o+4x1o+2x1e+2x2o+3x2e+2x3o"
    cfg = dict(num_types=20, irreps_edge_sh="0e+1o+2e+3o", edge_sh_normalization="component", edge_sh_normalize=True, build_internal_graph=False,
               cutoff=26.0, rbf_func="bessel", num_radial=8, num_layers=2, irreps_node_features=mini, use_kan=False, radial_MLP=[16, 16],
               correlation=2, num_hidden_features=4, use_corr_prod=True)
    mk = lambda: dict(representation=HamGNNConvE3(cfg), output=HamGNNPlusPlusOut(mini, mini, nao_max=19, ham_type="openmx", ham_only=True, soc_switch=False))
    torch.manual_seed(0)
    a = Model(**mk())
    path = a.save_checkpoint(str(tmp_path / "last.ckpt"), epoch=3)
    sd = read_checkpoint_state_dict(path)
    assert all(k.startswith(("representation.", "output_module.")) for k in sd) and len(sd) == len(a.state_dict())
    torch.manual_seed(1)
    b = Model.load_from_checkpoint(path, **mk())
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka


def test_precision_64_raises_instead_of_running_fp32():
    """the reference's `precision: 64` flow (main.py:469-474: set_default_dtype(float64), model.to(float64), float64 graphs) is not built; it must
    raise -- on the backbone and on the head, before any tensor is cast down -- rather than return fp32-accurate rows"""
    from hamgnn.main import build_hamgnn_model
    from hamgnn_amd.data import synthetic as S
    rep, out, _ = build_hamgnn_model(_config())
    g = S.random_cell(4, [14, 8], seed=0, density=0.004)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        with pytest.raises(NotImplementedError, match="precision: 64"):
            rep(g)
        with pytest.raises(NotImplementedError, match="precision: 64"):
            out(g, {"node_attr": None, "edge_attr": None})
    finally:
        torch.set_default_dtype(prev)
    rep64, out64, _ = build_hamgnn_model(_config())
    rep64.double()
    out64.double()
    with pytest.raises(NotImplementedError, match="parameters are float64"):
        rep64(g)
    with pytest.raises(NotImplementedError, match="parameters are float64"):
        out64(g, {"node_attr": None, "edge_attr": None})
    g64 = S.random_cell(4, [14, 8], seed=0, density=0.004)
    g64.pos = g64.pos.double()
    with pytest.raises(NotImplementedError, match="data.pos is float64"):
        rep(g64)

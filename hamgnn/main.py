"""The two names the hot path's callers import from the reference's hamgnn/main.py: `Model` (Uni-HamGNN/Uni-HamiltonianPredictor.py:16)
and `build_hamgnn_model(config)` (hamgnn/main.py:178-263: representation + output module from the parsed YAML; same defaults, same
SystemExit(1) for unknown network / property names).  The CLI / trainer around them is the reference's training harness (out of scope)."""
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
from hamgnn_amd.models.hamgnn_transformer import HamGNNTransformer
from hamgnn_amd.models.model import Model  # noqa: F401


def _get(c, k, d=None):
    return c.get(k, d) if isinstance(c, dict) else getattr(c, k, d)


def _has(c, k):
    return (k in c) if isinstance(c, dict) else hasattr(c, k)


def _set(c, k, v):
    if isinstance(c, dict):
        c[k] = v
    else:
        setattr(c, k, v)


# defaults of initialize_output_parameters (hamgnn/main.py:107-112: add_H_nonsoc, get_nonzero_mask_tensor, zero_point_shift=True,
# soc_basis) and of the config parser for keys a hand-written config may omit (config/config_parsing.py:22-119)
_OUT_DEFAULTS = {"add_H_nonsoc": False, "get_nonzero_mask_tensor": False, "zero_point_shift": True, "soc_basis": "so3",
                 "spin_constrained": False, "collinear_spin": False, "minMagneticMoment": 0.5, "calculate_band_energy": False,
                 "num_k": 8, "k_path": None, "band_num_control": None, "nonlinearity_type": "gate", "symmetrize": True, "ham_only": True}


def build_hamgnn_model(config):
    rep_cfg = _get(config, "representation_nets")
    pre = _get(rep_cfg, "HamGNN_pre")
    out = _get(_get(config, "output_nets"), "HamGNN_out")
    setup = _get(config, "setup")
    _set(pre, "radius_type", str(_get(out, "ham_type")).lower())
    net = str(_get(setup, "GNN_Net")).lower()
    if net in ("hamgnnconv", "hamgnnpre", "hamgnn_pre"):
        if not _has(pre, "use_corr_prod"):
            _set(pre, "use_corr_prod", True)
        graph_representation = HamGNNConvE3(rep_cfg)
    elif net == "hamgnntransformer":
        graph_representation = HamGNNTransformer(rep_cfg)
    else:
        print(f"The network: {_get(setup, 'GNN_Net')} is not yet supported!")
        raise SystemExit(1)
    if str(_get(setup, "property")).lower() != "hamiltonian":
        print(f'Property type "{str(_get(setup, "property")).lower()}" is not supported!')
        raise SystemExit(1)
    p = {k: _get(out, k, d) for k, d in _OUT_DEFAULTS.items()}
    output_module = HamGNNPlusPlusOut(
        irreps_in_node=graph_representation.irreps_node_features, irreps_in_edge=graph_representation.irreps_node_features,
        nao_max=_get(out, "nao_max"), ham_type=_get(out, "ham_type"), ham_only=p["ham_only"], symmetrize=p["symmetrize"],
        calculate_band_energy=p["calculate_band_energy"], num_k=p["num_k"], k_path=p["k_path"], band_num_control=p["band_num_control"],
        soc_switch=_get(out, "soc_switch", False), soc_basis=p["soc_basis"], nonlinearity_type=p["nonlinearity_type"],
        add_H0=_get(out, "add_H0", False), spin_constrained=p["spin_constrained"], collinear_spin=p["collinear_spin"],
        minMagneticMoment=p["minMagneticMoment"], add_H_nonsoc=p["add_H_nonsoc"], get_nonzero_mask_tensor=p["get_nonzero_mask_tensor"],
        zero_point_shift=p["zero_point_shift"])
    return graph_representation, output_module, None

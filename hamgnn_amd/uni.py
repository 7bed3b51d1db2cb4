"""Uni-HamGNN wiring on the MI355X path (BASELINE config #5): the two-model chain of
Uni-HamGNN/Uni-HamiltonianPredictor.py and the ingestion of its pickled predictor.

* `build_hamgnn_components(config)`  -- :34-77: the (representation, output) pair with the universal model's forced settings
  (`use_corr_prod=False`, `legacy_edge_update=True`, `add_H_nonsoc = soc_switch`, `zero_point_shift = not soc_switch`,
  `get_nonzero_mask_tensor=True`).
* `uni_forward(non_soc_model, soc_model, non_soc_batch, soc_batch)` -- `_model_forward` :290-319: the non-SOC prediction is split
  into `Hon_nonsoc` / `Hoff_nonsoc` (:306-311, one crystal per batch as the reference's DataLoader(batch_size=1) :269-275) and
  handed to the SOC model, whose so3 head only adds the xi L spin blocks (hamgnn_output.py:3026-3049).
* `load_model_predictor(path)` -- :96-137 without e3nn / Lightning / the reference package: the pickle is read with STUB classes
  (every global outside torch's tensor machinery becomes an inert attribute bag; nothing of the pickle's own code runs), the two
  models' parameters are collected by walking the stubbed nn.Module trees, and loaded -- verified key by key -- into HIP modules
  built from the configs stored in the same pickle.  Legacy module aliases (`HamGNN_v_2_1.*`, :85-93) need no registration here
  because class identity is never used.
"""
from __future__ import annotations

import io
import pickle
from typing import Dict, Optional, Tuple

import torch

from .models.hamgnn_conv import HamGNNConvE3
from .models.hamgnn_output import HamGNNPlusPlusOut
from .models.model import Model, load_reference_state_dict
from .topo import gget, ghas


# ------------------------------------------------------------------------------------------------ the chain
def _cfg(c, k, d=None):
    if isinstance(c, dict):
        return c.get(k, d)
    return getattr(c, k, d)


def build_hamgnn_components(config) -> Tuple[torch.nn.Module, torch.nn.Module]:
    rep_cfg = _cfg(config, "representation_nets")
    pre = _cfg(rep_cfg, "HamGNN_pre")
    out = _cfg(_cfg(config, "output_nets"), "HamGNN_out")
    forced = {"radius_type": str(_cfg(out, "ham_type")).lower(), "ham_type": str(_cfg(out, "ham_type")).lower(),
              "nao_max": _cfg(out, "nao_max"), "use_corr_prod": False, "legacy_edge_update": True}
    for k, v in forced.items():
        if isinstance(pre, dict):
            pre[k] = v
        else:
            setattr(pre, k, v)
    rep = HamGNNConvE3(rep_cfg)
    soc = bool(_cfg(out, "soc_switch"))
    head = HamGNNPlusPlusOut(
        irreps_in_node=rep.irreps_node_features, irreps_in_edge=rep.irreps_node_features, nao_max=_cfg(out, "nao_max"),
        ham_type=_cfg(out, "ham_type"), ham_only=_cfg(out, "ham_only", True), symmetrize=_cfg(out, "symmetrize", True),
        calculate_band_energy=_cfg(out, "calculate_band_energy", False), num_k=_cfg(out, "num_k", 8), k_path=_cfg(out, "k_path"),
        band_num_control=_cfg(out, "band_num_control"), soc_switch=soc, nonlinearity_type=_cfg(out, "nonlinearity_type", "gate"),
        add_H0=_cfg(out, "add_H0", True), spin_constrained=_cfg(out, "spin_constrained", False),
        collinear_spin=_cfg(out, "collinear_spin", False), minMagneticMoment=_cfg(out, "minMagneticMoment", 0.5),
        add_H_nonsoc=soc, zero_point_shift=not soc, get_nonzero_mask_tensor=True)
    return rep, head


def _ensure_hamiltonian_key(model, batch, keys):
    if ghas(batch, "hamiltonian"):
        return
    if model.output_module.zero_point_shift:
        batch["hamiltonian"] = torch.cat([batch[k] for k in keys])
    else:
        batch["hamiltonian"] = 0.0


def uni_forward(non_soc_model, soc_model, non_soc_batch, soc_batch=None) -> dict:
    """one crystal -- or a batch of crystals -- through the universal model(s); mutates the batches exactly where the reference does"""
    _ensure_hamiltonian_key(non_soc_model, non_soc_batch, ("Hon", "Hoff"))
    if soc_model is None:
        return non_soc_model(non_soc_batch)
    _ensure_hamiltonian_key(soc_model, soc_batch, ("Hon", "Hoff", "iHon", "iHoff"))
    pred = non_soc_model(non_soc_batch)
    # the reference runs DataLoader(batch_size=1) and splits [on-site; off-site] of that ONE crystal (:306-311); a batch of several crystals
    # (the same crystals in the same order in both batches) carries the rows per crystal, so they are split back with the head's own inverse
    # of concatenate_hamiltonians_by_crystal -- one launch sequence for the whole batch instead of one per crystal
    head = non_soc_model.output_module
    edge_counts = head._global_inverse(non_soc_batch)[1]
    on, off = head._split_by_crystal(non_soc_batch, pred["hamiltonian"], edge_counts)
    if on.shape[0] != len(soc_batch.z) or off.shape[0] != soc_batch.edge_index.shape[1]:
        raise ValueError("uni_forward: the non-SOC and the SOC batch must hold the same crystals in the same order")
    soc_batch["Hon_nonsoc"] = on
    soc_batch["Hoff_nonsoc"] = off
    return soc_model(soc_batch)


# ------------------------------------------------------------------------------------------------ stubbed unpickling
class _Bag:
    """inert stand-in for any class named in a pickle: keeps constructor args / state, runs no code"""

    def __init__(self, *a, **k):
        self.__dict__["_args"], self.__dict__["_kwargs"] = a, k

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[1], dict):      # (dict, slots)
            if isinstance(state[0], dict):
                self.__dict__.update(state[0])
            self.__dict__.update(state[1])
        else:
            self.__dict__["_state"] = state

    def __call__(self, *a, **k):                               # e.g. functools.partial-like reducers
        return _Bag(*a, **k)

    # dict-like bags (EasyDict subclasses dict: pickled via copyreg with the dict items set through __setitem__)
    def __setitem__(self, k, v):
        self.__dict__.setdefault("_items", {})[k] = v

    def __getitem__(self, k):
        return self.__dict__["_items"][k]

    def get(self, k, d=None):
        return self.__dict__.get("_items", {}).get(k, self.__dict__.get(k, d))

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        items = self.__dict__.get("_items", {})
        if k in items:
            return items[k]
        raise AttributeError(k)


_TENSOR_GLOBALS = {
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch", "Size"), ("torch", "device"),
    ("torch._tensor", "_rebuild_from_type_v2"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
    ("collections", "OrderedDict"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"),
    ("builtins", "frozenset"), ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"),
    ("builtins", "complex"), ("builtins", "slice"), ("builtins", "range"), ("_codecs", "encode"),
    ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
}


class _StubUnpickler(pickle.Unpickler):
    """tensors, containers and numpy arrays are real; EVERY other global is a _Bag subclass (no foreign code runs)"""

    def __init__(self, f, storages=None):
        super().__init__(f)
        self._storages = storages

    def find_class(self, module, name):
        if (module, name) in _TENSOR_GLOBALS:
            return super().find_class(module, name)
        if module == "torch" and (name.endswith("Storage") or name in {"float16", "float32", "float64", "bfloat16", "int8", "uint8", "int16",
                                                                       "int32", "int64", "bool", "complex64", "complex128"}):
            return getattr(torch, name)
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return lambda b: torch.load(io.BytesIO(b), weights_only=True)      # the stock helper unpickles the bytes without restrictions
        if module == "copyreg" and name == "_reconstructor":
            return lambda cls, base, state: cls()              # new-style reduce of a dict/object subclass -> empty bag, state follows
        return type(name, (_Bag,), {"__module__": module})

    def persistent_load(self, pid):                            # torch.save zip format: ('storage', type, key, location, numel)
        if self._storages is None:
            raise pickle.UnpicklingError("persistent id outside a torch zip archive")
        return self._storages(pid)


def stub_load(f, torch_zip: bool = False):
    """Unpickle with stub classes.  torch_zip: the file is a torch.save zip archive (Lightning .ckpt); else a plain pickle stream whose
    tensors were pickled by value (pickle.dump of an object holding nn.Modules, Uni-HamiltonianPredictor.py:80-82)."""
    if not torch_zip:
        return _StubUnpickler(f).load()
    import zipfile
    zf = zipfile.ZipFile(f)
    root = zf.namelist()[0].split("/")[0]
    cache = {}

    def storages(pid):
        _, stype, key, _loc, numel = pid
        if key not in cache:
            raw = zf.read(f"{root}/data/{key}")
            dtype = stype.dtype if hasattr(stype, "dtype") else torch.uint8
            cache[key] = torch.frombuffer(bytearray(raw), dtype=dtype).untyped_storage() if len(raw) else torch.UntypedStorage(0)
        return torch.storage.TypedStorage(wrap_storage=cache[key], dtype=stype.dtype if hasattr(stype, "dtype") else torch.uint8, _internal=True)
    return _StubUnpickler(io.BytesIO(zf.read(f"{root}/data.pkl")), storages).load()


def module_state_dict(stub, prefix="") -> Dict[str, torch.Tensor]:
    """state_dict of a stubbed nn.Module tree (walks `_parameters`, `_buffers`, `_modules` like nn.Module.state_dict)"""
    out = {}
    d = getattr(stub, "__dict__", {})
    for kind in ("_parameters", "_buffers"):
        for k, v in (d.get(kind) or {}).items():
            if torch.is_tensor(v):
                out[prefix + k] = v.detach()
    for k, m in (d.get("_modules") or {}).items():
        if m is not None:
            out.update(module_state_dict(m, prefix + k + "."))
    return out


def _bag_to_dict(b):
    """EasyDict-style config bags -> plain nested dicts"""
    if isinstance(b, _Bag):
        items = dict(b.__dict__.get("_items", {}))
        for k, v in b.__dict__.items():
            if not k.startswith("_"):
                items.setdefault(k, v)
        return {k: _bag_to_dict(v) for k, v in items.items()}
    if isinstance(b, dict):
        return {k: _bag_to_dict(v) for k, v in b.items()}
    if isinstance(b, (list, tuple)):
        return type(b)(_bag_to_dict(v) for v in b)
    return b


class HamiltonianPredictor:
    """the loaded universal predictor: `.non_soc_model`, `.soc_model` (HIP `Model`s), `.soc_enabled`, `.predict(batch[, soc_batch])`"""

    def __init__(self, non_soc_model, soc_model=None, device="cuda"):
        self.non_soc_model, self.soc_model, self.soc_enabled, self.device = non_soc_model, soc_model, soc_model is not None, device

    def predict(self, non_soc_batch, soc_batch=None):
        self.non_soc_model.output_module.zero_point_shift = False      # predict_hamiltonians(calculate_mae=False), :247-251
        with torch.no_grad():
            return uni_forward(self.non_soc_model, self.soc_model if self.soc_enabled else None, non_soc_batch, soc_batch)


def load_model_predictor(model_filepath: str, device: Optional[str] = None) -> HamiltonianPredictor:
    with open(model_filepath, "rb") as f:
        stub = stub_load(f)
    d = stub.__dict__

    def rebuild(model_stub, cfg_bag):
        cfg = _bag_to_dict(cfg_bag)
        rep, head = build_hamgnn_components(cfg)
        sd = module_state_dict(model_stub)
        load_reference_state_dict(rep, sd, prefix="representation.")
        load_reference_state_dict(head, sd, prefix="output_module.")
        return Model(representation=rep, output=head, losses=None, validation_metrics=None, lr=None, lr_decay=None, lr_patience=None)
    non_soc = rebuild(d["non_soc_model"], d["config_nonsoc"])
    soc = rebuild(d["soc_model"], d["config_soc"]) if d.get("soc_model") is not None and d.get("soc_enabled", True) else None
    pred = HamiltonianPredictor(non_soc, soc, device or d.get("device", "cuda"))
    if device:
        pred.non_soc_model = pred.non_soc_model.to(device)
        if pred.soc_model is not None:
            pred.soc_model = pred.soc_model.to(device)
    return pred

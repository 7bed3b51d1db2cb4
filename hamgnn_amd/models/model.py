"""Model -- Lightning-free mirror of the reference's training wrapper as far as the hot path needs it
(hamgnn/models/Model.py:63-82 ctor, :359-376 forward): ``forward(batch) = output_module(batch, representation(batch))`` with the
sub-module names ``representation`` / ``output_module`` that Lightning checkpoints prefix their keys with (Model.py:81-82).
``Model.load_from_checkpoint(checkpoint_path=..., representation=..., output=..., **ignored)`` has the call shape of
hamgnn/main.py:374-377, :527-537 and Uni-HamGNN/Uni-HamiltonianPredictor.py:213-225.  Losses / optimiser / logging are the
training harness (out of scope) and are accepted but unused."""
from __future__ import annotations

import pickle
from typing import Dict, Iterable, Optional

import torch
from torch import nn

# Non-learned buffers the reference / e3nn 0.5.0 keep in their state_dicts (constants a loader may drop): e3nn `output_mask` of
# o3.Linear / o3.TensorProduct and the w3j constants of the code-generated sub-modules; BesselBasis.freqs (utils/basis_functions.py:190),
# CosineCutoff.cutoff (utils/cutoff_functions.py:48), GaussianSmearing.offset (:216), ClebschGordanCoefficients.cg_* (physics/
# Clebsch_Gordan_coefficients.py:27), the generalized-CG tensors U_matrix_{nu} of the MACE symmetric contraction (toolbox/mace/modules/
# symmetric_contraction.py: register_buffer) and AttentionBlockE3.max_radius.  Anything else the model has no slot for is an error (a learned tensor would be lost silently).
_IGNORABLE_LAST = ("output_mask", "freqs", "cutoff", "offset")
_IGNORABLE_PREFIX_LAST = ("cg_", "_w3j", "w3j", "_big_w3j", "U_matrix_")


def _is_ignorable(key: str) -> bool:
    last = key.rsplit(".", 1)[-1]
    return last in _IGNORABLE_LAST or last.startswith(_IGNORABLE_PREFIX_LAST) or "_compiled" in key or "_codegen" in key


def load_reference_state_dict(module: nn.Module, state_dict: Dict[str, torch.Tensor], prefix: str = "", allow_unexpected: Iterable[str] = ()):
    """Load reference-named tensors into a HIP module and VERIFY the result (load_state_dict(strict=False) alone would leave a renamed
    or missing parameter at its random initial value, silently): every parameter of `module` must be covered with the right shape, and
    keys the module does not know must be non-learned e3nn / reference buffers.  Returns the list of ignored keys."""
    sd = {}
    for k, v in state_dict.items():
        if prefix and not k.startswith(prefix):
            continue
        sd[k[len(prefix):]] = v if torch.is_tensor(v) else torch.as_tensor(v)
    params = dict(module.named_parameters())
    missing = sorted(k for k in params if k not in sd)
    if missing:
        raise KeyError(f"checkpoint lacks {len(missing)} parameter(s) of {type(module).__name__}, e.g. {missing[:4]}")
    bad = sorted(k for k in params if tuple(sd[k].shape) != tuple(params[k].shape) and sd[k].numel() != params[k].numel())
    if bad:
        k = bad[0]
        raise ValueError(f"shape mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(params[k].shape)} (+{len(bad) - 1} more)")
    own = set(params) | set(dict(module.named_buffers()))
    unexpected = sorted(k for k in sd if k not in own)
    allow = tuple(allow_unexpected)
    # (e3nn keeps an EMPTY `weight` buffer on tensor products whose weights are external or absent: nothing to lose there)
    rogue = [k for k in unexpected if not _is_ignorable(k) and not k.startswith(allow) and sd[k].numel() > 0]
    if rogue:
        raise KeyError(f"checkpoint holds {len(rogue)} tensor(s) the model has no slot for, e.g. {rogue[:4]}")
    with torch.no_grad():
        for k, p in params.items():
            p.copy_(sd[k].to(p.dtype).reshape(p.shape))
    for m in module.modules():                                 # weights changed: packed MFMA fragments are stale
        if hasattr(m, "_compiled_for"):
            m._compiled_for = None
    return unexpected


def read_checkpoint_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """state_dict of a Lightning ``.ckpt`` (a torch.save'd dict with key 'state_dict') or of a bare torch.save'd state_dict.
    Tensors only (weights_only=True): nothing else in the file is executed."""
    try:
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:
        # Lightning checkpoints also pickle hyper-parameter objects (EasyDict ...): read them with permissive STUB classes, never the real ones
        from ..uni import stub_load
        with open(path, "rb") as f:
            obj = stub_load(f, torch_zip=True)
    if isinstance(obj, dict) and "state_dict" in obj:
        obj = obj["state_dict"]
    if not isinstance(obj, dict) or not all(torch.is_tensor(v) for v in obj.values()):
        raise ValueError(f"{path}: no state_dict found")
    return dict(obj)


class Model(nn.Module):
    def __init__(self, representation: nn.Module, output: nn.Module, losses=None, validation_metrics=None, lr: Optional[float] = 1e-3,
                 lr_decay: Optional[float] = 0.1, lr_patience: Optional[int] = 100, lr_monitor: str = "training/total_loss", epsilon: float = 1e-8,
                 beta1: float = 0.99, beta2: float = 0.999, amsgrad: bool = True, max_points_to_scatter: int = 100000, post_processing=None):
        super().__init__()
        self.representation = representation
        self.output_module = output
        self.losses, self.metrics = losses, validation_metrics
        self.lr, self.lr_decay, self.lr_patience, self.lr_monitor = lr, lr_decay, lr_patience, lr_monitor
        self.post_processing = post_processing
        # return_forces: the reference only enables autograd on batch.pos in its Lightning steps (Model.py:227, 285, 459-460) and computes no
        # force anywhere; the flag is carried for interface parity and changes nothing here either
        self.requires_derivatives = bool(getattr(self.output_module, "derivative", False))
        # forward() hands the representation to the output module and to nobody else (Model.py:459-465): the backbone may skip what that head never reads
        if hasattr(self.representation, "declare_consumer"):
            self.representation.declare_consumer(self.output_module)

    def forward(self, batch):
        representation = self.representation(batch)
        return self.output_module(batch, representation)

    @torch.no_grad()
    def test(self, batches, log_dir: Optional[str] = None, device=None):
        """The test stage of the reference (Model.test_step / test_epoch_end / _save_predictions_and_targets, Model.py:268-357, 548-567)
        without Lightning: forward every batch, collect `predictions[loss["prediction"].lower()]` and `batch[loss["target"].lower()]` for
        every loss entry that names a target (default: hamiltonian vs hamiltonian), concatenate over the batches and -- if `log_dir` is
        given -- write `prediction_{key}.npy` / `target_{key}.npy` there, as the reference's `stage: test` does.  Returns
        ({prediction key: array}, {target key: array}).  `batches`: an iterable of graph batches (hamgnn_amd.data.collate output)."""
        import os
        import numpy as np
        from ..topo import gget
        pairs = [(d["prediction"], d["target"]) for d in (self.losses or [{"prediction": "hamiltonian", "target": "hamiltonian"}])
                 if (d.get("target") if hasattr(d, "get") else "target" in d)]
        preds = {p: [] for p, _ in pairs}
        targets = {t: [] for _, t in pairs}
        for batch in batches:
            if device is not None:
                batch = batch.to(device)
            out = self(batch)
            for pk, tk in pairs:
                preds[pk].append(out[pk.lower()].detach().float().cpu().numpy())
                tgt = gget(batch, tk.lower())
                if tgt is not None:
                    targets[tk].append(tgt.detach().float().cpu().numpy())
        preds = {k: np.concatenate(v) for k, v in preds.items() if v}
        targets = {k: np.concatenate(v) for k, v in targets.items() if v}
        if log_dir is not None:
            os.makedirs(log_dir, exist_ok=True)
            for pk, tk in pairs:
                if pk in preds and tk in targets:
                    np.save(os.path.join(log_dir, f"prediction_{pk}.npy"), preds[pk])
                    np.save(os.path.join(log_dir, f"target_{tk}.npy"), targets[tk])
        return preds, targets

    def save_checkpoint(self, path: str, **extra):
        """write the model in the checkpoint layout the reference reads back (`Model.load_from_checkpoint`, hamgnn/main.py:374-377,
        527-537): ``{"state_dict": {"representation.*", "output_module.*"}, ...}`` with the reference's parameter names and flat e3nn
        layouts (tensors only, on the CPU).  The non-learned e3nn / reference buffers the product does not carry (w3j constants,
        `output_mask`, `freqs`, ...) are rebuilt by the reference's constructors and need `strict=False` there -- they are exactly the
        keys `load_reference_state_dict` ignores on the way in.  (Never read back by a real Lightning here: none is installed; the
        round trip is tested with this repository's own loader only.)"""
        sd = {k: v.detach().cpu().clone() for k, v in self.state_dict().items() if k.startswith(("representation.", "output_module."))}
        # 'pytorch-lightning_version' / 'hyper_parameters': what Lightning's checkpoint migration and load_from_checkpoint look for
        ckpt = {"state_dict": sd, "hamgnn_amd": True, "pytorch-lightning_version": "2.0.0", "epoch": 0, "global_step": 0, "hyper_parameters": {}}
        ckpt.update(extra)
        torch.save(ckpt, path)
        return path

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path: str, map_location=None, strict: bool = True, **model_kwargs):
        """Lightning's classmethod, for the checkpoint layouts the reference writes: keys ``representation.*`` / ``output_module.*``."""
        model = cls(**model_kwargs)
        sd = read_checkpoint_state_dict(checkpoint_path)
        load_reference_state_dict(model.representation, sd, prefix="representation.")
        load_reference_state_dict(model.output_module, sd, prefix="output_module.")
        other = [k for k in sd if not k.startswith(("representation.", "output_module."))]
        if strict and other:
            raise KeyError(f"unexpected checkpoint keys outside representation./output_module.: {other[:4]}")
        if map_location is not None:
            model = model.to(map_location)
        return model

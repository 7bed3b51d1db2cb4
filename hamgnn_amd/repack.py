"""Device-side weight repack for training.

`compile()` packs a block's weights into MFMA fragment order on the HOST (hamgnn_amd/plan.py): fine once per checkpoint, 0.1 s per
message block when an optimiser steps every few hundred milliseconds.  Every element of a packed weight blob is either a structural
constant (aligned-frame coefficients, identity fragments, zero padding) or ONE source element times a constant (path normalisation,
1 / sqrt(fan)), where the sources are the block's flat parameters and the products  L' = linear_scaler.linear_out @ linear_out  of its
two trailing Linears:
        blob[p] = const[p] + coef[p] * source[idx[p]].
`AffinePack` DISCOVERS (const, coef, idx) by running the unmodified host builder on probe sources (zeros, ones, 1..n, all in float64),
checks the discovered map on a random source, and from then on repacks on the device: one gather + multiply-add per program, the
L' products as small `torch` matmuls.  The planner stays the single definition of the layouts; nothing about fragments is restated here.

Reference context: the reference has no packing step (e3nn evaluates its flat weights directly); this exists because the MI355X kernels
read MFMA-ordered fragments.  hamgnn/main.py:389-420 (`trainer.fit`) is the loop this makes affordable."""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Callable, Dict

import numpy as np
import torch

from . import plan as P


class AffinePack:
    def __init__(self, builder: Callable[[Dict[str, np.ndarray]], np.ndarray], sizes: "OrderedDict[str, int]", seed: int = 0):
        """builder(sources: name -> flat float64 array) -> flat blob; sizes: name -> number of elements, in source order"""
        self.sizes = OrderedDict(sizes)
        self.total = int(sum(self.sizes.values()))
        self._builder = builder
        b0 = self._run(np.zeros(self.total))
        b1 = self._run(np.ones(self.total))
        b2 = self._run(np.arange(1, self.total + 1, dtype=np.float64))
        coef = b1 - b0
        nz = coef != 0
        idx = np.zeros(b0.shape, dtype=np.int64)
        idx[nz] = np.rint((b2[nz] - b0[nz]) / coef[nz]).astype(np.int64) - 1
        if idx.min(initial=0) < 0 or idx.max(initial=0) >= self.total:
            raise ValueError("AffinePack: a blob element depends on more than one source element")
        self.const, self.coef, self.idx = b0, coef, idx
        rnd = np.random.default_rng(seed).normal(size=self.total)
        want, got = self._run(rnd), self.apply_flat(rnd)
        if np.abs(want - got).max(initial=0.0) > 1e-9 * max(1.0, np.abs(want).max(initial=0.0)):
            raise ValueError("AffinePack: the builder is not affine with single dependencies in these sources")
        self._dev = None

    def _split(self, flat):
        out, o = {}, 0
        for k, n in self.sizes.items():
            out[k] = flat[o:o + n]
            o += n
        return out

    def _run(self, flat):
        with P.probe_dtype():
            return np.asarray(self._builder(self._split(flat)), dtype=np.float64).reshape(-1)

    def apply_flat(self, flat: np.ndarray) -> np.ndarray:
        return self.const + self.coef * flat[self.idx]

    def apply_np(self, sources: Dict[str, np.ndarray]) -> np.ndarray:
        return self.apply_flat(np.concatenate([np.asarray(sources[k], dtype=np.float64).reshape(-1) for k in self.sizes]))

    def apply(self, sources: Dict[str, torch.Tensor]) -> torch.Tensor:
        """the float32 blob on the device of the sources (float64 arithmetic, rounded once -- as the host packer)"""
        first = next(iter(sources.values()))
        dev = first.device
        if self._dev is None or self._dev[0] != dev:
            self._dev = (dev, torch.from_numpy(self.const).to(dev), torch.from_numpy(self.coef).to(dev), torch.from_numpy(self.idx).to(dev))
        _, const, coef, idx = self._dev
        flat = torch.cat([sources[k].reshape(-1).double() for k in self.sizes])
        assert flat.numel() == self.total, (flat.numel(), self.total)
        return torch.addcmul(const, coef, flat[idx]).float()


# ------------------------------------------------------------------------------------------------ MessagePackBlock sources
def mp_branch_layouts(irreps_node, irreps_edge, irreps_sh, irreps_out):
    """per branch: (name, [(k, ls offset, fan, lo offset, mul_k)]) -- see plan.linear_scaler_layout"""
    return [("node", P.linear_scaler_layout(2, P.PlanarLayout(P.Irreps(irreps_node)), irreps_sh, irreps_out)),
            ("edge", P.linear_scaler_layout(1, P.PlanarLayout(P.Irreps(irreps_edge)), irreps_sh, irreps_out))]


def mp_source_sizes(sd_shapes: Dict[str, tuple], last_keys: Dict[str, str], with_skip: int = 0) -> "OrderedDict[str, int]":
    sizes = OrderedDict()
    for name in ("node", "edge"):
        sizes[f"{name}_tp"] = int(np.prod(sd_shapes[f"{name}_tensor_product.weight"]))
        sizes[f"{name}_w3"] = int(np.prod(sd_shapes[last_keys[name]]))
        sizes[f"{name}_lp"] = int(np.prod(sd_shapes[f"{name}_linear_scaler.linear_out.weight"]))
    if with_skip:
        sizes["skip"] = int(with_skip)
    return sizes


def mp_probe_state_dict(src: Dict[str, np.ndarray], sd_shapes, last_keys, layouts, irreps_out) -> Dict[str, np.ndarray]:
    """the state dict the host builders see while probing: L' enters through linear_scaler with the trailing o3.Linear set to the
    identity (times sqrt(mul_k), which the builder divides out again)"""
    irreps_out = P.Irreps(irreps_out)
    sd = {}
    for name, lay in layouts:
        sd[f"{name}_tensor_product.weight"] = src[f"{name}_tp"]
        sd[last_keys[name]] = src[f"{name}_w3"].reshape(sd_shapes[last_keys[name]])
        sd[f"{name}_linear_scaler.linear_out.weight"] = src[f"{name}_lp"]
        lo = np.zeros(sum(mk * mk for mk, _, _ in irreps_out))
        for (k, off, fan, lo_off, mk) in lay:
            lo[lo_off:lo_off + mk * mk] = (np.eye(mk) * math.sqrt(mk)).reshape(-1)
        sd[f"{name}_linear_out.weight"] = lo
    return sd


def lp_block_units(lay):
    """csrc/block_gemm.hip units of one branch's L' products (flat layouts of linear_scaler.linear_out / linear_out): unit k writes
    Ls_k [fan, mul_k] @ Lo_k [mul_k, mul_k] / sqrt(mul_k) behind the blocks before it"""
    units, o = [], 0
    for (k, off, fan, lo_off, mk) in lay:
        units.append((off, mk, 0, lo_off, mk, 0, o, mk, fan, mk, mk, 1.0 / math.sqrt(mk)))
        o += fan * mk
    return units, o


def mp_sources(get, last_keys, layouts, skip=None, lib=np, lp=None):
    """the real sources of a block from its parameters: `get(name)` -> flat array / tensor; lib = numpy or torch.
    L' block of output irrep k, in linear_scaler's flat layout:  Ls_k @ Lo_k / sqrt(mul_k)  (the builder applies 1 / sqrt(fan)).
    lp: {branch: the concatenated L' blocks} when the caller has them already (one hg_block_gemm launch per branch on the device)."""
    out = {}
    for name, lay in layouts:
        out[f"{name}_tp"] = get(f"{name}_tensor_product.weight")
        out[f"{name}_w3"] = get(last_keys[name])
        if lp is not None and name in lp:
            out[f"{name}_lp"] = lp[name]
            continue
        ls, lo = get(f"{name}_linear_scaler.linear_out.weight"), get(f"{name}_linear_out.weight")
        parts = []
        for (k, off, fan, lo_off, mk) in lay:
            parts.append(((ls[off:off + fan * mk].reshape(fan, mk) @ lo[lo_off:lo_off + mk * mk].reshape(mk, mk)) / math.sqrt(mk)).reshape(-1))
        out[f"{name}_lp"] = lib.concatenate(parts) if lib is np else torch.cat(parts)
    if skip is not None:
        out["skip"] = skip
    return out

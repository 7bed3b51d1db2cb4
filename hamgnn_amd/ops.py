"""Thin torch-tensor wrappers over the C ABI (include/hamgnn_hip.h).  PyTorch is plumbing here: device memory, streams.
Every function launches hand-written HIP kernels from libhamgnn_hip.so; nothing falls back to torch arithmetic."""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import plan as P
from ._lib import check, f32, i32, i64, lib, ptr


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_tensor_device(fn):
    """Run the launch with the device of the first tensor argument current (its current stream is then the launch stream): a caller
    that drives a second GPU without torch.cuda.set_device would otherwise enqueue on the wrong device's stream."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        for a in args:
            t = a if torch.is_tensor(a) else (a[0] if isinstance(a, (list, tuple)) and a and torch.is_tensor(a[0]) else None)
            if t is not None:
                if t.is_cuda and t.device.index != torch.cuda.current_device():
                    with torch.cuda.device(t.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapped


def _dev(arr: np.ndarray, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device)


def require_fp32(module, data=None):
    """The HIP path computes in fp32 (tables, kernels, outputs).  The reference's `precision: 64` flow -- torch.set_default_dtype(float64) + model.to(float64)
    + float64 graph tensors (hamgnn/main.py:469-474, hamgnn/models/hamgnn_conv.py:248-250) -- must not silently come back as fp32-accurate rows: refuse it."""
    why = None
    if torch.get_default_dtype() == torch.float64:
        why = "torch.get_default_dtype() is float64"
    elif module is not None and next((p.dtype for p in module.parameters()), None) == torch.float64:      # (model.to(float64) converts every parameter: the first one tells; O(1) per forward)
        why = "the model's parameters are float64"
    elif data is not None:
        pos = data["pos"] if isinstance(data, dict) else getattr(data, "pos", None)
        if torch.is_tensor(pos) and pos.dtype == torch.float64:
            why = "data.pos is float64"
    if why:
        raise NotImplementedError(f"precision: 64 is not built ({why}): the MI355X path computes in fp32 and does not down-cast silently "
                                  "(reference: hamgnn/main.py:469-474)")


def _require_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("hamgnn_amd: the MI355X hot path needs CUDA(ROCm) tensors; there is no CPU fallback")


S_SPLIT_OFF = os.environ.get("HG_S_SPLIT", "1") == "0"      # A/B switch: the fp32 form of the radial scale (16 fp32 MFMAs per row tile instead of 6 half-precision ones)
W3_SPLIT_PENDING: list = []     # (DeviceProgram, device scalar max |W3|) of refreshed programs whose half-precision range check is still to be read (check_w3_split)
REPLAY_SPLIT = False         # set by graph_capture.CapturedForward while it warms up and captures: launches of the smallest crystals take the finer 2d split
REPLAY_SPLIT_TILES = int(os.environ.get("HG_REPLAY_SPLIT_TILES", "1024"))    # the 2d split while tiles x segments x phase shares stay below this (two rounds of the chip's 512 slots)


_BUILD_CONFIG_OK = False


def check_build_config():
    """the planner's wave / ring counts against the compiled ones of the loaded library (once per process; raises on a mismatch instead of launching a
    schedule the kernel would misread -- the counts are environment-tunable for the A/B scripts under tools/)"""
    global _BUILD_CONFIG_OK
    if _BUILD_CONFIG_OK:
        return
    L = lib()
    want = {0: ("HG_IS_WAVES", P.IS_WAVES), 1: ("HG_LITE_WAVES", P.IS_WAVES_LITE), 2: ("HG_LITE_SRING", P.LITE_SRING)}
    for what, (name, val) in want.items():
        got = int(L.hg_build_config(what))
        if got != int(val):
            raise RuntimeError(f"{name}={val} does not match the loaded library (compiled with {got}): rebuild the variant or unset the variable")
    _BUILD_CONFIG_OK = True


def w3_split_refill(w: torch.Tensor, se, so, dh, dl, scale: float, lo_scale: float) -> torch.Tensor:
    """the split-half-precision twins of the fp32 W3 blocks of the packed weight blob `w`, rewritten IN PLACE by one launch (csrc/aux_kernels.hip:hg_w3_split_refill; index
    tensors: DeviceProgram.refresh_w3_split); returns the device scalar max |w 2^s| for the lazy range check (check_w3_split)"""
    _require_gpu(w)
    assert w.dtype == torch.float32 and w.is_contiguous() and all(t.dtype == torch.int64 and t.is_contiguous() for t in (se, so, dh, dl))
    mx = torch.empty(1, device=w.device, dtype=torch.float32)
    check(lib().hg_w3_split_refill(ptr(w), ptr(se), ptr(so), ptr(dh), ptr(dl), i64(se.numel()), C.c_float(scale), C.c_float(lo_scale), ptr(mx), _stream()), "hg_w3_split_refill")
    return mx[0]


def check_w3_split():
    """one host read for all programs refreshed since the last launch: a W3 weight beyond the half-precision range switches that program's launches to the fp32
    form of the radial scale (and back, once the weights have come back)"""
    if not W3_SPLIT_PENDING:
        return
    pend = list(W3_SPLIT_PENDING)
    W3_SPLIT_PENDING.clear()
    mx = torch.stack([m for _, m in pend]).cpu()
    for (dp, _), m in zip(pend, mx.tolist()):
        dp._w3_split_off = not (m <= P.W3_SPLIT_MAX)


class DeviceProgram:
    """A plan.Program uploaded to the GPU."""

    def __init__(self, prog: P.Program, device, schedule: str = "seg"):
        """schedule: "seg" = segment-stationary kernel only (hg_tp_fused); "is" / "auto" = also build the input-stationary
        schedule (hg_tp_is); "auto" silently keeps "seg" when it does not fit; "is_parts" = input-stationary with the output segments
        spread over as many workgroups per 16-edge tile as their LDS tiles need (plan.lds_partition)."""
        self.prog = prog
        self.weights = _dev(prog.weights, device)
        self.segs = _dev(prog.seg_table, device)
        self.items = _dev(prog.item_table, device)
        self.nseg = int(prog.seg_table.shape[0])
        self.hidden = int(prog.hidden_pad)
        self.out_dim = int(prog.out_layout.dim)
        self.lds_bytes = int(prog.tile_floats) * 4
        self.flags = 1 if (prog.item_table.shape[0] and (prog.item_table[:, 0] == P.IT_POST).any()) else 0
        # input-stationary schedule of the same items (csrc/tp_is.hip) when the tiles of all output segments fit the LDS
        self.sched = None
        self._device = device
        self._is_tables = {}                                   # parts -> (IsSchedule, device tables)
        self._is_weights = {}                                  # parts -> weight blob with the schedule's own streams appended (lite_mode runs)
        if prog.vsegs and schedule not in ("is", "is_parts"):
            raise ValueError("a program with merged items runs on the input-stationary kernel only")
        self.fixed_parts = None                                # "lds": the tiles of all output segments need several workgroups per 16 edges
        if schedule in ("is", "auto", "is_parts"):             # "is_parts": input-stationary, over several workgroups per tile if need be
            try:
                self.sched = self.is_tables(1)[0]
            except NotImplementedError:
                if schedule == "is":
                    raise
                if schedule == "is_parts":
                    self.sched = self.is_tables("lds")[0]
                    self.fixed_parts = "lds"

    def weights_changed(self):
        """after the packed weight blob was rewritten in place (nn.MessagePackBlock.refresh): rebuild what is derived from it"""
        self.refresh_w3_split()
        self._is_weights.clear()                               # (lite programs are recompiled, not refreshed; kept consistent anyway)
        self._is_tables = {k: v for k, v in self._is_tables.items() if v[0].extra_weights is None}

    def refresh_w3_split(self):
        """the split-half-precision twins of the W3 fragment blocks (plan/program.py:w3_split_fill) recomputed ON THE DEVICE from the fp32 blocks of the blob --
        the device-side repack (hamgnn_amd/repack.py) is affine in the parameters, hi = f16(x) / lo = f16(x - hi) is not.  Whether every weight is inside the
        half-precision range is checked lazily, by ONE host read before the next launch (check_w3_split): outside it the launches keep the fp32 form."""
        regs = getattr(self.prog, "w3_regions", None)
        if not regs:
            return
        if getattr(self, "_w3_idx", None) is None:
            se, so, dh, dl = [], [], [], []
            for off, rtm in regs:
                n = 4 * rtm * 256
                src, dst = P.w3_split_index(rtm)
                se.append(off + src[:, 0]); so.append(off + src[:, 1]); dh.append(off + n + dst[0]); dl.append(off + n + dst[1])
            self._w3_idx = tuple(_dev(np.concatenate(a).astype(np.int64), self._device) for a in (se, so, dh, dl))
        se, so, dh, dl = self._w3_idx
        sc_, lo_ = float(2.0 ** int(getattr(self.prog, "w3_exp", 0))), float(2.0 ** P.SPLIT_LO_EXP)
        W3_SPLIT_PENDING.append((self, w3_split_refill(self.weights, se, so, dh, dl, sc_, lo_)))

    def part_table_host(self, sc) -> np.ndarray:
        """the schedule's part table as the launch passes it in host memory: [12] (W3 split twins present) cleared when a refreshed weight left the half-precision range"""
        if (getattr(self, "_w3_split_off", False) or S_SPLIT_OFF) and int(sc.part_table[0][12]):
            pt = sc.part_table.copy()
            pt[:, 12] = 0
            return pt
        return sc.part_table

    def is_tables(self, parts):
        """(schedule, device tables) of the input-stationary kernel split into `parts` sub-schedules (built on first use)"""
        if parts not in self._is_tables:
            sc = P.is_schedule(self.prog, parts)
            self._is_tables[parts] = (sc, tuple(_dev(t, self._device) for t in (sc.seg_table, sc.block_table, sc.phase_table, sc.group_table,
                                                                                  sc.item_table, sc.part_table, sc.rowtab)))
            if sc.extra_weights is not None:                   # lite_mode runs: their step streams ride behind the program's weights
                self._is_weights[parts] = torch.cat([self.weights, _dev(sc.extra_weights, self._device, torch.float32)])
        return self._is_tables[parts]

    def is_weights(self, parts) -> torch.Tensor:
        """the weight blob a launch with `parts` sub-schedules reads"""
        return self._is_weights.get(parts, self.weights)

    def is_parts_for(self, rows: int) -> int:
        """How many workgroups share one 16-edge tile.  The chip holds 512 workgroups of this kernel (2 per CU); a launch with fewer
        tiles than that is a latency problem -- every workgroup walks the whole program serially -- so the output segments are spread
        over 8 (or one per segment) workgroups per tile while the extra staging work still fits the idle CUs."""
        if self.fixed_parts is not None:
            return self.fixed_parts
        forced = os.environ.get("HG_IS_PARTS")
        if forced:
            return max(1, int(forced))
        tiles = (rows + 15) // 16
        nseg = int(self.prog.seg_table.shape[0])
        # under hipGraph capture only (graph_capture.CapturedForward; HG_REPLAY_SPLIT = "2d<K>" forces it, "0" forbids it): one workgroup per (output segment,
        # K-th of its phases), tiles ADDED into zero-filled rows -- the one schedule without a fixed summation order, hence never the eager default
        mode = os.environ.get("HG_REPLAY_SPLIT", "2d4" if REPLAY_SPLIT else "0")
        if mode.startswith("2d") and tiles * nseg * int(mode[2:] or 3) <= REPLAY_SPLIT_TILES and not int(self.sched.part_table[0][11]):
            return ("2d", nseg, int(mode[2:] or 3))
        if tiles * nseg <= 512:
            return nseg
        if tiles <= 300 and nseg >= 8:
            return 8
        return 1


class DeviceLinear:
    """plan.LinearTables uploaded to the GPU (hg_linear_planar)"""

    def __init__(self, tabs: P.LinearTables, device):
        self.tabs = tabs
        self.nitems = int(tabs.items.shape[0])
        self.items, self.units, self.paths = _dev(tabs.items, device), _dev(tabs.units, device), _dev(tabs.paths, device)
        self.weights = _dev(tabs.weights, device)
        self.in_dim, self.out_dim = int(tabs.in_dim), int(tabs.out_dim)


@_on_tensor_device
def linear_planar(dl: DeviceLinear, x: torch.Tensor, res: Sequence[Optional[torch.Tensor]] = (), tag: str = "linear") -> torch.Tensor:
    """o3.Linear on planar rows, one streaming pass (csrc/linear.hip); res: up to two residual row tensors (output layout) added on the way"""
    _require_gpu(x)
    rows = int(x.shape[0])
    assert x.shape[1] == dl.in_dim and x.stride(1) == 1
    res = [r for r in res if r is not None]
    assert len(res) <= 2
    for r in res:
        assert r.shape == (rows, dl.out_dim) and r.stride(1) == 1
    out = torch.empty(rows, dl.out_dim, device=x.device, dtype=torch.float32)         # every column is written, padding included
    if PROFILE_EVENTS is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().hg_linear_planar(ptr(x), i64(x.stride(0)), ptr(dl.items), i32(dl.nitems), ptr(dl.units), ptr(dl.paths), ptr(dl.weights),
                                 C.c_void_p(res[0].data_ptr() if res else 0), i64(res[0].stride(0) if res else 0),
                                 C.c_void_p(res[1].data_ptr() if len(res) > 1 else 0), i64(res[1].stride(0) if len(res) > 1 else 0),
                                 i64(rows), ptr(out), i64(dl.out_dim), _stream()), "hg_linear_planar")
    if PROFILE_EVENTS is not None:
        ev1.record()
        PROFILE_EVENTS.append((ev0, ev1, rows, tag))
    return out


LW_ROWS = 1024                     # rows of a chunk of csrc/linear_wgrad.hip
_LW_TABLES: dict = {}


def linear_wgrad_tables(irreps_in, irreps_out, device):
    """(units int32 [nunits, 8] on the device, gather index int64 [numel] into the summed partial blocks, scale float32 [numel]) of the weight
    gradient of o3.Linear(irreps_in -> irreps_out): a unit = 16 input channels x <= 64 output channels of one path; the flat e3nn weight
    (paths (i_in, i_out), each [mul_in, mul_out] row-major, 1 / sqrt(fan_in)) is a gather of the unit blocks.  Cached per signature."""
    key = (str(irreps_in), str(irreps_out), str(device))
    if key not in _LW_TABLES:
        from .so3 import Irreps
        ii, io = Irreps(irreps_in), Irreps(irreps_out)
        li, lo = P.PlanarLayout(ii), P.PlanarLayout(io)
        paths = [(i, k) for i, (_, l1, p1) in enumerate(ii) for k, (_, l2, p2) in enumerate(io) if (l1, p1) == (l2, p2)]
        fan = {}
        for i, k in paths:
            fan[k] = fan.get(k, 0) + ii[i][0]
        units, gather, scale = [], [], []
        for i, k in paths:
            mi, l, _ = ii[i]
            mk = io[k][0]
            ublock = {}
            for u0 in range(0, mi, 16):
                for v0 in range(0, mk, 64):
                    ublock[(u0, v0)] = len(units)
                    units.append([li.off[i], li.mulp[i], lo.off[k], lo.mulp[k], 2 * l + 1, u0, v0, min(64, mk - v0)])
            for u in range(mi):
                for v in range(mk):
                    gather.append(ublock[(u - u % 16, v - v % 64)] * 1024 + (u % 16) * 64 + v % 64)
                    scale.append(1.0 / math.sqrt(fan[k]))
        _LW_TABLES[key] = (_dev(np.asarray(units if units else [[0] * 8], np.int32), device), len(units),
                           torch.tensor(gather, dtype=torch.int64, device=device), torch.tensor(scale, dtype=torch.float32, device=device))
    return _LW_TABLES[key]


def linear_wgrad(irreps_in, irreps_out, x: torch.Tensor, gy: torch.Tensor) -> torch.Tensor:
    """d sum(y * gy) / d weight of an o3.Linear in e3nn's flat layout, all paths in ONE launch of csrc/linear_wgrad.hip + a fixed-order sum over
    the row chunks (x, gy: planar rows of the Linear's input and of the gradient of its output)"""
    _require_gpu(x)
    units, nunits, gather, scale = linear_wgrad_tables(irreps_in, irreps_out, x.device)
    if nunits == 0:
        return x.new_zeros(0)
    rows = int(x.shape[0])
    assert gy.shape[0] == rows and x.stride(1) == 1 and gy.stride(1) == 1
    assert x.dtype == torch.float32 and gy.dtype == torch.float32, "hg_linear_wgrad reads fp32 rows (raw pointers: any other dtype would be misread)"
    nchunk = (rows + LW_ROWS - 1) // LW_ROWS
    part = torch.empty(nchunk, nunits * 1024, device=x.device, dtype=torch.float32)
    check(lib().hg_linear_wgrad(ptr(x), i64(x.stride(0)), ptr(gy), i64(gy.stride(0)), i64(rows), ptr(units), i32(nunits), ptr(part), _stream()), "hg_linear_wgrad")
    return part.sum(0)[gather] * scale


class BlockGemm:
    """unit table of csrc/block_gemm.hip on the device: units = [(a_off, a_ld, a_trans, b_off, b_ld, b_trans, c_off, c_ld, M, N, K, scale)]"""

    def __init__(self, units, device):
        arr = np.zeros((max(1, len(units)), 12), np.int32)
        for n, u in enumerate(units):
            arr[n, :11] = [int(v) for v in u[:11]]
            arr[n, 11] = np.float32(u[11]).view(np.int32)
        self.nunits = len(units)
        self.max_tiles = max([-(-int(u[8]) // 64) * -(-int(u[9]) // 64) for u in units], default=1)
        self.units_np, self.units = arr, _dev(arr, device)


def use_block_gemm(t: torch.Tensor) -> bool:
    """hg_block_gemm is a HIP entry point: device tensors only (the CPU suite swaps block_gemm for its numpy twin and takes it too)"""
    return (t.is_cuda or block_gemm.__module__ != __name__) and os.environ.get("HG_BLOCK_GEMM", "1") != "0"


def block_gemm(bg: BlockGemm, a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """c[c_off + m * c_ld + n] = scale * sum_k op(a)[m, k] op(b)[k, n] for every unit of `bg` (a, b: flat fp32; c: flat fp32 or fp64, written in place)"""
    _require_gpu(a)
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and c.dtype in (torch.float32, torch.float64) and a.is_contiguous() and b.is_contiguous() and c.is_contiguous()
    check(lib().hg_block_gemm(ptr(a), ptr(b), ptr(c), i32(1 if c.dtype == torch.float64 else 0), ptr(bg.units), i32(bg.nunits), i32(bg.max_tiles), _stream()), "hg_block_gemm")
    return c


def wig_offsets(lmax):
    offs, tot = P.wigner_offsets(lmax)
    arr = (C.c_int * 8)(*([int(o) for o in offs] + [0] * (8 - len(offs))))
    return arr, tot


class Geometry:
    """rbf, packed per-edge Wigner matrices and the receiver CSR of one graph batch (computed once per forward)."""

    def __init__(self, pos, edge_index, nbr_shift, cutoff, num_radial, lmax, jtab_dev, rbf_func: str = "bessel"):
        _require_gpu(pos)
        if pos.device.index != torch.cuda.current_device():
            raise RuntimeError(f"hamgnn_amd: make {pos.device} the current device (torch.cuda.set_device) before the forward")
        E = edge_index.shape[1]
        dev = pos.device
        self.E, self.lmax = E, lmax
        self.wig_off, self.nW = wig_offsets(lmax)
        self.rbf = torch.empty(E, num_radial, device=dev, dtype=torch.float32)
        self.wig = torch.empty(E, self.nW, device=dev, dtype=torch.float32)
        self.length = torch.empty(E, device=dev, dtype=torch.float32)
        ang = torch.empty(E, 4, device=dev, dtype=torch.float32)
        pos = pos.contiguous().float()
        self.edge_index = edge_index.contiguous()
        nbr_shift = nbr_shift.contiguous().float()
        check(lib().hg_edge_geometry(ptr(pos), ptr(self.edge_index), ptr(nbr_shift), i64(E), f32(cutoff), i32(num_radial), i32(lmax),
                                     ptr(jtab_dev), ptr(self.rbf), ptr(self.wig), ptr(self.length), ptr(ang), _stream()), "hg_edge_geometry")
        if rbf_func == "gaussian":                             # GaussianSmearing(0, cutoff, num_radial) x cosine cutoff, from the lengths
            offs = _gaussian_offsets(float(cutoff), int(num_radial), dev)
            delta = float((offs[2][1] - offs[2][0]).item())    # the reference's width: spacing of its fp32 linspace (host copy, no sync)
            check(lib().hg_radial_basis(ptr(self.length), i64(E), i32(1), f32(cutoff), ptr(offs[1]), f32(delta), i32(num_radial), ptr(self.rbf),
                                        _stream()), "hg_radial_basis")
        elif rbf_func != "bessel":
            raise ValueError(f"Unsupported radial basis function on the MI355X path: {rbf_func}")
        self.src = self.edge_index[0].contiguous()
        self.dst = self.edge_index[1].contiguous()


_GAUSS_OFFS = {}


def _gaussian_offsets(cutoff, num_radial, dev):
    key = (cutoff, num_radial, str(dev))
    if key not in _GAUSS_OFFS:
        host = torch.linspace(0.0, cutoff, num_radial, dtype=torch.float32)
        _GAUSS_OFFS[key] = (key, host.to(dev), host)
    return _GAUSS_OFFS[key]


@_on_tensor_device
def radial_hidden(rbf: torch.Tensor, layers: Sequence[torch.Tensor], act_cst: float) -> torch.Tensor:
    """hidden activations of a FullyConnectedNet (every hidden layer: x @ W / sqrt(fan_in) folded into W, normalised SiLU); the kernel holds up to three
    layers per launch, deeper radial MLPs (`radial_MLP` with more than three entries; r6) chain launches -- every layer of the list is a hidden one"""
    if len(layers) > 3:
        return radial_hidden(radial_hidden(rbf, layers[:3], act_cst), layers[3:], act_cst)
    E = rbf.shape[0]
    dims = [int(layers[0].shape[0])] + [int(w.shape[1]) for w in layers]
    W = torch.cat([w.reshape(-1) for w in layers]).contiguous()
    out = torch.empty(E, dims[-1], device=rbf.device, dtype=torch.float32)
    darr = (C.c_int * len(dims))(*dims)
    check(lib().hg_radial_hidden(ptr(rbf), i64(E), ptr(W), darr, i32(len(layers)), f32(act_cst), ptr(out), _stream()), "hg_radial_hidden")
    return out


def radial_hidden_cached(geo: "Geometry", layers: Sequence[torch.Tensor], act_cst: float) -> torch.Tensor:
    """radial_hidden(geo.rbf, layers) through the per-forward cache on the geometry object (filled for all weight generators of a
    backbone at once by prefill_radial_hidden; computed singly otherwise)"""
    cache = geo.__dict__.setdefault("_hcache", {})
    key = id(layers[0])
    if key not in cache:
        cache[key] = radial_hidden(geo.rbf, layers, act_cst)
    return cache[key]


def prefill_radial_hidden(geo: "Geometry", generators: Sequence[Sequence[torch.Tensor]], act_cst: float) -> bool:
    """all weight generators of a forward in ONE launch (hg_radial_hidden_multi); False if their shapes do not allow it"""
    H = radial_hidden_multi(geo.rbf, generators, act_cst)
    if H is None:
        return False
    cache = geo.__dict__.setdefault("_hcache", {})
    for m, g in enumerate(generators):
        cache[id(g[0])] = H[m]
    return True


def mfma_probe(device, iters: int = 4000, reps: int = 3) -> float:
    """fp32 MFMA TFLOP/s the device sustains right now on random operands (hg_mfma_probe: two waves per SIMD on every CU, nothing but
    v_mfma_f32_16x16x4_f32 in the loop) -- the attainable ceiling bench.py quotes beside the nominal peak."""
    dev = torch.device(device)
    with torch.cuda.device(dev):
        src = torch.randn(65536, device=dev)
        nblocks = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
        out = torch.empty(nblocks * 256, device=dev)
        run = lambda: check(lib().hg_mfma_probe(ptr(src), ptr(out), i32(nblocks), i32(iters), _stream()), "hg_mfma_probe")
        run()
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            run()
        ev1.record()
        torch.cuda.synchronize(dev)
        ms = ev0.elapsed_time(ev1) / reps
    return nblocks * 4 * iters * 8 * 2048 / (ms * 1e-3) / 1e12


@_on_tensor_device
def radial_hidden_multi(rbf: torch.Tensor, generators: Sequence[Sequence[torch.Tensor]], act_cst: float) -> Optional[torch.Tensor]:
    """hidden activations of SEVERAL radial weight generators that read the same basis rows, one launch: [n, E, 64]; None when a
    generator is not of the shipped 64 -> 64 -> 64 shape (callers then use radial_hidden per generator)."""
    if not generators or any(len(g) != 2 or tuple(g[0].shape) != (64, 64) or tuple(g[1].shape) != (64, 64) for g in generators):
        return None
    if rbf.shape[1] != 64:
        return None
    E = rbf.shape[0]
    W = torch.cat([w.reshape(-1) for g in generators for w in g]).contiguous()
    out = torch.empty(len(generators), E, 64, device=rbf.device, dtype=torch.float32)
    check(lib().hg_radial_hidden_multi(ptr(rbf), i64(E), ptr(W), i32(len(generators)), f32(act_cst), ptr(out), _stream()), "hg_radial_hidden_multi")
    return out


@_on_tensor_device
def rotate_gather(x: torch.Tensor, idx: Optional[torch.Tensor], geo: Geometry, chan_tab: torch.Tensor, transpose=False,
                  x2: Optional[torch.Tensor] = None, idx2: Optional[torch.Tensor] = None):
    """one or two gathered sources rotated with the same per-edge frames; returns out (or (out, out2))."""
    E = geo.E
    Dp = int(x.shape[1])
    out = torch.empty(E, Dp, device=x.device, dtype=torch.float32)              # the kernel writes every slot incl. zero padding
    out2 = torch.empty(E, Dp, device=x.device, dtype=torch.float32) if x2 is not None else None
    if x2 is not None:
        assert x2.stride(0) == x.stride(0)
    check(lib().hg_rotate_gather(ptr(x), ptr(x2), i64(x.stride(0)), ptr(idx), ptr(idx2), ptr(geo.wig), i32(geo.nW), geo.wig_off,
                                 ptr(chan_tab), i32(chan_tab.shape[0]), i64(E), i32(1 if transpose else 0), ptr(out), ptr(out2), i64(Dp),
                                 _stream()), "hg_rotate_gather")
    return out if x2 is None else (out, out2)


PROFILE_EVENTS = None        # bench.py sets this to a list: (start, end, rows, tag) HIP event pairs around hg_tp_fused launches


@_on_tensor_device
def tp_fused(dp: DeviceProgram, srcs: List[torch.Tensor], rows: int, h2n=None, h2e=None, geo: Optional[Geometry] = None,
             tag: str = "linear", gather: Optional[List[Optional[torch.Tensor]]] = None, rot_mask: int = 0,
             res: Sequence[Optional[torch.Tensor]] = (), reduce=None) -> torch.Tensor:
    """gather / rot_mask (input-stationary schedule only): srcs[i] holds global-frame node rows, gathered by gather[i] and rotated
    into the edge frame inside the kernel (bit i of rot_mask) instead of a separate hg_rotate_gather pass.
    res: up to two residual row tensors in the output's planar layout, added in the epilogue (segment-stationary programs).
    reduce = (eperm, run_id, R) of topo.Topology.receiver_major(): the launch walks the rows in the order eperm and writes the R run sums
    instead of one row per edge (the fused node scatter; input-stationary single-part launches only)."""
    _require_gpu(srcs[0])
    assert (gather is None and rot_mask == 0) or dp.sched is not None
    res = [r for r in res if r is not None]
    assert len(res) <= 2 and (not res or dp.sched is None)
    for r in res:
        assert r.shape == (rows, dp.out_dim) and r.stride(1) == 1
    assert reduce is None or (dp.sched is not None and dp.is_parts_for(rows) == 1)
    parts_ = dp.is_parts_for(rows) if (dp.sched is not None and reduce is None) else 1
    alloc = torch.zeros if isinstance(parts_, tuple) else torch.empty  # (the 2d split ADDS its tiles into zero-filled rows: plan.is_schedule ("2d", P, K))
    out = alloc(rows if reduce is None else reduce[2], dp.out_dim, device=srcs[0].device, dtype=torch.float32)      # the kernel writes every slot incl. zero channel padding
    n = len(srcs)
    sp = (C.c_void_p * 4)(*([s.data_ptr() for s in srcs] + [0] * (4 - n)))
    ss = (C.c_int64 * 4)(*([int(s.stride(0)) for s in srcs] + [0] * (4 - n)))
    wig, nW, woff = (ptr(geo.wig), geo.nW, geo.wig_off) if geo is not None else (C.c_void_p(0), 0, (C.c_int * 8)())
    if PROFILE_EVENTS is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()                                   # torch's current stream == the launch stream (see _stream())
    if dp.sched is not None:
        check_build_config()
        check_w3_split()
    if dp.sched is not None:
        sc, (t_segs, t_blocks, t_phases, t_groups, t_items, t_parts, t_rowtab) = dp.is_tables(dp.is_parts_for(rows))
        gl = list(gather) + [None] * (4 - len(gather)) if gather is not None else [None] * 4
        gp = (C.c_void_p * 4)(*[(t.data_ptr() if t is not None else 0) for t in gl])
        check(lib().hg_tp_is(sp, ss, i32(n), ptr(h2n), ptr(h2e), i32(dp.hidden), wig, i32(nW), woff, ptr(dp.is_weights(dp.is_parts_for(rows))), ptr(t_segs),
                             ptr(t_blocks), ptr(t_phases), ptr(t_groups), ptr(t_items), ptr(t_parts),
                             np.ascontiguousarray(dp.part_table_host(sc)).ctypes.data_as(C.c_void_p), i32(sc.part_table.shape[0]), ptr(t_rowtab),
                             i32(sc.lds_floats * 4), gp, i32(rot_mask), ptr(reduce[0]) if reduce is not None else C.c_void_p(0),
                             ptr(reduce[1]) if reduce is not None else C.c_void_p(0), ptr(out), i64(dp.out_dim), i64(rows), _stream()), "hg_tp_is")
    else:
        check(lib().hg_tp_fused(sp, ss, i32(n), ptr(h2n), ptr(h2e), i32(dp.hidden), wig, i32(nW), woff, ptr(dp.weights), ptr(dp.segs),
                                i32(dp.nseg), ptr(dp.items), ptr(out), i64(dp.out_dim), i64(rows), i32(dp.lds_bytes), i32(dp.flags),
                                C.c_void_p(res[0].data_ptr() if res else 0), i64(res[0].stride(0) if res else 0),
                                C.c_void_p(res[1].data_ptr() if len(res) > 1 else 0), i64(res[1].stride(0) if len(res) > 1 else 0), _stream()),
              "hg_tp_fused")
    if PROFILE_EVENTS is not None:
        ev1.record()
        PROFILE_EVENTS.append((ev0, ev1, rows, tag))
    return out


import collections

_SCATTER_CACHE: "collections.OrderedDict" = collections.OrderedDict()     # LRU over (sort order, counts) of index tensors, bounded in entries AND bytes
_SCATTER_CACHE_MAX_ENTRIES = 64
_SCATTER_CACHE_MAX_BYTES = 256 << 20


def scatter_rows(index: torch.Tensor, src: torch.Tensor, n: int, persistent: bool = True) -> torch.Tensor:
    """out[i] = sum of the rows src[q] with index[q] == i, in a FIXED order (stable sort by index, then a segmented sum): the
    deterministic form of torch.zeros(n, ...).index_add_(0, index, src), whose float atomics make a training step differ from run to run.
    torch tensor ops (sort / segment_reduce), any device; edge-level glue of the backward passes, not a hot kernel.
    persistent = False: `index` is a throw-away view (not a constant of the model or the graph): its sort order is not cached.  The cache holds strong
    references to what it keeps, so it is a small LRU with a byte cap -- the edge-level indices of graphs already discarded do not pin device memory
    (13 MB per 822 k-edge index, 257 entries before)."""
    if src.shape[0] == 0:
        return src.new_zeros((n,) + tuple(src.shape[1:]))
    key = (index.data_ptr(), index._version, int(index.shape[0]), int(n), index.device)
    hit = _SCATTER_CACHE.get(key) if persistent else None
    if hit is None or hit[0] is not index:                     # the index tensors of the backward glue (tp_idx, l_idx, z, the contraction tables) are
        order = torch.sort(index.long(), stable=True).indices  # constants of a model: sorted once (keyed on the tensor object and its version)
        counts = torch.bincount(index.long(), minlength=n)
        if counts.shape[0] != n:                               # an index >= n: index_add_ raised here, segment_reduce(unsafe=True) would return extra rows
            raise IndexError(f"scatter_rows: index {int(index.max())} out of range for {n} rows")
        hit = (index, order, counts)
        if persistent:
            _SCATTER_CACHE[key] = hit
            size = lambda h: sum(t.numel() * t.element_size() for t in h)
            while len(_SCATTER_CACHE) > _SCATTER_CACHE_MAX_ENTRIES or (len(_SCATTER_CACHE) > 1 and sum(size(h) for h in _SCATTER_CACHE.values()) > _SCATTER_CACHE_MAX_BYTES):
                _SCATTER_CACHE.popitem(last=False)
    elif persistent:
        _SCATTER_CACHE.move_to_end(key)
    _, order, counts = hit
    return torch.segment_reduce(src[order].contiguous(), "sum", lengths=counts, axis=0, unsafe=True)


def scatter_cols(index: torch.Tensor, src: torch.Tensor, n: int) -> torch.Tensor:
    """out[:, i] = sum of src[:, q] with index[q] == i (src [rows, Q, ...] -> [rows, n, ...]) in a fixed order: scatter_rows along dim 1,
    the deterministic form of torch.zeros(rows, n, ...).index_add_(1, index, src)."""
    return scatter_rows(index, src.transpose(0, 1).contiguous(), n).transpose(0, 1)


class DeviceRowProgram:
    """plan.RowProgram uploaded to the GPU (csrc/rowprog.hip)"""

    def __init__(self, rp: "P.RowProgram", device):
        self.rp = rp
        self.stages = _dev(rp.stages, device)
        self.units = _dev(rp.units if rp.units.size else np.zeros((1, P.RP_UNIT_I32), np.int32), device)
        self.weights = _dev(rp.weights, device, torch.float32)
        self.act_tab = _dev(rp.act_tab if rp.act_tab.size else np.zeros((1, 2), np.int32), device)
        self.out_tab = _dev(rp.out_tab if rp.out_tab.size else np.zeros((1, 2), np.int32), device)
        self.consts = (C.c_float * 5)(*[float(c) for c in P.ACT_CONSTS])


@_on_tensor_device
def row_program(drp: DeviceRowProgram, x: torch.Tensor, res: Sequence[torch.Tensor] = (), row_idx: Optional[torch.Tensor] = None, tag: str = "row_program") -> torch.Tensor:
    """run a plan.RowProgram on planar rows x [rows, din] (rows gathered by row_idx when given) -> [rows, dout] (+ the rows in `res`)"""
    _require_gpu(x)
    rp = drp.rp
    x = x if x.stride(1) == 1 else x.contiguous()
    rows = int(row_idx.shape[0]) if row_idx is not None else int(x.shape[0])
    assert x.shape[1] >= rp.din and len(res) <= 2
    y = torch.empty(rows, rp.dout, device=x.device, dtype=torch.float32)
    r = [t if t.stride(1) == 1 else t.contiguous() for t in res]
    rp_ = lambda i: C.c_void_p(r[i].data_ptr() if i < len(r) else 0)
    rs_ = lambda i: i64(r[i].stride(0) if i < len(r) else 0)
    if PROFILE_EVENTS is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().hg_row_program(ptr(x), i64(x.stride(0)), C.c_void_p(row_idx.data_ptr() if row_idx is not None else 0), i32(rp.din), ptr(y), i64(y.stride(0)),
                               i32(rp.dout), rp_(0), rs_(0), rp_(1), rs_(1), ptr(drp.stages), i32(int(rp.stages.shape[0])), ptr(drp.units), ptr(drp.weights),
                               ptr(drp.act_tab), ptr(drp.out_tab), i32(int(rp.act_tab.shape[0])), i32(int(rp.out_tab.shape[0])), drp.consts, i32(rp.in_buf), i32(rp.out_buf), i32(rp.rs[0]), i32(rp.rs[1]), i32(rp.strip),
                               i64(rows), _stream()), "hg_row_program")
    if PROFILE_EVENTS is not None:
        ev1.record()
        PROFILE_EVENTS.append((ev0, ev1, rows, tag))
    return y


class DeviceWgFused:
    """plan.WgFused uploaded to the GPU (fused weight-gradient kernel, csrc/tp_wgrad.hip)"""

    def __init__(self, wf: "P.WgFused", device):
        self.wf = wf
        self.units = _dev(wf.units, device)
        self.weights = _dev(wf.weights, device, torch.float32)
        self.chtab = _dev(wf.chtab, device)
        self.tp_pos = [None if t is None else torch.from_numpy(t).to(device) for t in wf.tp_pos]
        self.tp_scale = [None if t is None else torch.from_numpy(np.asarray(t, dtype=np.float32)).to(device) for t in wf.tp_scale]
        self.l_pos = [torch.from_numpy(t).to(device) for t in wf.l_pos]

    def nsplit_for(self, rows: int) -> int:
        """edge splits of a launch: enough workgroups to fill the chip several times over, at least a few iterations each"""
        tiles = (rows + 15) // 16
        return int(max(1, min(64, -(-8192 // int(self.units.shape[0])), tiles // 8)))      # measured (profiles/r03_wgrad.md): 8 -> 64 splits 11.5 -> 9.3 ms at 44 k edges


def tp_wgrad(dwf: DeviceWgFused, srcs: Sequence[Optional[torch.Tensor]], g: torch.Tensor, h_node: torch.Tensor, h_edge: Optional[torch.Tensor],
             nsplit: Optional[int] = None):
    """one launch of the fused weight-gradient kernel over all rows of `g`: (acc [nsplit, acc_floats], [gs per branch [rows, n_channels]]).
    srcs: edge-frame planar source rows by slot (None for slots the block does not use)."""
    _require_gpu(g)
    wf = dwf.wf
    rows = int(g.shape[0])
    S = int(nsplit or dwf.nsplit_for(rows))
    acc = torch.zeros(S, wf.acc_floats, device=g.device, dtype=torch.float32)
    alloc = torch.empty if wf.gs_complete else torch.zeros      # (every channel of every row is written by exactly one wave when the row tiles cover all channels)
    gs = [alloc(rows, n, device=g.device, dtype=torch.float32) for n in wf.nch]
    n = len(srcs)
    keep = [(t if t.stride(1) == 1 else t.contiguous()) if t is not None else None for t in srcs]      # (column windows of wider rows pass as they are: pointer + row stride)
    for t in keep:
        assert t is None or (t.dtype == torch.float32 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0), "hg_tp_wgrad stages float4 pieces of fp32 rows"
    sp = (C.c_void_p * 4)(*([(t.data_ptr() if t is not None else 0) for t in keep] + [0] * (4 - n)))
    ss = (C.c_int64 * 4)(*([(t.stride(0) if t is not None else 0) for t in keep] + [0] * (4 - n)))
    g = g.contiguous()
    h_node = h_node.contiguous()
    h_edge = h_edge.contiguous() if h_edge is not None else None
    assert h_node.shape[1] >= wf.hidden and (h_edge is None or h_edge.stride(0) == h_node.stride(0))
    with torch.cuda.device(g.device):
        check(lib().hg_tp_wgrad(sp, ss, i32(n), ptr(g), i64(g.stride(0)), ptr(h_node), C.c_void_p(h_edge.data_ptr() if h_edge is not None else 0),
                                i64(h_node.stride(0)), i32(wf.hidden), ptr(gs[0]), i64(gs[0].stride(0)),
                                C.c_void_p(gs[1].data_ptr() if len(gs) > 1 else 0), i64(gs[1].stride(0) if len(gs) > 1 else 0),
                                ptr(acc), i64(wf.acc_floats), i32(S), ptr(dwf.units), i32(int(dwf.units.shape[0])), ptr(dwf.weights), ptr(dwf.chtab),
                                i32(wf.lds_bytes), i64(rows), _stream()), "hg_tp_wgrad")
    return acc, gs


@_on_tensor_device
def segment_sum(msg: torch.Tensor, rowptr: torch.Tensor, perm: torch.Tensor, N: int) -> torch.Tensor:
    Dp = msg.shape[1]
    out = torch.empty(N, Dp, device=msg.device, dtype=torch.float32)
    check(lib().hg_segment_sum(ptr(msg), i64(msg.stride(0)), ptr(rowptr), ptr(perm), i64(N), i32(Dp), ptr(out), i64(Dp), _stream()), "hg_segment_sum")
    return out


@_on_tensor_device
def attention_aggregate(K: torch.Tensor, V: torch.Tensor, geo: "Geometry", rowptr, perm, head_tab: torch.Tensor, H: int, head_dim: int,
                        cut_param: torch.Tensor, cutoff: float) -> torch.Tensor:
    """AttentionAggregation of the reference (hamgnn/nn/attention.py:126-164) with key = K[sender], query = K[receiver] and the soft
    cutoff weight: [N, Dp] planar node rows from [E, Dp] planar value rows."""
    N, Dp, E = int(K.shape[0]), int(K.shape[1]), geo.E
    assert V.shape == (E, Dp) and V.stride(1) == 1 and K.stride(1) == 1
    logits = torch.empty(E, H, device=K.device, dtype=torch.float32)
    check(lib().hg_attn_logits(ptr(K), i64(K.stride(0)), ptr(geo.src), ptr(geo.dst), ptr(geo.length), ptr(head_tab), i32(Dp), i32(H),
                               ptr(cut_param), f32(cutoff), f32(1.0 / math.sqrt(head_dim)), i64(E), ptr(logits), _stream()), "hg_attn_logits")
    out = torch.empty(N, Dp, device=K.device, dtype=torch.float32)
    check(lib().hg_attn_aggregate(ptr(logits), i32(H), ptr(V), i64(V.stride(0)), ptr(rowptr), ptr(perm), ptr(head_tab), i64(N), i32(Dp),
                                  ptr(out), i64(Dp), _stream()), "hg_attn_aggregate")
    return out


@_on_tensor_device
def attention_logits(K: torch.Tensor, geo: "Geometry", head_tab: torch.Tensor, H: int, head_dim: int, cut_param: torch.Tensor, cutoff: float) -> torch.Tensor:
    """[E, H] soft-cutoff-weighted per-head dot products of the two gathered key rows (the first half of attention_aggregate; used on its
    own by the edge-sharded attention, which needs the per-node soft-max statistics of the rank's edges)"""
    Dp, E = int(K.shape[1]), geo.E
    logits = torch.empty(E, H, device=K.device, dtype=torch.float32)
    check(lib().hg_attn_logits(ptr(K), i64(K.stride(0)), ptr(geo.src), ptr(geo.dst), ptr(geo.length), ptr(head_tab), i32(Dp), i32(H),
                               ptr(cut_param), f32(cutoff), f32(1.0 / math.sqrt(head_dim)), i64(E), ptr(logits), _stream()), "hg_attn_logits")
    return logits


@_on_tensor_device
def gate(x: torch.Tensor, tabs, consts: torch.Tensor) -> torch.Tensor:
    """tabs = (act_tab [nact,2], out_tab [Dout,2]) device int32 tensors of plan.gate_tables_compact"""
    act_tab, out_tab = tabs
    rows, Dout = x.shape[0], int(out_tab.shape[0])
    out = torch.empty(rows, Dout, device=x.device, dtype=torch.float32)
    check(lib().hg_gate(ptr(x), i64(x.stride(0)), ptr(act_tab), i32(act_tab.shape[0]), ptr(out_tab), i32(Dout), ptr(consts), i64(rows), ptr(out),
                        i64(Dout), _stream()), "hg_gate")
    return out


NORM_ACT_EPS = 1e-8          # epsilon of the reference's NormActivation (interaction_blocks.py:329)


def norm_act(x: torch.Tensor, chan_tab: torch.Tensor) -> torch.Tensor:
    """e3nn NormActivation (ssp, normalize, eps 1e-8) on planar rows; chan_tab = plan.norm_act_table(irreps) on the device"""
    _require_gpu(x)
    assert x.dtype == torch.float32 and x.stride(1) == 1
    rows, D = x.shape
    out = torch.empty(rows, D, device=x.device, dtype=torch.float32)
    check(lib().hg_norm_act(ptr(x), i64(x.stride(0)), ptr(chan_tab), i32(chan_tab.shape[0]), f32(NORM_ACT_EPS), i64(rows), ptr(out), i64(D), i32(D), _stream()),
          "hg_norm_act")
    return out


def norm_act_backward(x: torch.Tensor, gy: torch.Tensor, chan_tab: torch.Tensor) -> torch.Tensor:
    _require_gpu(x)
    assert x.dtype == torch.float32 and gy.dtype == torch.float32 and x.stride(1) == 1 and gy.stride(1) == 1 and gy.shape == x.shape
    rows, D = x.shape
    gx = torch.empty(rows, D, device=x.device, dtype=torch.float32)
    check(lib().hg_norm_act_backward(ptr(x), i64(x.stride(0)), ptr(gy), i64(gy.stride(0)), ptr(chan_tab), i32(chan_tab.shape[0]), f32(NORM_ACT_EPS), i64(rows),
                                     ptr(gx), i64(D), i32(D), _stream()), "hg_norm_act_backward")
    return gx


@_on_tensor_device
def add_rows(a, b, c=None):
    rows, D = a.shape
    out = torch.empty_like(a)
    check(lib().hg_add_rows(ptr(a), i64(a.stride(0)), ptr(b), i64(b.stride(0)), ptr(c), i64(c.stride(0) if c is not None else 0), i64(rows),
                            i32(D), ptr(out), i64(D), _stream()), "hg_add_rows")
    return out


@_on_tensor_device
def to_planar(x: torch.Tensor, imap: torch.Tensor, Dp: int) -> torch.Tensor:
    x = x.contiguous().float()
    out = torch.empty(x.shape[0], Dp, device=x.device, dtype=torch.float32)
    check(lib().hg_to_planar(ptr(x), i64(x.shape[0]), i32(x.shape[1]), ptr(imap), ptr(out), i32(Dp), _stream()), "hg_to_planar")
    return out


@_on_tensor_device
def from_planar(xp: torch.Tensor, imap: torch.Tensor) -> torch.Tensor:
    D = int(imap.shape[0])
    out = torch.empty(xp.shape[0], D, device=xp.device, dtype=torch.float32)
    check(lib().hg_from_planar(ptr(xp), i64(xp.shape[0]), i32(xp.shape[1]), ptr(imap), ptr(out), i32(D), _stream()), "hg_from_planar")
    return out


@_on_tensor_device
def embed_lookup(Ta, Tb, z, idx_a, idx_b, rows, T, Tp):
    out = torch.empty(rows, Tp, device=Ta.device, dtype=torch.float32)
    check(lib().hg_embed_lookup(ptr(Ta), ptr(Tb), ptr(z), ptr(idx_a), ptr(idx_b), i64(rows), i32(T), i32(Tp), ptr(out), _stream()), "hg_embed_lookup")
    return out


@_on_tensor_device
def gate_backward(x: torch.Tensor, gy: torch.Tensor, tabs, consts: torch.Tensor) -> torch.Tensor:
    """data gradient of gate(x, tabs, consts): [rows, Din] from the gate input rows and the gradient of the gate output rows"""
    act_tab, out_tab = tabs
    rows, Din, Dout = x.shape[0], int(x.shape[1]), int(out_tab.shape[0])
    assert gy.shape == (rows, Dout) and x.stride(1) == 1 and gy.stride(1) == 1
    gx = torch.empty(rows, Din, device=x.device, dtype=torch.float32)
    check(lib().hg_gate_backward(ptr(x), i64(x.stride(0)), ptr(gy), i64(gy.stride(0)), ptr(act_tab), i32(act_tab.shape[0]), ptr(out_tab), i32(Dout),
                                 ptr(consts), i64(rows), i32(Din), ptr(gx), i64(Din), _stream()), "hg_gate_backward")
    return gx


@_on_tensor_device
def ham_merge(coeff, geo: Optional[Geometry], slot_tab, cg_ptr, cg_idx, cg_val, nout):
    rows = coeff.shape[0]
    out = torch.empty(rows, nout, device=coeff.device, dtype=torch.float32)
    wig, nW, woff = (ptr(geo.wig), geo.nW, geo.wig_off) if geo is not None else (C.c_void_p(0), 0, (C.c_int * 8)())
    check(lib().hg_ham_merge(ptr(coeff), i64(coeff.stride(0)), wig, i32(nW), woff, ptr(slot_tab), i32(slot_tab.shape[0]), ptr(cg_ptr),
                             ptr(cg_idx), ptr(cg_val), i32(nout), i64(rows), ptr(out), _stream()), "hg_ham_merge")
    return out


@_on_tensor_device
def ham_finish(Hraw, inv, H0, orb_mask, z, idx_a, idx_b, nao, sign=1.0, symmetrize=True, h0_after_mask=False, out=None):
    """Hraw: [rows, >= nao^2] (a column slice of a wider buffer is fine: the row stride is passed on).
    out: optional contiguous [rows, nao^2] destination (e.g. the row range of the [N+E, nao^2] result: no torch.cat afterwards)."""
    rows = Hraw.shape[0]
    assert Hraw.stride(1) == 1
    if out is None:
        out = torch.empty(rows, nao * nao, device=Hraw.device, dtype=torch.float32)
    assert out.shape == (rows, nao * nao) and out.is_contiguous() and out.dtype == torch.float32
    mask_w = int(orb_mask.shape[1]) if orb_mask is not None else 0
    _require_gpu(Hraw)
    check(lib().hg_ham_finish(C.c_void_p(Hraw.data_ptr()), i64(Hraw.stride(0)), ptr(inv), ptr(H0), ptr(orb_mask), i32(mask_w), ptr(z), ptr(idx_a), ptr(idx_b),
                              i32(nao), f32(sign), i32((1 if symmetrize else 0) | (2 if h0_after_mask else 0)), i64(rows), ptr(out),
                              _stream()), "hg_ham_finish")
    return out


@_on_tensor_device
def ham_readout(coeff, geo: Optional[Geometry], slot_tab, cg_ptr, cg_idx, cg_val, nao, pairs, H0, orb_mask, z, idx_a, idx_b, out,
                lmax_ham, sign=1.0, symmetrize=True, h0_after_mask=False):
    """one-pass non-SOC read-out of `coeff` rows into `out` [rows, nao^2]; pairs = (pair_a, pair_b) int64 row indices or None (rows pair
    with themselves: on-site)."""
    _require_gpu(coeff)
    rows = coeff.shape[0]
    assert out.shape == (rows, nao * nao) and out.is_contiguous() and out.dtype == torch.float32
    wig, nW, woff = (ptr(geo.wig), geo.nW, geo.wig_off) if geo is not None else (C.c_void_p(0), 0, (C.c_int * 8)())
    pa, pb = pairs if pairs is not None else (None, None)
    npairs = rows if pa is None else pa.shape[0]
    mask_w = int(orb_mask.shape[1]) if orb_mask is not None else 0
    check(lib().hg_ham_readout(ptr(coeff), i64(coeff.stride(0)), i32(coeff.shape[1]), wig, i32(nW), woff, i32(lmax_ham), ptr(slot_tab), i32(slot_tab.shape[0]), ptr(cg_ptr),
                               ptr(cg_idx), ptr(cg_val), i32(cg_idx.shape[0]), i32(nao), ptr(pa), ptr(pb), i64(npairs), ptr(H0), ptr(orb_mask),
                               i32(mask_w), ptr(z), ptr(idx_a), ptr(idx_b), f32(sign), i32((1 if symmetrize else 0) | (2 if h0_after_mask else 0)),
                               C.c_void_p(out.data_ptr()), _stream()), "hg_ham_readout")
    return out


@_on_tensor_device
def block_mean(x, tab, nao):
    rows = x.shape[0]
    out = torch.empty(rows, nao * nao, device=x.device, dtype=torch.float32)
    check(lib().hg_block_mean(ptr(x), i64(x.stride(0)), ptr(tab), i32(nao), i64(rows), ptr(out), _stream()), "hg_block_mean")
    return out


@_on_tensor_device
def soc_assemble(H, ksi, L, inv, H0r, H0i, nao, symmetrize=True, zero_diag=False):
    rows = H.shape[0]
    outr = torch.empty(rows, 4 * nao * nao, device=H.device, dtype=torch.float32)
    outi = torch.empty_like(outr)
    check(lib().hg_soc_assemble(ptr(H), ptr(ksi), ptr(L), ptr(inv), ptr(H0r), ptr(H0i), i32(nao), i32(1 if symmetrize else 0),
                                i32(1 if zero_diag else 0), i64(rows), ptr(outr), ptr(outi), _stream()), "hg_soc_assemble")
    return outr, outi


@_on_tensor_device
def hk_assemble(on, off, nbr_shift, kvec, pair_ptr, pair_edges, pair_ij, n_atoms, nao, orank, ooff, M):
    """[nk, M, M] complex64 k-space matrix of one crystal in the compact orbital basis (see include/hamgnn_hip.h:hg_hk_assemble)"""
    _require_gpu(on)
    nk = int(kvec.shape[0])
    out = torch.zeros(nk, M, M, 2, device=on.device, dtype=torch.float32)
    check(lib().hg_hk_assemble(ptr(on), ptr(off), ptr(nbr_shift), ptr(kvec), i32(nk), ptr(pair_ptr), ptr(pair_edges), ptr(pair_ij),
                               i64(pair_ij.shape[0]), i32(n_atoms), i32(nao), ptr(orank.to(torch.int32).contiguous()), ptr(ooff), i32(M), ptr(out),
                               _stream()), "hg_hk_assemble")
    return torch.view_as_complex(out)


@_on_tensor_device
def zero_point_shift(H, Href, S, nao, soc=False, threshold=1e-6):
    """in place on H; returns the shift (device scalar)."""
    _require_gpu(H)
    rows = S.shape[0]
    nparts = 256
    scratch = torch.empty(2 * nparts, device=H.device, dtype=torch.float64)
    shift = torch.empty(1, device=H.device, dtype=torch.float32)
    check(lib().hg_zero_point_shift(ptr(H), ptr(Href), ptr(S), i64(rows), i32(nao), i32(1 if soc else 0), f32(threshold), ptr(scratch),
                                    i32(nparts), ptr(shift), _stream()), "hg_zero_point_shift")
    return shift


@_on_tensor_device
def sym_contraction3(h, z, C, tab, W3, out):
    """adds the nu = 3 term of a `correlation: 3` block to the rows `out` that sym_contraction returned (in place; csrc/corr3.hip)"""
    _require_gpu(h)
    check(lib().hg_sym_contraction3(ptr(h), i64(h.stride(0)), ptr(z), i64(h.shape[0]), i32(C), i32(tab["num_ell"]), ptr(tab["ell_off"]), i32(tab["nout"]),
                                    ptr(tab["out_off"]), ptr(tab["ptr3"]), ptr(tab["ent3"]), ptr(W3), i32(W3.shape[1]), ptr(out), i64(out.stride(0)),
                                    _stream()), "hg_sym_contraction3")
    return out


@_on_tensor_device
def sym_contraction(h, z, C, tab, W1, W2, out_dim):
    """tab: device tensors of plan.sym_contraction_tables; W1 [nel, K1, C], W2 [nel, K2, C]; returns planar hidden rows [N, out_dim]"""
    _require_gpu(h)
    N = h.shape[0]
    out = torch.zeros(N, out_dim, device=h.device, dtype=torch.float32)           # channel padding stays zero
    check(lib().hg_sym_contraction(ptr(h), i64(h.stride(0)), ptr(z), i64(N), i32(C), i32(tab["num_ell"]), ptr(tab["ell_off"]), i32(tab["nout"]),
                                   ptr(tab["out_off"]), ptr(tab["ptr1"]), ptr(tab["ent1"]), ptr(tab["ptr2"]), ptr(tab["ent2"]), ptr(W1),
                                   i32(W1.shape[1]), ptr(W2), i32(W2.shape[1]), ptr(out), i64(out_dim), _stream()), "hg_sym_contraction")
    return out

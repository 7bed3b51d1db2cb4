"""Backward of a lite_mode MessagePackBlock (reference: hamgnn/nn/message_passing.py:99-125, 197-215).

Forward, per edge and output irrep k (edge-aligned frame; csrc: IT_LINC items + the IT_POST combine):
    t_k[w, m]   = sum_{paths p = (i, l_sh, k)} cf_p[m] sum_u Wp[u, w] x_i[u, src_p(m)]          uvu products folded with the _MidLinears
    out_k[w', m] = sum_w Lc_k[w, w'] s[ch_k + w] t_k[w, m]                                       s = radial MLP (one weight per channel of out)
Everything the gradients need is [E, planar(irreps_out)]-sized -- unlike the weighted blocks, nothing large has to be materialised:
    g_u = Lc g_out  (the o3.Linear's adjoint tables),   g_t = s * g_u,   g_s = sum_m t * g_u,   g_Lc = (s t)^T g_out,
    g_x  through the adjoint IT_LINC program (plan.build_message_pack_lite_adjoint_program) on the same fused kernels,
    g_Wp[u, w] = sum_{e, m} cf[m] x_i[u, src(m)] g_t[w, m]   (one small GEMM per path on planar blocks),
    the radial MLP by autograd on its dense layers.
Device-agnostic torch ops around two kernel launches (callbacks): the CPU suite runs the same code on the emulator (tests/emu.py) and
checks every gradient against autograd through the oracle."""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

from . import ops, plan as P
from .backward_mp import _block, radial_mlp


class LiteBackward:
    def __init__(self, sd: Dict[str, np.ndarray], irreps_node, irreps_edge, irreps_sh, irreps_out):
        self.irreps_node, self.irreps_edge = P.Irreps(irreps_node), P.Irreps(irreps_edge)
        self.irreps_sh, self.irreps_out = P.Irreps(irreps_sh), P.Irreps(irreps_out)
        self.sd = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items()}
        args = (self.irreps_node, self.irreps_edge, self.irreps_sh, self.irreps_out)
        self.prog_t = P.build_message_pack_program_lite(sd, *args, unrotate=False, post=False)          # the pre-combine rows t, edge frame
        self.prog_adj = P.build_message_pack_lite_adjoint_program(sd, *args)
        self.lc_adj = P.build_linear_adjoint_tables(self.sd["combine_messages.linear_out.weight"], self.irreps_out, self.irreps_out)
        self.paths = {"node": list(P.lite_paths(P.PlanarLayout(self.irreps_node), 2, self.irreps_sh, self.irreps_out, self.sd["node_linear_scaler.weight"])),
                      "edge": list(P.lite_paths(P.PlanarLayout(self.irreps_edge), 1, self.irreps_sh, self.irreps_out, self.sd["edge_linear_scaler.weight"]))}
        gl = P.PlanarLayout(self.irreps_out)
        chan = np.full(gl.dim, -1, dtype=np.int64)             # planar column of an output row -> radial channel (irreps_out.simplify() order)
        co = 0
        for k, (mk, lk, pk) in enumerate(self.irreps_out):
            for a in range(2 * lk + 1):
                chan[gl.off[k] + a * gl.mulp[k]:gl.off[k] + a * gl.mulp[k] + mk] = co + np.arange(mk)
            co += mk
        self.chan, self.nch = chan, co
        self.gen_keys = sorted(k for k in self.sd if k.startswith("weight_generator_combine.layer") and k.endswith(".weight"))

    def run(self, run_program, run_linear, linear_weight_grad, xs, xd, f, g_out, rbf, act_cst: float, params=None):
        """xs / xd / f: planar edge-frame input rows; g_out: gradient of the block's output rows (edge frame, planar(irreps_out)).
        run_program(prog, sources) -> rows; run_linear(tables, rows) -> rows; linear_weight_grad(irreps_in, irreps_out, x, gy) -> flat.
        params: optional {name: device tensor} of the CURRENT parameters (else the packed copy).
        Returns (adjoint rows [E, planar(message_pack_adjoint_layout)], {parameter name: gradient})."""
        dev, dt = g_out.device, g_out.dtype
        par = lambda k: (params[k].detach().to(dt) if params is not None else torch.as_tensor(self.sd[k], device=dev, dtype=dt))
        gen = [par(k).clone().requires_grad_() for k in self.gen_keys]
        H = gen[-1].shape[0]
        with torch.no_grad():
            h = radial_mlp(rbf, [w.detach() for w in gen[:-1]], act_cst)
            s = h @ (gen[-1].detach() / math.sqrt(H))                                            # [E, nch]
            chan = torch.as_tensor(self.chan, device=dev)
            valid = torch.nonzero(chan >= 0).reshape(-1)
            s_cols = torch.zeros_like(g_out)
            s_cols[:, valid] = s[:, chan[valid]]
            t = run_program(self.prog_t, [xs, xd, f])
            g_u = run_linear(self.lc_adj, g_out)
            g_t = (s_cols * g_u).contiguous()
            g_s = ops.scatter_cols(chan[valid], (t * g_u)[:, valid], self.nch)       # fixed summation order (index_add_ sums with float atomics)
            grads = {"combine_messages.linear_out.weight": linear_weight_grad(self.irreps_out, self.irreps_out, s_cols * t, g_out),
                     self.gen_keys[-1]: h.t() @ g_s / math.sqrt(H)}
            g_h = g_s @ (gen[-1].detach().t() / math.sqrt(H))
            rows_adj = run_program(self.prog_adj, [g_t])
            gl = P.PlanarLayout(self.irreps_out)
            for name, srcs, lay in (("node", (xs, xd), P.PlanarLayout(self.irreps_node)), ("edge", (f,), P.PlanarLayout(self.irreps_edge))):
                gw = torch.zeros(self.sd[f"{name}_linear_scaler.weight"].size, device=dev, dtype=dt)
                for pth in self.paths[name]:
                    i, k, mm, li, lk, mi, mk = pth["i"], pth["k"], pth["mm"], pth["li"], pth["lk"], pth["mi"], pth["mk"]
                    nc = 2 * mm + 1
                    comps = [(li + mm - c) if pth["par"] else (li - mm + c) for c in range(nc)]
                    X = torch.cat([_block(x_, lay.off[i], lay.mulp[i], comps, mi) for x_ in srcs], 2)              # [E, nc, nsrc mi]
                    G = _block(g_t, gl.off[k], gl.mulp[k], [lk - mm + c for c in range(nc)], mk)                   # [E, nc, mk]
                    cf = torch.as_tensor(pth["cf"], device=dev, dtype=dt)
                    blk = torch.einsum("ecu,ecw->uw", X * cf[None, :, None], G) * pth["scale"]
                    gw[pth["w_off"]:pth["w_off"] + blk.numel()] = blk.reshape(-1)
                grads[f"{name}_linear_scaler.weight"] = gw
        if len(gen) > 1:                                        # hidden layers of the radial MLP: dense 64-wide layers, torch.autograd
            with torch.enable_grad():
                hfull = radial_mlp(rbf, gen[:-1], act_cst)
                for k, w, gk in zip(self.gen_keys[:-1], gen[:-1], torch.autograd.grad(hfull, gen[:-1], grad_outputs=g_h, allow_unused=True)):
                    grads[k] = gk if gk is not None else torch.zeros_like(w)
        return rows_adj, grads

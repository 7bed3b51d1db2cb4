"""Plan builder for the fused edge kernel (host side, numpy; PRODUCT code -- never imports oracle/).

Turns the reference's modules/parameters (flat e3nn weight layouts, reference names) into the data the HIP kernel
``hg_tp_fused`` (csrc/tp_fused.hip) executes:

  * a *planar* feature layout: per irrep (mul, l, p) a block [2l+1][mulp] with mulp = ceil4(mul); channel index fastest,
  * SEGMENTS (one per output irrep) and ITEMS (one per (input irrep, output irrep) super-path row chunk), and
  * one flat fp32 weight buffer holding, for every item, operands already in MFMA 16x16x4-f32 *fragment order*:
        A1  [nsrc][ksteps][rtm][64]   uvw weights * path coefficient                (GEMM1: rows = stacked (l_sh, w) channels)
        W3  [hsteps][rtm][64]         last radial-MLP layer columns of those rows   (per-edge scale via MFMA)
        CF  [rtm][nc][4][4]           aligned-frame CG coefficient per (row, m)
        A2  [rto][rtm][4][64]         Linear(mid->out) folded with the trailing o3.Linear(out->out)   (GEMM2)

Math (per edge, edge-aligned frame, see hamgnn_amd/so3.py): for a path p=(i, l_sh, k) of the reference's uvw tensor
product (hamgnn/nn/message_passing.py:136-171) followed by LinearScaleWithWeights (tensor_products.py:25-47) and the
out linear (message_passing.py:133-134, 229):
    out'_k[w'', m] += sum_w L'_k[(p,w), w''] * s_e[(p,w)] * coef_p[m] * sum_u (c_p W_p[u,w]) x'_i[u, src_p(m)]
MFMA lane conventions (v_mfma_f32_16x16x4_f32):  A[i = lane&15][k = lane>>4],  B[k = lane>>4][j = lane&15],
C/D: col = lane&15, row = 4*(lane>>4) + reg.  Edges are the MFMA *columns*; channels are rows; results chain
GEMM1 -> scale -> GEMM2 without any cross-lane movement (C regs feed the next B operand with a permuted K order).

Modules (r6: the 3 000-line plan.py of rounds 1-5 split by what a table is FOR; every name is re-exported here, `from hamgnn_amd import plan as P` keeps working):
  layout        planar rows, record constants, the reference's instruction table
  program       the Program container, MFMA-fragment packing, item builders (tensor product, its adjoint, plain Linear)
  schedule_is   the input-stationary schedule of a Program (phases / work groups / parts; csrc/tp_is.hip)
  message_pack  state_dict -> Program for the MessagePackBlock (forward, data gradient, lite_mode), the pair embedding, o3.Linear programs
  wgrad         the fused weight-gradient kernel's tables (csrc/tp_wgrad.hip) + the materialisation route
  tables        streaming Linears, gates, row programs, attention heads, the read-out's CG merge maps, symmetric contraction
"""
from . import layout, message_pack, program, schedule_is, tables, wgrad

for _m in (layout, program, schedule_is, message_pack, wgrad, tables):
    globals().update({_k: _v for _k, _v in vars(_m).items() if not _k.startswith("__") and not isinstance(_v, type(layout))})
del _m

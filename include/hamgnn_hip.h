/* hamgnn_hip.h -- C ABI of libhamgnn_hip.so: the MI355X (gfx950) drop-in for HamGNN's equivariant message-passing
 * hot path.  The reference (QuantumLab-ZY/HamGNN) has NO native/FFI interface on this path: its boundary is the Python
 * module API (hamgnn/models/hamgnn_conv.py:88,248 ; hamgnn/models/hamgnn_output.py:96,2916).  Each entry point below
 * therefore names the reference *PyTorch-op cluster* it replaces (file:line relative to the reference root).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch's allocator in the Python host layer) EXCEPT the tiny
 *    host arrays `src`, `src_stride`, `wig_off`, `dims` (<= 8 entries, copied into the kernel arguments); no hidden
 *    allocation, no host synchronisation; `stream` is a hipStream_t passed as void*.
 *  - features use the PLANAR layout of hamgnn_amd/plan.py: per irrep (mul,l,p) a block [2l+1][mulp], mulp = ceil4(mul).
 *  - return value: 0 = ok, negative = error (hg_last_error() gives the text).  Plain C types only.
 */
#ifndef HAMGNN_HIP_H
#define HAMGNN_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* hg_last_error(void);
int hg_version(void);

/* Workspace query.  No entry point of this library allocates or synchronises: every output and every scratch buffer is provided by
 * the caller (PyTorch's allocator in the shipped host code).  Scratch that is not an output: hg_edge_geometry (ang_scratch, rows = E)
 * and hg_zero_point_shift (partial_scratch, arg = nparts); all other entry points report 0.  Host-only, needs no GPU.
 * (SURVEY 8b sketched a plan-object ABI: plan create / workspace-bytes / forward-with-plan calls.  The built boundary keeps the
 * planner on the host side of the ABI instead: the C functions take the planner's tables as plain device arrays, so the library
 * holds no state, and this query stands in for the plan's workspace-bytes call.)                                                */
int64_t hg_scratch_bytes(const char* entry_point, int64_t rows, int arg);

/* (a1+a2) SphericalHarmonicEdgeAttrs.forward  hamgnn/toolbox/nequip/nn/embedding/_edge.py:59-67
 *         RadialBasisEdgeEncoding.forward      hamgnn/nn/embeddings.py:73-100  (+ utils/basis_functions.py:193-208,
 *         utils/cutoff_functions.py:50-61).
 * Per edge e (j = edge_index[0][e] centre, i = edge_index[1][e] neighbour): v = pos[i] + nbr_shift[e] - pos[j];
 * rbf[e][n] = sin((n+1) pi r / rc)/r * 0.5 (cos(pi r/rc)+1) [r<rc];  wig[e] = packed Wigner matrices D^l(R_e), l=0..lmax_wig,
 * of the rotation that takes the edge direction onto the pole (edge-aligned frame; REPLACES the explicit SH tensor: in that
 * frame Y^l = sqrt(2l+1) e_0, so no `sh` output exists); edge_len[e] = |v|.
 * jtab: DEVICE copy of the constants from hamgnn_amd/so3.py:wigner_tables, packed [sum (2l+1)^2] J matrices then
 * [sum (l+1)] sine signs.  ang_scratch: caller-provided [E][4] floats (rotation angles; no hidden allocation).        */
int hg_edge_geometry(const float* pos, const int64_t* edge_index, const float* nbr_shift, int64_t E, float cutoff,
                     int num_radial, int lmax_wig, const float* jtab, float* rbf, float* wig, float* edge_len,
                     float* ang_scratch, void* stream);

/* rbf_func = "gaussian": GaussianSmearing(0, cutoff, num_radial)  hamgnn/utils/basis_functions.py:211-224 (hamgnn_conv.py:123-125)
 * x CosineCutoff (nn/embeddings.py:93-97), from the edge lengths hg_edge_geometry wrote:
 * rbf[e][n] = exp(-0.5 (r - offsets[n])^2 / delta^2) * 0.5 (cos(pi r / rc) + 1) [r < rc].  kind: 1 = gaussian (the only one; Bessel
 * comes from hg_edge_geometry; the reference's exp-gaussian / exp-bernstein / bernstein bases return float64 and need `precision: 64`).
 * offsets: DEVICE [num_radial] fp32 centres = torch.linspace(0, rc, num_radial); delta = offsets[1] - offsets[0] in fp32.          */
int hg_radial_basis(const float* edge_len, int64_t E, int kind, float cutoff, const float* offsets, float delta, int num_radial,
                    float* rbf, void* stream);

/* e3nn FullyConnectedNet hidden layers (all but the last) of a radial weight generator:
 * hamgnn/nn/message_passing.py:173-189, 218, 223 ; tensor_products.py:152-168, 183.
 * h = act(... act(rbf @ W0) @ W1 ...), act(x) = act_cst * silu(x); W_k [dims[k], dims[k+1]] already scaled by 1/sqrt(h_in). */
int hg_radial_hidden(const float* rbf, int64_t E, const float* weights, const int32_t* dims, int nlayers, float act_cst,
                     float* h_out, void* stream);
/* the same for `nmlp` weight generators of the shipped shape 64 -> 64 -> 64 that read the same radial basis rows, in ONE launch
 * (grid.y = generator): weights [nmlp][2][64][64] (layer 1 | layer 2, 1/sqrt(fan_in) folded in), h_out [nmlp][E][64].            */
int hg_radial_hidden_multi(const float* rbf, int64_t E, const float* weights, int nmlp, float act_cst, float* h_out, void* stream);

/* Measurement aid for bench.py's roofline block (no reference counterpart): nblocks workgroups of 4 waves issue iters x 8
 * v_mfma_f32_16x16x4_f32 each on independent accumulators, operands from in65536 (65 536 floats, caller-filled, e.g. random);
 * out: nblocks * 256 floats (checksum sink).  FLOPs issued = nblocks * 4 * iters * 8 * 2048. */
int hg_mfma_probe(const float* in65536, float* out, int nblocks, int iters, void* stream);

/* node_features[sender] / [receiver] gathers of ConvBlockE3.forward (hamgnn/nn/convolution.py:138-141) and
 * PairInteractionBlock.forward (interaction_blocks.py:141-145), fused with the rotation into the edge-aligned frame:
 * out_s[e][i][a][u] = sum_b D_e^{l_i}[a][b] x_s[idx_s[e]][i][b][u]  for up to two sources s = 0,1 sharing the edge's D
 * (x1 == NULL: one source; idx == NULL: identity gather; transpose != 0: D^T, i.e. back to the global frame).
 * grp_tab: int32[ngroups][4] = {l, planar offset of (component 0, first channel), mulp, valid channels 1..4}, one entry per
 * group of 4 channel slots, sorted by l (plan.py:rotate_table); padding channel slots are written as zeros; row strides
 * must be multiples of 4 floats.                                                                                         */
int hg_rotate_gather(const float* x0, const float* x1, int64_t x_stride, const int64_t* idx0, const int64_t* idx1,
                     const float* wig, int nW, const int32_t* wig_off, const int32_t* grp_tab, int ngroups, int64_t E,
                     int transpose, float* out0, float* out1, int64_t out_stride, void* stream);

/* THE hot kernel.  Replaces, per launch, one whole MessagePackBlock.forward (hamgnn/nn/message_passing.py:191-231:
 * AttentionHeadsToVector + 2 x o3.TensorProduct(uvw, ~255 paths each) + 2 x LinearScaleWithWeights (tensor_products.py:25-47)
 * + last radial-MLP layer + 2 x o3.Linear) [+ PairInteractionBlock skip linear, interaction_blocks.py:151-152]; the
 * embedding TP (tensor_products.py:170-189); or any o3.Linear (IT_LIN items).  Programs come from hamgnn_amd/plan.py.
 * src[k]/src_stride[k]: planar source rows (slot 0: rotated src-node rows, 1: rotated dst-node rows, 2: edge rows).
 * rows: number of edges (or nodes for node-level linears).  lds_bytes: prog.tile_floats*4 (dynamic LDS).
 * program_flags: bit 0 = the program contains lite_mode segment post-ops (plan.IT_POST; selects that kernel instantiation).
 * res0 / res1 (nullable; res1 needs res0): residual rows in the OUTPUT's planar layout, added in the epilogue -- the
 * `x + Lin2(Gate(Lin1 x))` and `+ skip` adds of ResidualBlock / ConvBlockE3 (hamgnn/nn/interaction_blocks.py:352-357,
 * convolution.py:158) ride on the producing launch; not combined with un-rotating segments.                            */
int hg_tp_fused(const float* const* src, const int64_t* src_stride, int nsrc, const float* h2_node, const float* h2_edge,
                int hidden, const float* wig, int nW, const int32_t* wig_off, const float* weights,
                const int32_t* seg_table, int nseg, const int32_t* item_table, float* out, int64_t out_stride,
                int64_t rows, int lds_bytes, int program_flags, const float* res0, int64_t res0_stride, const float* res1,
                int64_t res1_stride, void* stream);

/* Input-stationary schedule of the same fused MessagePackBlock (same reference ops as hg_tp_fused: message_passing.py:191-231
 * [+ interaction_blocks.py:151-152]); csrc/tp_is.hip.  One workgroup = 16 edges with the tiles of ALL output segments in LDS;
 * outer loop over phases (plan.py:is_schedule): the input irrep blocks of a phase are staged once per 16 edges and shared by
 * the four waves, which claim the phase's work groups dynamically (a shared tile is updated by one wave per phase: fixed order).
 *   seg_table   int32[nseg][8]   = {lk, mul_k, rto, out_off, out_mulp, tile_off, Wigner stage_off, flags (| 1<<16: new batch)}
 *   block_table int32[nblock][8] = {s0, s1, in_off, in_mulp, li, nsrc, stage_off0, stage_off1}
 *   phase_table int32[nphase][8] = {block_begin, block_end, group_begin, group_end, radial generator (0 / 1) whose hidden rows the kernel
 *               keeps in registers for the phase's items or -1, 0, 0, 0};  group_table int32[ngroup][2] = item range
 *   item_table  int32[nitems][24]: as for hg_tp_fused with [1], [2] = stage offsets of source 0 / 1 (-1), [19] = segment,
 *               [20..22] = {lk, mul_k, rto} of that segment, [23] = first row-table entry of the rows its GEMM2 writes
 *   row_table   int32: per part, for every output row (segment, 16-row tile, row) of GEMM2 the LDS float offset (relative to a tile
 *               copy) of that row's centre column; rows beyond the segment's multiplicity point at the trash row
 *   part_table  int32[nparts][16] = {first segment, segments, first phase, phases, trash_off, stage_off, ctr_off, copy_stride,
 *               rowtab_off (LDS float offset of the part's copy of the row table), rowtab_begin, rowtab_len, lite flag, 0, 0, 0, 0}: the launch runs
 *               nparts sub-schedules (grid.y) that own disjoint sets of output segments; one part = the whole program, several
 *               parts spread a 16-edge tile's serial pass over several workgroups when there are fewer tiles than CUs (small
 *               crystals: BASELINE configs #1 and #5).  trash_off / stage_off / ctr_off: float offsets of the padding-row sink,
 *               the staging area and the claim counter inside the workgroup's LDS (lds_bytes = the largest part's need);
 *               [7] = copy_stride > 0: every wave accumulates into a private copy of the part's tiles; such a part's work groups are DEALT,
 *               not claimed (r6): group_begin + k * waves + w is the k-th work group of wave w (empty groups end a short stream), the
 *               copies are folded in a fixed order -- one summation order per launch, bit-identical results between runs.  part_table_host: the
 *               same table in HOST memory (one of the tiny host arrays; validated, and a single part's scalars travel as
 *               kernel arguments).  r5: parts may SHARE a segment range and take different phase ranges of it; the shared
 *               segments carry flag bit 1 (SEG_ATOMIC) in seg_table[.][7], their epilogues ADD into `out`, which the caller has
 *               zero-filled.  The order of THOSE adds is not fixed (sums differ between runs at fp32 rounding level): the host uses this
 *               form under hipGraph capture only (graph_capture.CapturedForward), never for eager forwards.
 * src_idx[i] (nullable): row gather of source i; rot_mask bit i: source i holds GLOBAL-frame
 * node rows that are gathered and rotated by D^l(R_e) while staged -- the node_features[sender/receiver] gathers of
 * convolution.py:138-141 / interaction_blocks.py:141-145 fused into the operand staging (no hg_rotate_gather pass, no per-edge
 * copies of the node rows).
 * edge_perm (nullable): DEVICE int64[rows], tile slot -> edge whose rows the slot reads (receiver-major launch order); run_id (nullable,
 * single-part launches only): DEVICE int32[rows], slot -> OUTPUT row that takes the sum of the slot's run of equal receivers (runs never
 * cross a 16-slot tile; -1: padding slot) -- the receiver scatter of convolution.py:147-149 as a segmented reduce in the epilogue; `out`
 * then holds one row per run (hamgnn_amd/topo.py:Topology.receiver_major), summed per receiver by hg_segment_sum.
 * lite_mode programs: items of type 6 (IT_STREAM, plan._lite_streams) carry [8] = steps, [11] / [12] = float offsets of the fragment and
 * descriptor streams inside `weights` (64-byte aligned).                                                                      */
int hg_tp_is(const float* const* src, const int64_t* src_stride, int nsrc, const float* h2_node, const float* h2_edge,
             int hidden, const float* wig, int nW, const int32_t* wig_off, const float* weights, const int32_t* seg_table,
             const int32_t* block_table, const int32_t* phase_table, const int32_t* group_table, const int32_t* item_table,
             const int32_t* part_table, const int32_t* part_table_host, int nparts, const int32_t* row_table, int lds_bytes,
             const int64_t* const* src_idx, int rot_mask, const int64_t* edge_perm, const int32_t* run_id, float* out, int64_t out_stride,
             int64_t rows, void* stream);

/* e3nn NormActivation of a ResidualBlock with nonlinearity_type = "norm" (hamgnn/nn/interaction_blocks.py:311-330 -> e3nn.nn.NormActivation with the
 * scalar nonlinearity ShiftedSoftPlus as given, normalize = True, epsilon = 1e-8, no bias; the head's HamLayers, hamgnn/models/hamgnn_output.py:38-58)
 * on planar rows: every irrep copy is scaled by ssp(n) / n, n = max(|x|, eps).  chan_tab int32[nchan][2] = {float offset of the copy's first
 * component, component stride | components << 16}; D = row width (channel-padding slots come out as zeros).  hg_norm_act_backward: its data
 * gradient for the gradient rows gy of the output (training: Model.py:150-196).                                                              */
int hg_norm_act(const float* x, int64_t x_stride, const int32_t* chan_tab, int nchan, float eps, int64_t rows, float* out, int64_t out_stride, int D,
                void* stream);
int hg_norm_act_backward(const float* x, int64_t x_stride, const float* gy, int64_t gy_stride, const int32_t* chan_tab, int nchan, float eps,
                         int64_t rows, float* gx, int64_t gx_stride, int D, void* stream);

/* Compile-time shape of the loaded library (no reference counterpart; host-only): what = 0 waves per workgroup of hg_tp_is, 1 ... of its lite_mode
 * instantiation, 2 request-ring depth of the lite_mode streams; -1 for anything else.  The host planner
 * (hamgnn_amd/plan.py) deals work to that many streams / waves and refuses to launch when its settings differ.                                 */
int hg_build_config(int what);

/* Many small independent matrix products in one launch (csrc/block_gemm.hip):  C_u = scale_u * op(A_u) @ op(B_u).  Replaces the per-irrep
 * products of a MessagePackBlock's two trailing Linears -- linear_scaler.linear_out @ linear_out / sqrt(mul_k)
 * (hamgnn/nn/message_passing.py:122-130, nn/tensor_products.py:118-140) -- in the device-side weight repack and, transposed, in the backward of
 * those Linears: 13 irreps x 2 branches per block as library GEMMs before.  a, b: DEVICE fp32 buffers the units index into; c: DEVICE fp32 or
 * fp64 (c_is_double) buffer; units int32[nunits][12] = {a_off, a_ld, a_trans, b_off, b_ld, b_trans, c_off, c_ld, M, N, K, scale (float bits)}
 * (element offsets; a_trans: op(A)[m, k] = A[a_off + k * a_ld + m]); max_tiles = max over the units of ceil(M / 64) * ceil(N / 64).
 * fp64 accumulation.                                                                                                            */
int hg_block_gemm(const float* a, const float* b, void* c, int c_is_double, const int32_t* units, int nunits, int max_tiles, void* stream);

/* r6: the split-half-precision twins of the radial weights inside a packed weight blob (hamgnn_amd/plan/program.py:w3_split_fill; read by hg_tp_is when the part record's
 * field [12] is set) recomputed on the device after the fp32 blocks were rewritten in place (training: hamgnn_amd/repack.py).  Pair k: hi / lo halves of
 * weights[src_even[k]] and weights[src_odd[k]] (x scale), packed into the dwords dst_hi[k] / dst_lo[k]; *maxabs = max |x scale| (the caller keeps the fp32 form of the
 * radial scale while it exceeds the f16 range).  No counterpart in the reference (the reference has no packed weights); replaces 18 torch launches per program and step. */
int hg_w3_split_refill(float* weights, const int64_t* src_even, const int64_t* src_odd, const int64_t* dst_hi, const int64_t* dst_lo, int64_t npairs,
                       float scale, float lo_scale, float* maxabs, void* stream);

/* Weight gradient of an o3.Linear on planar rows, all paths in one launch (csrc/linear_wgrad.hip): what torch.autograd computes for the weight of
 * the e3nn o3.Linears of the path when the reference trains (hamgnn/models/Model.py:150-196; nn/interaction_blocks.py:332-358,
 * nn/convolution.py:127, models/hamgnn_output.py:49-58).  units int32[nunits][8] = {x_off, x_mulp, g_off, g_mulp, 2 l + 1, first input channel,
 * first output channel, output channels (<= 64)} (hamgnn_amd/ops.py:linear_wgrad_units); partial: [ceil(rows / 1024)][nunits][16][64] floats of
 * scratch (hg_scratch_bytes), block (chunk, unit) = sum over the chunk's (row, component) of x[., u0 + a] g[., v0 + b]; the caller adds the
 * chunks in a fixed order and applies 1 / sqrt(fan_in).                                                                                  */
int hg_linear_wgrad(const float* x, int64_t x_stride, const float* g, int64_t g_stride, int64_t rows, const int32_t* units, int nunits,
                    float* partial, void* stream);

/* Fused WEIGHT gradients of the weighted tensor-product branches of a MessagePackBlock (csrc/tp_wgrad.hip): what torch.autograd computes
 * for o3.TensorProduct.weight, LinearScaleWithWeights.linear_out.weight and the trailing o3.Linear of
 * hamgnn/nn/message_passing.py:112-160, 191-231 (the reference has no hand-written backward).  Tables from hamgnn_amd/plan.py:
 * build_tp_wgrad_fused (WgFused): units int32[nunits][32], weights (B-operand fragments per unit and 16-row tile), chtab (radial channel
 * of every row).  src[slot]: planar per-edge source rows in the edge frame (sender rows, receiver rows, edge rows), g: gradient of the
 * block's output rows (edge frame), h_node / h_edge: hidden rows [rows, hidden] of the two radial generators (h_edge may be NULL for
 * one-generator blocks).  Outputs: gs_node / gs_edge [rows, n_channels] = gradient with respect to the last radial layer's OUTPUT
 * (per edge; every channel of every row written once), acc [nsplit][acc_floats] partial sums of the weight gradients in the kernel's
 * fragment layout (the host adds the splits and gathers them into the reference's flat layouts: WgFused.tp_pos / l_pos).
 * grid = (nunits, nsplit).  Returns 0, or a negative code for arguments the kernel cannot run.                                        */
int hg_tp_wgrad(const float* const* src, const int64_t* src_stride, int nsrc_slots, const float* g, int64_t g_stride,
                const float* h_node, const float* h_edge, int64_t h_stride, int hidden,
                float* gs_node, int64_t gs_node_stride, float* gs_edge, int64_t gs_edge_stride,
                float* acc, int64_t acc_floats, int nsplit, const int32_t* units, int nunits, const float* weights, const int32_t* chtab,
                int lds_bytes, int64_t rows, void* stream);

/* A chain of row-local stages on planar feature rows (csrc/rowprog.hip; tables: hamgnn_amd/plan.py:build_row_program): HamLayer.forward
 * (hamgnn/models/hamgnn_output.py:51-58) = linear_transform(x + Linear2(Gate(Linear1(x)))) -- ResidualBlock.forward
 * (hamgnn/nn/interaction_blocks.py:332-358) followed by an o3.Linear -- as ONE pass: 16 rows staged in LDS, every stage LDS -> LDS.
 * x [rows, din] (optionally gathered by row_idx), y [rows, dout] (+ up to two residual row tensors added on the way out).
 * stages int32[nstages][24], units int32[.][12], weights (A-operand fragments), act_tab / out_tab int32[.][2] (the gates' tables as
 * hg_gate's, concatenated: nact / nout entries in all; staged in LDS when they fit), consts_host float[5] (normalize2mom constants by activation id, HOST memory), in_buf / out_buf (which of
 * the two LDS buffers holds the input / the result), rs_a / rs_b (LDS row strides in floats), strip (floats per wave for activations). */
int hg_row_program(const float* x, int64_t x_stride, const int64_t* row_idx, int din, float* y, int64_t y_stride, int dout,
                   const float* res0, int64_t res0_stride, const float* res1, int64_t res1_stride,
                   const int32_t* stages, int nstages, const int32_t* units, const float* weights, const int32_t* act_tab,
                   const int32_t* out_tab, int nact, int nout, const float* consts_host, int in_buf, int out_buf, int rs_a, int rs_b, int strip,
                   int64_t rows, void* stream);

/* torch_scatter.scatter(messages, receiver, dim_size=N) of ConvBlockE3.forward (hamgnn/nn/convolution.py:147-149) as a
 * deterministic segmented reduction: out[n] = sum_{q in [rowptr[n], rowptr[n+1])} msg[perm[q]].                     */
int hg_segment_sum(const float* msg, int64_t msg_stride, const int64_t* rowptr, const int64_t* perm, int64_t N, int Dp,
                   float* out, int64_t out_stride, void* stream);

/* e3nn Gate of ResidualBlock (hamgnn/nn/interaction_blocks.py:311-323, 348) on planar rows.  Tables from
 * hamgnn_amd/plan.py:gate_tables_compact: act_tab int32[nact][2] = {input index, act id (0 none,1 ssp,2 tanh,3 silu,4 abs)} -- the
 * row's distinct activated scalars (scalars and gate channels), each evaluated ONCE per row; out_tab int32[Dout][2] =
 * {source code, gate code}: source code = input index | (0x40000000 | act slot) | -1 (structural zero), gate code = act slot | -1.
 * consts[act id] = normalize2mom constant.                                                                            */
int hg_gate(const float* x, int64_t x_stride, const int32_t* act_tab, int nact, const int32_t* out_tab, int Dout, const float* consts,
            int64_t rows, float* out, int64_t out_stride, void* stream);

/* data gradient of hg_gate (same tables): gx [rows, Din] from the gate's input rows x and the gradient gy [rows, Dout] of its output:
 * activated scalars gy * act'(x), gated components gy * act(gate), gate channels act'(gate) * sum gy * x over the components they gate.  */
int hg_gate_backward(const float* x, int64_t x_stride, const float* gy, int64_t gy_stride, const int32_t* act_tab, int nact,
                     const int32_t* out_tab, int Dout, const float* consts, int64_t rows, int Din, float* gx, int64_t gx_stride, void* stream);

/* y = a + b (+ c) on [rows, D] planar rows: ResidualBlock "+x" (interaction_blocks.py:355-356), ConvBlockE3 "+= skip"
 * (convolution.py:155-156).  c may be NULL.                                                                         */
int hg_add_rows(const float* a, int64_t sa, const float* b, int64_t sb, const float* c, int64_t sc, int64_t rows, int D,
                float* out, int64_t so, void* stream);

/* layout conversion e3nn [u][a] <-> planar [a][mulp] via an index map (int32[D_e3nn] -> planar index).  hg_from_planar is a plain
 * column gather out[r][k] = xp[r][map[k]] (map[k] = -1: 0) and also splits the rows of the data-gradient programs.             */
int hg_to_planar(const float* x, int64_t rows, int D, const int32_t* map, float* out, int Dp, void* stream);
int hg_from_planar(const float* xp, int64_t rows, int Dp, const int32_t* map, float* out, int D, void* stream);

/* embedding source rows: out[e][t] = Ts[z[src[e]]][t] + Td[z[dst[e]]][t]   (PairInteractionEmbeddingBlock.forward,
 * hamgnn/nn/embeddings.py:325-326 with one-hot node attrs: the two o3.Linear are row look-ups);
 * idx_b == NULL => single table (chemical embedding AtomwiseLinear, toolbox/nequip/nn/_atomwise.py:55-57).           */
int hg_embed_lookup(const float* Ta, const float* Tb, const int64_t* z, const int64_t* idx_a, const int64_t* idx_b,
                    int64_t rows, int T, int Tp, float* out, void* stream);

/* Read-out head, stage 1 (hamgnn/models/hamgnn_output.py:851-891 merge_tensor_components + :1056-1096 reorder_matrix;
 * SOC/su2: hamgnn/nn/tensor_decomposition.py:553-603 E3TensorDecomposition.get_H + hamgnn_output.py:3149-3152):
 * coeff: planar rows of the HamLayer output regrouped by (L,p) (rotated frame if wig != NULL -> un-rotated here);
 * slot_tab int32[nslots][4] = {L, component, planar index of component 0, component stride} of every coefficient read;
 * cg_ptr/idx/val: CSR rows = output elements (after reorder and sign), columns = slots; Hraw[e][nout]
 * (nout = nao^2, or 2 (2 nao)^2 = [real plane | imag plane] for su2).                                                   */
int hg_ham_merge(const float* coeff, int64_t c_stride, const float* wig, int nW, const int32_t* wig_off,
                 const int32_t* slot_tab, int nslots, const int32_t* cg_ptr, const int32_t* cg_idx, const float* cg_val,
                 int nout, int64_t rows, float* Hraw, void* stream);

/* Read-out head in ONE pass (non-SOC branch, hamgnn_output.py:3772-3799): stage 1 (merge_tensor_components + reorder_matrix,
 * tables as for hg_ham_merge with nout = nao^2) and stage 2 (symmetrize :1231-1285, + H0, orbital masks :2288-2365) for PAIRS of
 * rows (pair_a[p], pair_b[p]) = (edge, inverse edge) -- both merged blocks live in the LDS and are symmetrised against each
 * other, so no intermediate Hraw exists.  pair_b == NULL: every row pairs with itself (on-site blocks; pair_a == NULL: row p).
 * Each row must occur in exactly one pair.  H: [rows, nao^2] result rows (e.g. the row range of the [N+E, nao^2] output).
 * c_width: floats per coefficient row; lmax_ham: largest L among the slots (the leading Wigner blocks l <= lmax_ham of each row are
 * staged).  flags / sign / masks as hg_ham_finish.  Tables must fit 64 KB of LDS (else use the two-stage entry points).      */
int hg_ham_readout(const float* coeff, int64_t c_stride, int c_width, const float* wig, int nW, const int32_t* wig_off,
                   int lmax_ham, const int32_t* slot_tab, int nslots, const int32_t* cg_ptr, const int32_t* cg_idx,
                   const float* cg_val, int nnz, int nao, const int64_t* pair_a, const int64_t* pair_b, int64_t npairs,
                   const float* H0, const float* orb_mask, int mask_w, const int64_t* z, const int64_t* idx_a, const int64_t* idx_b,
                   float sign, int flags, float* H, void* stream);

/* Read-out head, stage 2 (:1231-1285 symmetrize, :3782-3795 +H0, :2288-2365 orbital masks):
 * H[e] = mask(z_a, z_b) * (0.5 (Hraw[e] + sign * Hraw[inv[e]]^T) + H0[e]);  inv == NULL => on-site (own transpose).
 * Hraw rows are h_stride floats apart; orb_mask is [Z][mask_w] and indexed modulo mask_w (nao = 2 mask_w for the su2
 * spin-block matrices, :3163-3168); flags: bit 0 symmetrise, bit 1 add H0 after the mask (SOC, :3603-3609).            */
int hg_ham_finish(const float* Hraw, int64_t h_stride, const int64_t* inv, const float* H0, const float* orb_mask, int mask_w,
                  const int64_t* z, const int64_t* idx_a, const int64_t* idx_b, int nao, float sign, int flags, int64_t rows,
                  float* H, void* stream);

/* CorrProductBlock's symmetric contraction (hamgnn/nn/interaction_blocks.py:234-260 -> toolbox/mace/modules/
 * symmetric_contraction.py:212-230 with U_matrix_real of toolbox/mace/tools/cg.py:89-131), the nu <= 2 terms, on planar hidden
 * node rows h [N, .] (num_hidden x every node irrep); a `correlation: 3` block adds its nu = 3 term with hg_sym_contraction3:
 *   out[o, c] = sum_x ( sum_kap U1[o, x, kap] W1[z, kap, c] + sum_{i, kap} U2[o, x, i, kap] W2[z, kap, c] x[c, i] ) x[c, x]
 * with the U tensors given sparsely (plan.py:sym_contraction_tables): ell_off / out_off = planar offsets of the input
 * components / output elements, ptr1/ent1 {x, kappa, -, value bits} and ptr2/ent2 {x, i, kappa, value bits} CSR rows per output
 * element; W1 [num_elements][K1][C], W2 [num_elements][K2][C] = the contractions' weights concatenated over the targets.   */
int hg_sym_contraction(const float* h, int64_t h_stride, const int64_t* z, int64_t N, int C, int num_ell, const int32_t* ell_off,
                       int nout, const int32_t* out_off, const int32_t* ptr1, const int32_t* ent1, const int32_t* ptr2,
                       const int32_t* ent2, const float* W1, int K1, const float* W2, int K2, float* out, int64_t out_stride,
                       void* stream);

/* The nu = 3 term of the same contraction for a block built with `correlation: 3` (symmetric_contraction.py:148-230: the main einsum over
 * U_matrix_real(.., correlation = 3), cg.py:16-131), ADDED onto the rows hg_sym_contraction wrote (csrc/corr3.hip):
 *   out[o, c] += sum_{(x, i, j, kap, v) in ent3[o]} v W3[z, kap, c] h[c, x] h[c, i] h[c, j]
 * ptr3 int32[nout + 1], ent3 int32[.][5] = {x, i, j, kappa, value bits} (plan.py:sym_contraction_tables), W3 [num_elements][K3][C].           */
int hg_sym_contraction3(const float* h, int64_t h_stride, const int64_t* z, int64_t N, int C, int num_ell, const int32_t* ell_off,
                        int nout, const int32_t* out_off, const int32_t* ptr3, const int32_t* ent3, const float* W3, int K3,
                        float* out, int64_t out_stride, void* stream);

/* zero-point energy shift (hamgnn_output.py:3971-3981; SOC spin-diagonal real blocks :3892-3913), in place:
 *   dE = sum_{S>thr}(H - Href) / sum_{S>thr} S ;  H -= dE * S    -- one dE per call (= per batch, as the reference).
 * soc != 0: H/Href rows are (2 nao)^2, S rows nao^2; (uu+dd) differences, denominator 2 sum S, both diagonal blocks shifted.
 * partial_scratch: 2*nparts doubles of caller scratch (fixed-order two-stage reduction => deterministic).              */
int hg_zero_point_shift(float* H, const float* Href, const float* S, int64_t rows, int nao, int soc, float threshold,
                        double* partial_scratch, int nparts, float* shift_out, void* stream);

/* SOC / so3 branch (hamgnn_output.py:3026-3144).  hg_block_mean = symmetrize_orbital_coefficients (:2367-2431): each element
 * of the nao x nao xi matrix -> mean over its (row shell, col shell) block; tab int32[nao^2][4] = {r0, r1, c0, c1}.          */
int hg_block_mean(const float* x, int64_t x_stride, const int32_t* tab, int nao, int64_t rows, float* out, void* stream);

/* spin-block assembly: A_k = antiherm(xi * L[...,k]) through inv (NULL => on-site);
 * real = [[H, A_y],[A_y, H]] + H0r (spin-diagonal H0r skipped if zero_diag: add_H_nonsoc, :3034-3049);
 * imag = [[A_z, A_x],[-A_x, -A_z]] + H0i.   H [rows,nao^2], xi [rows,nao^2], L [rows,nao^2,3], outputs [rows,(2 nao)^2].  */
int hg_soc_assemble(const float* H, const float* ksi, const float* L, const int64_t* inv, const float* H0r, const float* H0i,
                    int nao, int symmetrize, int zero_diag, int64_t rows, float* out_real, float* out_imag, void* stream);

/* k-space assembly of ONE crystal for the band-energy step (hamgnn/models/hamgnn_output.py:1776-1905 inside
 * calculate_band_energies): X(k)[i a, j b] = delta_ij X_on[i][a][b] + sum_{e: i->j} exp(2 pi i k . nbr_shift_e) X_off[e][a][b], written in
 * the COMPACT orbital basis (the reference builds the nao_max-padded matrix, then masked_selects the valid orbitals).  Edges are
 * grouped by atom pair on the host (pair_ptr [npairs+1], pair_edges: crystal-local edge ids, pair_ij [npairs][2]); one block owns one
 * (pair, k) -- fixed summation order instead of index_put(accumulate=True) atomics.  orank [n_atoms][nao]: rank of an orbital in its
 * atom's valid set or -1; ooff [n_atoms]: first compact index of an atom; Hk: [nk][M][M] complex64 (re, im), ZERO-initialised by the
 * caller.  The generalized eigenproblem that follows (Cholesky of S(k), eigh) is hipSOLVER's job, reached through torch.linalg.   */
int hg_hk_assemble(const float* on, const float* off, const float* nbr_shift, const float* kvec, int nk, const int64_t* pair_ptr,
                   const int64_t* pair_edges, const int64_t* pair_ij, int64_t npairs, int n_atoms, int nao, const int32_t* orank,
                   const int32_t* ooff, int M, float* Hk, void* stream);

/* e3nn o3.Linear on planar rows as one streaming pass (every o3.Linear of the hot path: hamgnn/nn/interaction_blocks.py:332-358,
 * 141-152; nn/convolution.py:127; models/hamgnn_output.py:49-58).  Tables from hamgnn_amd/plan.py:linear_tables:
 * items int32[nitems][2] = {unit, component a}: the wave units of one block of 32 rows (one workgroup per four of them),
 * units int32[nunit][8] = {out_off, out_mulp, row tiles, channels to store, path_begin, path_end, 0, 0} (<= 64 output channels each),
 * paths int32[npath][4] = {in_off, in_mulp, K groups of 16, weight offset}; weights: MFMA A fragments of the normalised blocks.
 * y[row] = Linear(x[row]) (+ res0[row] + res1[row] if given: the residual / skip adds of ResidualBlock); every output column is written. */
int hg_linear_planar(const float* x, int64_t x_stride, const int32_t* items, int nitems, const int32_t* units, const int32_t* paths, const float* weights, const float* res0, int64_t res0_stride,
                     const float* res1, int64_t res1_stride, int64_t rows, float* y, int64_t y_stride, void* stream);

/* AttentionBlockE3 / AttentionAggregation (hamgnn/nn/attention.py:91-164, 337-350; heads: hamgnn/nn/attention_utils.py:17-120).
 * K [N, Dp]: planar rows of linear_key(node_feats) (key = K[sender], query = K[receiver], :339-340); head_tab int32[Dp]: head of a
 * planar column (a head is a channel range of every irrep block) or -1 for padding columns; 1 <= H <= 8.
 *   logits[e, h] = soft_unit_step(cut_param * (1 - length_e / cutoff)) * scale * <K[dst_e] | K[src_e]>_h ,  scale = 1 / sqrt(head dim);
 *   cut_param: ONE device float (SoftUnitStepCutoff.cut_param, a learnable parameter, hamgnn/utils/cutoff_functions.py:65-100).      */
int hg_attn_logits(const float* K, int64_t k_stride, const int64_t* src, const int64_t* dst, const float* length,
                   const int32_t* head_tab, int Dp, int H, const float* cut_param, float cutoff, float scale, int64_t E,
                   float* logits, void* stream);
/* out[n, col] = sum_{q in [rowptr[n], rowptr[n+1])} softmax_q(logits[perm[q], head(col)]) * V[perm[q], col]  -- torch_geometric's
 * softmax (max-shifted exp over a node's incoming edges, divided by (sum + 1e-16)) and torch_scatter.scatter, fixed order.           */
int hg_attn_aggregate(const float* logits, int H, const float* V, int64_t v_stride, const int64_t* rowptr, const int64_t* perm,
                      const int32_t* head_tab, int64_t N, int Dp, float* out, int64_t out_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif

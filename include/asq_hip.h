/*
 * asq_hip.h -- C-ABI of libasq_hip.so: the MI355X (gfx950 / CDNA4) implementation of
 * AutoSmoothQuant's W8A8 linear hot path.
 *
 * This is the drop-in boundary.  Each entry point names the reference interface it
 * replaces (paths are into the upstream AutoSmoothQuant tree):
 *
 *   csrc/int8gemm/bindings.cpp:145-155   pybind class I8CUGEMM (5 methods)
 *   autosmoothquant/layers/nn/linear.py  the eager-PyTorch quantise / dequantise code
 *                                        around the GEMM (:83-106, :158-208, :278-302)
 *
 * Conventions
 *   - plain pointers + sizes; no torch / pybind types.  All pointers are DEVICE pointers
 *     on the current HIP device unless stated otherwise.  Tensors are dense row-major.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  The library
 *     keeps no state: no handle, no lock, no captured stream (the reference captures one
 *     stream at construction, bindings.cpp:13, and serialises callers on a mutex,
 *     cublasINT8MMWrapper.cc:228,353 -- deliberately not reproduced).
 *   - every call returns an int status: 0 = ok, <0 = argument error (ASQ_ERR_*),
 *     >0 = hipError_t from the launch.  asq_last_error() returns a thread-local message.
 *   - nothing is allocated inside; calls that need scratch take a caller-owned workspace.
 *   - env ASQ_DEBUG_SYNC=1: hipStreamSynchronize + error check after every launch (the
 *     analogue of FT_DEBUG_LEVEL=DEBUG, csrc/int8gemm/cuda_utils.h:88-104).
 */
#ifndef ASQ_HIP_H
#define ASQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASQ_VERSION 126 /* 0.1.8: + asq_rmsnorm, asq_silu_mul (caller-side glue of the reference's module composition: the norm and the gated activation as one pass each, floating outputs); 0.1.7: + asq_fp8_grouped_gate_up_supported, asq_linear_fp8_grouped_gate_up (FP8LinearDynamic experts' w1 || w3 as one grouped launch with the SiLU * up epilogue); 0.1.6: + asq_rope (caller-side glue: rotary embedding of a q / k projection's output in one pass); 0.1.5: + asq_silu_mul_quantize_fp8 (SiLU * up fused with the per-token e4m3 quantiser of the FP8 linear behind it); 0.1.4: + asq_linear_w8a8_gate_up_q8 (gate || up with an int8-out epilogue for per-tensor consumers); 0.1.3: + asq_grouped_gate_up_supported, asq_linear_w8a8_grouped_gate_up (Mixtral's w1 || w3 as one grouped launch with the SiLU * up epilogue); 0.1.2: + asq_forward_fused_supported, asq_linear_w8a8_forward_fused (the one-launch forward for decode-sized inputs; asq_linear_w8a8_forward takes it
                           * by itself where it wins); ASQ_ROCTX=1 ranges.  0.1.1: + offset operand images (asq_*_off); workspace sizes include the 8 KiB header
                           * (asq_workspace_init is mandatory for a workspace handed to a GEMM entry point); asq_silu_mul_quantize's `per_token` is a bit field (bit 0
                           * per-token, ASQ_SILU_FAST) */

/* element types of floating tensors crossing the boundary */
#define ASQ_F32 0
#define ASQ_F16 1
#define ASQ_BF16 2

/* activation-quantisation mode (the `act_quant` ctor argument + class of the module) */
#define ASQ_ACT_ROUND 0     /* per-tensor, scale folded upstream: round+clamp only (linear.py:95-96,174) */
#define ASQ_ACT_DIV 1       /* per-tensor, x / quant_scale in x's dtype (linear.py:289-292)               */
#define ASQ_ACT_PER_TOKEN 2 /* per-token dynamic absmax/127 (linear.py:88-92,164-168,283-287)            */

/* association order of the dequant epilogue */
#define ASQ_EPI_SCALE_FIRST 0 /* (s_col*s_row) * float(acc) + bias   -- nn modules, linear.py:93,104,200-206 */
#define ASQ_EPI_ACC_FIRST 1   /* (float(acc)*s_col) * s_row  + bias  -- functional/quantization.py:103-120   */

/* status codes */
#define ASQ_OK 0
#define ASQ_ERR_NULL (-1)      /* required pointer is NULL          */
#define ASQ_ERR_DIM (-2)       /* negative / overflowing dimension  */
#define ASQ_ERR_DTYPE (-3)     /* unknown dtype / mode enum         */
#define ASQ_ERR_ALIGN (-4)     /* pointer not aligned to its element */
#define ASQ_ERR_WORKSPACE (-5) /* workspace missing or too small    */

int asq_version(void);
const char *asq_last_error(void);

/* ---- K1: I8CUGEMM::linear_a8_w8_o32_  (bindings.cpp:69-84 -> cublasINT8MMWrapper.cc:224-354)
 * out[M,N] (int32) = x[M,K] (int8) . w[N,K]^T (int8); alpha=1, beta=0; exact.
 * Also serves I8CUGEMM::linear_a8_w8_o32 (bindings.cpp:52-67): that flavour differs only in
 * cuBLASLt's COL32 operand layouts, which have no gfx950 counterpart. */
int asq_gemm_i8_i32(const int8_t *x, const int8_t *w, int32_t *out,
                    int64_t M, int64_t N, int64_t K,
                    void *workspace, size_t workspace_bytes, void *stream);

/* Optional scratch for the GEMM entry points.  When the M x N tile grid cannot fill the 256 CUs
 * (e.g. OPT-13B fc2 at 256 rows: 20 tiles) the dispatcher splits K across CUs, writes exact int32
 * partial slabs into `workspace` and reduces them in a second launch that applies the epilogue.  The 128 x 128 kernel (M ~ 192 ... 512 against LLaMA-sized weights)
 * reduces its K splits INSIDE the launch instead (round 5): write-through 64 KiB register images behind the header, one ticket per tile in the header, the last
 * arriver adds the others and runs the epilogue -- the same tickets-return-to-zero contract as the weight-streaming kernel below (csrc/asq_gemm_p8q2.h).
 * Independently of the workspace, a grid a few tiles over a multiple of 256 (1536 x 11008: 258 tiles) runs its last tile columns as a launch of
 * 128 x 128 tiles instead of paying a second wave for two tiles (-3 ... -19 % of the call); with a workspace that remainder may split K too.
 * asq_gemm_workspace_bytes() returns the size that enables this for a shape (0 = never needed).
 * workspace may be NULL / smaller (then fewer or no K splits are used); it must be 256-B aligned, initialised once with
 * asq_workspace_init() and not shared by concurrently running calls (contract below). */
size_t asq_gemm_workspace_bytes(int64_t M, int64_t N, int64_t K);

/* Workspace contract.  A workspace is [ header: asq_workspace_header_bytes() = 8 KiB | scratch ].  The header holds a magic word and the
 * arrival tickets of the in-launch reductions (the weight-streaming kernel, few rows against a large weight: decode, cfg1, OPT fc2 at 32 rows per GPU --
 * csrc/asq_gemm_wstream.h; the K splits of the 128 x 128 kernel, csrc/asq_gemm_p8q2.h); partial tiles / split-K slabs live behind it.
 *   - asq_workspace_init() must run ONCE on a buffer (on the stream it will be used on, or synchronised) before the first GEMM call that receives it;
 *     every launch leaves the tickets at zero, also under hipGraph replay, so there is nothing to reset between calls or shapes;
 *   - a buffer is used by one launch at a time (one stream): concurrent launches on one workspace corrupt each other's tickets;
 *   - sizes returned by asq_gemm_workspace_bytes / asq_linear_w8a8_workspace_bytes include the header; the header is always at offset 0;
 *   - a launch that finds no magic word, or a ticket above its contributor count, executes s_trap (the stream reports a launch failure)
 *     instead of returning wrong sums.  Passing NULL / a buffer smaller than the header selects the first-generation kernels, which need none.
 * 256-B aligned. */
int asq_workspace_init(void *workspace, size_t workspace_bytes, void *stream);
size_t asq_workspace_header_bytes(void);

/* ---- K3/K4/K5: I8CUGEMM::linear_a8_w8_o8 / linear_a8_w8_o8_ / linear_a8_w8_b8_o8_
 * (bindings.cpp:86-142 -> cublasINT8MMWrapper.cc:360-672)
 * out[M,N] (int8) = sat_i8(round_half_even(alpha * float(acc) + beta * float(c)))
 * c = the previous contents of `out` when beta != 0 (cuBLASLt C==D in-place), ignored when beta == 0.
 * For linear_a8_w8_b8_o8_ the caller pre-fills out with the broadcast int8 bias (bindings.cpp:132). */
int asq_gemm_i8_i8(const int8_t *x, const int8_t *w, int8_t *out,
                   int64_t M, int64_t N, int64_t K, float alpha, float beta,
                   void *workspace, size_t workspace_bytes, void *stream);

/* ---- GEMM prologue: the activation quantisers of layers/nn/linear.py
 * x is [M,K] of x_dtype.  xq is int8 [M,K].
 * ASQ_ACT_PER_TOKEN: s_row[m] = (absmax_k x[m,k] / 127 in x_dtype) as f32;
 *                    xq = int8(clamp(rne(f32(x) / s_row[m]), -128, 127)); s_row required.
 * ASQ_ACT_ROUND    : xq = int8(clamp(rne(x), -128, 127));           quant_scale, s_row ignored.
 * ASQ_ACT_DIV      : xq = int8(clamp(rne(x_dtype(f32(x) / quant_scale)), -128, 127)).
 * NaN quantises to 0 and +-inf saturates, as on the reference's CPU path. */
int asq_quantize_act(const void *x, int x_dtype, int mode, float quant_scale,
                     int8_t *xq, float *s_row, int64_t M, int64_t K, void *stream);

/* ---- N1: norm -> int8 fusion.  RMSNorm (bias == NULL; HF LlamaRMSNorm arithmetic) or LayerNorm
 * (bias != NULL; OPT) whose weight (and bias) already carry 1/input_scale (reference models/llama.py:27-37,
 * models/opt.py:20-29), fused with the activation quantiser of the linears that consume it -- the reference's
 * own unbuilt LayerNormQ (layers/nn/fused.py:10-15).  x, weight, bias share x_dtype; K <= 8192 (16-bit) / 4096 (f32).
 * per_token = 0: xq = int8(clamp(rne(y)));  1: s_row[m] = absmax(y)/127 (in x_dtype), xq = int8(clamp(rne(y / s_row))).
 * Every fp32 operation is fixed (per-thread accumulation order, butterfly reduction, 1/sqrt from one IEEE sqrt and
 * one IEEE division) and restated step by step in oracle/n1.py: the HIP result equals that restatement bit for bit.
 * Against "norm in PyTorch, then asq_quantize_act" (ATen's own reduction order) y can differ in its last place, i.e.
 * the int8 differs by +-1 only where y lies within ~1e-6 relative of a rounding boundary. */
int asq_norm_quantize(const void *x, int x_dtype, const void *weight, const void *bias, float eps, int per_token,
                      int8_t *xq, float *s_row, int64_t M, int64_t K, void *stream);

/* The residual-add form -- the reference's dq_add_layernorm_q (csrc/kernels/fused.cu:5-25;
 * layers/functional/fused.py:5-12) with a floating `x` (here the producing GEMM's epilogue has already dequantised):
 *   h_out = x_dtype(residual + x)        (the new residual stream, written once)
 *   xq    = quantised norm(h_out)        (exactly asq_norm_quantize applied to h_out)
 * x, residual, h_out [M,K] of x_dtype; h_out may alias residual or x. */
int asq_add_norm_quantize(const void *x, const void *residual, void *h_out, int x_dtype, const void *weight, const void *bias,
                          float eps, int per_token, int8_t *xq, float *s_row, int64_t M, int64_t K, void *stream);

/* SiLU(gate) * up fused with the quantiser of the linear that consumes it (LLaMA / Mixtral down_proj / w2,
 * a W8A8BFP32OFP32LinearWithQuantScale): per_token = 1 -> dynamic row scales (linear.py:283-287),
 * per_token = 0 -> x / quant_scale in x_dtype (linear.py:289-292).  gate, up [M,K] of x_dtype; K <= 16384 (16-bit).
 * silu(g) = x_dtype(g / (1 + exp(-g))) with exp evaluated by a fixed fp32 operation sequence (oracle/n1.py::exp_det
 * repeats it), so this kernel, too, is bit-identical to its oracle.
 * per_token | ASQ_SILU_FAST (opt-in): silu from the hardware transcendentals (v_exp_f32, v_rcp_f32: ~1 ulp) instead of the fixed operation sequence -- about
 * 1.3x the rows per second (the exact kernel is VALU-bound at 3.6 TB/s); the int8 result differs from the exact kernel by at most +-1 (bf16: +-2 on < 1e-4 of the elements), at rounding boundaries. */
#define ASQ_SILU_FAST 2
int asq_silu_mul_quantize(const void *gate, const void *up, int x_dtype, int per_token, float quant_scale,
                          int8_t *xq, float *s_row, int64_t M, int64_t K, void *stream);

/* ---- fused GEMM + dequant/bias epilogue (replaces the i32 round trip of linear.py:97-105;
 * subsumes csrc/kernels/linear.cu:201-291 `linear_a8_w8_bfp32_ofp32`)
 * out[M,N] (out_dtype) = epi(acc[m,n]) where acc = xq . w^T (int32, exact)
 *   ds    = (s_col ? s_col[n] : s_scalar) [* s_row[m] if s_row]
 *   out   = ds * float(acc) (+ bias[n])         (ASQ_EPI_SCALE_FIRST; two fp32 roundings, no FMA)
 * s_row f32[M] (per-token) | NULL;  s_col f32[N] (per-channel / QKV segments) | NULL;  bias f32[N] | NULL */
int asq_linear_w8a8(const int8_t *xq, const int8_t *w, void *out, int out_dtype,
                    int64_t M, int64_t N, int64_t K,
                    float s_scalar, const float *s_row, const float *s_col, const float *bias,
                    int epi_order, void *workspace, size_t workspace_bytes, void *stream);

/* ---- the same GEMM with an int8-OUT epilogue that feeds the NEXT W8A8 linear directly (the int8-out idea of
 * bindings.cpp:86-142 with the consumer's real prologue):  y = mid_dtype(dequant(acc) (+bias)) exactly as asq_linear_w8a8
 * would store it; a = relu(y) if act == 1 (OPT fc1 -> fc2, reference models/opt.py:127-128); then
 *   qmode ASQ_ACT_ROUND: out_q = int8(clamp(rne(a)))                                   (consumer: per-tensor W8A8BFP32OFP32Linear)
 *   qmode ASQ_ACT_DIV  : out_q = int8(clamp(rne(mid_dtype(a / quant_scale))))          (consumer: per-tensor ...WithQuantScale)
 * Bit-identical to "asq_linear_w8a8, activation, asq_quantize_act": every step is elementwise. */
int asq_linear_w8a8_q8(const int8_t *xq, const int8_t *w, int8_t *out_q, int mid_dtype,
                       int64_t M, int64_t N, int64_t K,
                       float s_scalar, const float *s_row, const float *s_col, const float *bias, int epi_order,
                       int act, int qmode, float quant_scale, void *workspace, size_t workspace_bytes, void *stream);

/* ---- grouped launch: ngroups independent linears (Mixtral experts, reference models/mixtral.py:99-145 runs
 * them as a Python loop of Int8Linear calls) in ONE kernel launch.
 *   xq  int8 [M,K]   rows sorted by group; group g owns rows group_offsets[g] .. group_offsets[g+1]-1
 *   group_offsets    DEVICE int32 [ngroups+1] (prefix sums; stays on the device: no host sync for the routing)
 *   w   int8 [ngroups][N][K];  s_group f32 [ngroups] (DEVICE): each expert's scalar dequant_scale
 *   s_row f32 [M] | NULL (per-token);  bias f32 [ngroups][N] | NULL;  out [M,N] of out_dtype
 * out[m,n] = (s_group[g(m)] [* s_row[m]]) * float(acc) (+ bias[g(m)][n]) -- per row identical to asq_linear_w8a8
 * on that group's slice.  Needs K % 128 == 0 and 16-B aligned operands; empty groups are fine. */
int asq_linear_w8a8_grouped(const int8_t *xq, const int8_t *w, void *out, int out_dtype,
                            const int32_t *group_offsets, int ngroups,
                            int64_t M, int64_t N, int64_t K,
                            const float *s_group, const float *s_row, const float *bias, void *stream);
/* The same launch with a workspace (contract above: asq_workspace_init once, one launch at a time).  Scheduling inside the launch, for
 * ngroups <= 64 (both entry points): each XCD gets an equal share of the launch's tiles (counted on the device from group_offsets); a group's
 * last tile with <= 128 rows runs as a half tile (half the matrix work instead of a full tile of padding).  With a workspace of
 * asq_grouped_workspace_bytes() (8 KiB header + 64 MiB; 0 = this shape never uses one) the tiles of an XCD's last incomplete round split
 * their K loop into 2..8 pieces that fill the round; the pieces are summed in the launch (exact int32, ticket in the header) before the
 * epilogue.  Mixtral w2 (N 4096, K 14336, 8192 routed rows): 2.25 rounds -> 2 rounds + a short one (517 -> 452 us).  Results are
 * bit-identical to asq_linear_w8a8_grouped. */
size_t asq_grouped_workspace_bytes(int64_t M, int64_t N, int64_t K, int ngroups);
int asq_linear_w8a8_grouped_ws(const int8_t *xq, const int8_t *w, void *out, int out_dtype,
                               const int32_t *group_offsets, int ngroups,
                               int64_t M, int64_t N, int64_t K,
                               const float *s_group, const float *s_row, const float *bias,
                               void *workspace, size_t workspace_bytes, void *stream);

/* ---- whole module forward: W8A8BFP32OFP32Linear / ...QKVLinear / ...LinearWithQuantScale .forward
 * (linear.py:83-106, :158-208, :278-302) = asq_quantize_act + asq_linear_w8a8 on `stream`.
 * out has x's dtype.  workspace: asq_linear_w8a8_workspace_bytes(M,N,K) bytes = [ header + GEMM scratch | int8 activations | row scales ],
 * 256-B aligned; the header part is reserved for EVERY shape, so an initialised buffer can be shared by calls of any shape.  Smaller buffers: with room for
 * header + activations + scales the layout is [ header | activations | scales ] and the GEMM runs without scratch (a shared buffer's header stays intact);
 * only a buffer smaller than that is used as [ activations | scales ] from offset 0 -- a private scratch buffer that never had a header. */
size_t asq_linear_w8a8_workspace_bytes(int64_t M, int64_t N, int64_t K);
int asq_linear_w8a8_forward(const void *x, int x_dtype, const int8_t *w, void *out,
                            int64_t M, int64_t N, int64_t K,
                            int act_mode, float quant_scale,
                            float s_scalar, const float *s_col, const float *bias,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ---- OFFSET OPERAND IMAGES: the same W8A8 linear, bit for bit, on operands that cost the matrix cores less energy.
 * On MI355X the 256 x 256 INT8 GEMM runs against the socket power limit, so its time is its energy, and most of the matrix cores' data-dependent
 * energy is two's-complement sign extension: weights and SmoothQuant-style activations are centred on 0 (profiles/r4_operand_offsets.md).
 * The INT8 MFMA is signed x signed only, but an offset per row is exact and never clamps when it is chosen from the row's own extremes:
 *     x'[m,k] = xq[m,k] + cx[m]      cx[m] = +3 if max_k xq[m,k] <= 124, else -3 if min_k xq[m,k] >= -125, else 0
 *     w'[n,k] = w[n,k]  + cw[n]      cw[n] = min(64, 127 - max_k w[n,k])
 *     sum_k xq[m,k] w[n,k] = sum_k x'[m,k] w'[n,k]  -  cx[m] * sum_k w[n,k]  -  cw[n] * sum_k x'[m,k]
 * The kernel multiplies the primed matrices with the plain K loop and subtracts the two correction terms from its int32 accumulators in front of the
 * epilogue (two's-complement wrap-around, also inside v_mfma_i32_*): the result equals asq_linear_w8a8 on (xq, w) in every bit.  There is no reference
 * counterpart (cuBLASLt sees the raw operands, cublasINT8MMWrapper.cc:224-354); what it replaces is the operand handling in front of
 * asq_linear_w8a8.  4096^3, fp16 out, bench operands: 48.2 -> 44.0 us (56.7 -> 62.1 % of the INT8 peak).
 *
 *   asq_weight_offset_image   once per weight: w [N,K] -> w_off [N,K] (int8) and col_off int32 [N][2] = {cw[n], sum_k w[n,k]}.
 *                             K % 16 == 0, K <= 65536, N % 4 == 0, 16-B aligned pointers.
 *   asq_quantize_act_off      asq_quantize_act (same modes, same arithmetic) emitting x' and row_off int32 [M][2] = {cx[m], sum_k x'[m,k]}.
 *                             K % 8 == 0 (f32: 4), K <= 40960 (f32: 20480), x 16-B aligned, row_off 8-B aligned.
 *   asq_linear_w8a8_off       asq_linear_w8a8 on (x', w') + the two vectors; any out_dtype, K % 128 == 0, 128 <= K <= 65536,
 *                             N % 4 == 0, 16-B aligned operands; always the 256 x 256 kernel, no workspace.
 *   asq_offsets_supported     1 when the dispatcher itself would run (M, N, K) on that kernel (>= 144 tiles of 256 x 256, no
 *                             column remainder launch) and the limits above hold; ASQ_OFFSETS=0 in the environment makes it return 0.
 *   asq_linear_w8a8_forward_off  asq_linear_w8a8_forward with the weight's image: uses the offset path when asq_offsets_supported() and the
 *                             workspace has asq_linear_w8a8_workspace_bytes(); otherwise it IS asq_linear_w8a8_forward (w_off / col_off may be NULL).
 *   asq_norm_quantize_off / asq_add_norm_quantize_off / asq_silu_mul_quantize_off   the N1 fusions (same arithmetic, same arguments + row_off) emitting
 *                             the offset image of their int8 result: xq' - cx[m] is exactly what asq_norm_quantize / ... write.
 *   asq_linear_w8a8_grouped_off   asq_linear_w8a8_grouped_ws on images: xq' from any *_off quantiser (rows sorted by group, row_off [M][2]), w' / col_off
 *                             from asq_weight_offset_image on the [ngroups * N, K] stack (col_off [ngroups][N][2]); out_dtype ASQ_F16 / ASQ_BF16,
 *                             K <= 65536, N % 4 == 0.  Bit-identical to the plain grouped launch, with or without the workspace's K split.
 * Development overrides (read once): ASQ_OFF_CX (default 3), ASQ_OFF_CW (default 64). */
int asq_weight_offset_image(const int8_t *w, int64_t N, int64_t K, int8_t *w_off, int32_t *col_off, void *stream);
int asq_quantize_act_off(const void *x, int x_dtype, int mode, float quant_scale,
                         int8_t *xq_off, float *s_row, int32_t *row_off, int64_t M, int64_t K, void *stream);
int asq_linear_w8a8_off(const int8_t *xq_off, const int8_t *w_off, void *out, int out_dtype,
                        int64_t M, int64_t N, int64_t K,
                        float s_scalar, const float *s_row, const float *s_col, const float *bias, int epi_order,
                        const int32_t *row_off, const int32_t *col_off, void *stream);
int asq_offsets_supported(int64_t M, int64_t N, int64_t K, int out_dtype);
/* 1 when asq_linear_w8a8_forward runs (M, N, K) with x_dtype activations and quantiser mode act_mode as ONE launch -- the activation quantiser (reference
 * layers/nn/linear.py:88-96, :283-292) as the GEMM's prologue, the dequant / bias (linear.py:93-104) as its epilogue -- on the dispatcher's weight-streaming shapes
 * whose int8 activation image (4 / 8 / 16 rows x K bytes) fits in 64 KiB of LDS, K % 128 == 0, 16-byte aligned x and w (assumed here; checked per call):
 *   <= 4 rows: every mode, up to 1024 tiles of 16 channels;   5 .. 16 rows: the per-tensor modes at <= 256 tiles (N <= 4096).
 * Elsewhere the two launches are faster (every block repeats the quantiser's work; measured: profiles/r5_fused_forward_*.txt).  Bit-identical to asq_quantize_act +
 * asq_linear_w8a8.  ASQ_FUSED_FORWARD=0 in the environment switches it off (A/B). */
int asq_forward_fused_supported(int64_t M, int64_t N, int64_t K, int x_dtype, int act_mode);
/* The one-launch forward on request, for any shape its kernel can run: 1 <= M <= 16, K % 128 == 0, rows(M) x K <= 65536 (rows = 4 / 8 / 16: the resident int8
 * activation image), N x K < 2^32, 16-byte aligned x and w; no workspace.  asq_linear_w8a8_forward takes it by itself only where it was measured to win
 * (asq_forward_fused_supported); this entry point exists for callers that prefer one launch regardless (hipGraph
 * node count, stream ordering) and for the parity tests.  Same arguments and results as asq_linear_w8a8_forward.  ASQ_ERR_DIM outside the limits above. */
int asq_linear_w8a8_forward_fused(const void *x, int x_dtype, const int8_t *w, void *out, int64_t M, int64_t N, int64_t K,
                                  int act_mode, float quant_scale, float s_scalar, const float *s_col, const float *bias, void *stream);
int asq_linear_w8a8_grouped_off(const int8_t *xq_off, const int8_t *w_off, void *out, int out_dtype,
                                const int32_t *group_offsets, int ngroups, int64_t M, int64_t N, int64_t K,
                                const float *s_group, const float *s_row, const float *bias,
                                const int32_t *row_off, const int32_t *col_off,
                                void *workspace, size_t workspace_bytes, void *stream);
int asq_norm_quantize_off(const void *x, int x_dtype, const void *weight, const void *bias, float eps, int per_token,
                          int8_t *xq_off, float *s_row, int32_t *row_off, int64_t M, int64_t K, void *stream);
int asq_add_norm_quantize_off(const void *x, const void *residual, void *h_out, int x_dtype, const void *weight, const void *bias,
                              float eps, int per_token, int8_t *xq_off, float *s_row, int32_t *row_off, int64_t M, int64_t K, void *stream);
/* gate || up as ONE GEMM whose epilogue writes SiLU(gate) * up (round 5): the two projections of a gated MLP -- reference models/llama.py:206-211 (HF LlamaMLP:
 * down_proj(act_fn(gate_proj(x)) * up_proj(x))), both W8A8BFP32OFP32Linear over the same quantised input -- with the [M, F] gate and up tensors never reaching HBM.
 *   w_gu int8 [2 F, K]: gate and up ROW-INTERLEAVED in blocks of 16 channels (rows 32 j .. 32 j + 15 = gate channels 16 j .. 16 j + 15, rows 32 j + 16 .. 32 j + 31 =
 *   the same channels of up); out [M, F] in out_dtype (ASQ_F16 / ASQ_BF16) = dt(dt(silu(y_g)) * y_u) with y = dt(s [* s_row[m]] * acc): bit-identical to
 *   asq_linear_w8a8 (gate), asq_linear_w8a8 (up) and the SiLU * up of asq_silu_mul_quantize with the same ASQ_SILU_FAST flag.  s_row: per-token activation scales or NULL.
 *   row_off / col_off: both NULL (plain operands) or the activation's row vector and asq_weight_offset_image(w_gu)'s column vector (offset operand images).
 * Runs on the persistent 256 x 256 kernel only: asq_gate_up_supported(M, F, K, out_dtype) = M % 256 == 0, F % 128 == 0, K % 256 == 0, K <= 65536, more than 256
 * tiles of 256 x 256 over [M, 2 F]; other shapes return ASQ_ERR_DIM (callers run the two linears + asq_silu_mul_quantize).  ASQ_GATE_UP=0 makes the query return 0. */
int asq_gate_up_supported(int64_t M, int64_t F, int64_t K, int out_dtype);
int asq_linear_w8a8_gate_up(const int8_t *xq, const int8_t *w_gu, void *out, int out_dtype, int64_t M, int64_t F, int64_t K,
                            float s_gate, float s_up, const float *s_row, int flags, const int32_t *row_off, const int32_t *col_off, void *stream);
/* The int8-out form for a per-tensor consumer (round 5, last session: reference models/llama.py:228-235 with fc2 per-tensor; its prologue linear.py:289-292): out_q int8 [M, F] =
 * asq_quantize_act(ASQ_ACT_DIV, quant_scale) of the [M, F] tensor asq_linear_w8a8_gate_up would have written in act_dtype (ASQ_F16 / ASQ_BF16: the dtype every intermediate is
 * rounded in) -- bit-identical to that composition and to asq_silu_mul_quantize(per-tensor) behind the two linears; the consumer's quantiser launch and the fp tensor disappear.
 * Same shapes as asq_linear_w8a8_gate_up (asq_gate_up_supported); quant_scale > 0. */
int asq_linear_w8a8_gate_up_q8(const int8_t *xq, const int8_t *w_gu, int8_t *out_q, int act_dtype, int64_t M, int64_t F, int64_t K, float s_gate, float s_up,
                               const float *s_row, int flags, float quant_scale, const int32_t *row_off, const int32_t *col_off, void *stream);
/* The grouped form (round 5, last session): Mixtral's experts (reference models/mixtral.py:99-101,142-145: w2(act(w1 x) * w3 x) per expert) -- w1 || w3 of ALL experts as ONE
 * grouped launch whose epilogue writes SiLU(w1 x) * (w3 x): asq_linear_w8a8_grouped[_off]'s launch (rows sorted by group, group_offsets[0 .. ngroups], device-side tile
 * scheduler, half tiles, in-launch K split of the tail round with a workspace) over w_gu int8 [ngroups, 2 F, K], each group's rows interleaved as above; s_gate / s_up
 * [ngroups] are the per-group dequant scales of the two projections; out [M, F].  Per-tensor activations (no row scales).  row_off / col_off: both NULL or the activation's
 * row vector and the column vector of asq_weight_offset_image over the [ngroups * 2 F, K] stack.  Bit-identical to asq_linear_w8a8_grouped (w1), (w3) and the SiLU * up of
 * asq_silu_mul_quantize with the same ASQ_SILU_FAST flag.  asq_grouped_gate_up_supported: F % 128 == 0, K % 128 == 0, K <= 65536, 2-byte out_dtype. */
int asq_grouped_gate_up_supported(int64_t M, int64_t F, int64_t K, int out_dtype);
int asq_linear_w8a8_grouped_gate_up(const int8_t *xq, const int8_t *w_gu, void *out, int out_dtype, const int32_t *group_offsets, int ngroups,
                                    int64_t M, int64_t F, int64_t K, const float *s_gate, const float *s_up, int flags,
                                    const int32_t *row_off, const int32_t *col_off, void *workspace, size_t workspace_bytes, void *stream);
int asq_silu_mul_quantize_off(const void *gate, const void *up, int x_dtype, int per_token, float quant_scale,
                              int8_t *xq_off, float *s_row, int32_t *row_off, int64_t M, int64_t K, void *stream);
int asq_linear_w8a8_forward_off(const void *x, int x_dtype, const int8_t *w, const int8_t *w_off, const int32_t *col_off, void *out,
                                int64_t M, int64_t N, int64_t K,
                                int act_mode, float quant_scale,
                                float s_scalar, const float *s_col, const float *bias,
                                void *workspace, size_t workspace_bytes, void *stream);

/* ---- FP8 (OCP e4m3fn) linear path: layers/functional/quantization.py:144-211 (quantisers),
 * layers/nn/linear.py:336-369 easy_fp8_gemm, :373-452 FP8LinearDynamic, :503-580 FP8LinearStatic.
 * fp8 tensors are raw bytes (torch.float8_e4m3fn storage).
 *
 * asq_quantize_act_fp8: x [M,K] -> xq e4m3fn [M,K] = cast(clamp(x / scale, +-448)), round-to-nearest-even.
 *   ASQ_FP8_PER_TOKEN : scale[m] = (rowabsmax / 448 in x_dtype) as f32, division in fp32; scale_out f32[M]
 *   ASQ_FP8_PER_TENSOR: scale = absmax / 448 in x_dtype, quotient rounded to x_dtype (dynamic per-tensor);
 *                       scale_out f32[2]: [0] receives the scale, [1] is scratch.  The scale stays on the
 *                       device (no host sync), exactly like the reference's 0-dim scale tensor.
 *   ASQ_FP8_STATIC    : scale = static_scale (calibrated input_scale / output_scale); scale_out may be NULL. */
#define ASQ_FP8_PER_TOKEN 0
#define ASQ_FP8_PER_TENSOR 1
#define ASQ_FP8_STATIC 2
int asq_quantize_act_fp8(const void *x, int x_dtype, int mode, float static_scale,
                         uint8_t *xq, float *scale_out, int64_t M, int64_t K, void *stream);

/* out[M,N] (out_dtype) = (xq . w^T as fp32, fp8 matrix cores) * (a_scale * w_scale) (+ bias[n])
 * a_scale: a_scale_dev (device f32; [M] if a_per_token else [1]) or, when NULL, a_scale_host.
 * The reference dequantises both operands and calls F.linear (native_fp8_support = False,
 * linear.py:342,364-368): results agree to fp32 rounding, not bit-for-bit. */
#define ASQ_FP8_E4M3 0
#define ASQ_FP8_E5M2 1
int asq_linear_fp8(const uint8_t *xq, const uint8_t *w, int fp8_format, void *out, int out_dtype,
                   int64_t M, int64_t N, int64_t K,
                   const float *a_scale_dev, int a_per_token, float a_scale_host, float w_scale,
                   const float *bias, void *stream);

/* Grouped fp8 launch (Mixtral experts with FP8LinearDynamic math): ngroups independent e4m3 linears in one
 * kernel launch, the fp8 twin of asq_linear_w8a8_grouped.
 *   xq e4m3 [M,K] rows sorted by group, group_offsets DEVICE int32 [ngroups+1], w e4m3 [ngroups][N][K],
 *   a_scale f32 [M] (per-token, DEVICE), w_scale_group f32 [ngroups] (DEVICE), bias f32 [ngroups][N] | NULL
 * out[m,n] = acc_f32 * (a_scale[m] * w_scale_group[g(m)]) (+ bias[g(m)][n]).  K % 128 == 0, 16-B aligned operands. */
int asq_linear_fp8_grouped(const uint8_t *xq, const uint8_t *w, void *out, int out_dtype, const int32_t *group_offsets, int ngroups,
                           int64_t M, int64_t N, int64_t K, const float *a_scale, const float *w_scale_group, const float *bias, void *stream);

/* SiLU(gate) * up fused with the per-token e4m3 quantiser of the FP8 linear that consumes it (round 6): the reference's gated MLP on FP8LinearDynamic modules --
 * models/mixtral.py:99-101 `w2(act_fn(w1(x)) * w3(x))`, w2's prologue per_token_quantize_fp8 (layers/functional/quantization.py:173-191, called from
 * layers/nn/linear.py:413-427).  gate, up [M,K] of x_dtype (K a multiple of 8 / 4 (fp32), <= 16384 / 8192); xq [M,K] e4m3fn bytes; scale f32 [M] =
 * f32(x_dtype(max_k |a| / 448)) with a = x_dtype(x_dtype(silu(gate)) * up); q = e4m3(clamp(f32(a) / scale, +-448)).  flags: 0 = the fixed-operation-order SiLU
 * (bit-identical to oracle/n1.py::silu_mul_quant_fp8_kernel_order), ASQ_SILU_FAST = the hardware transcendentals.  Replaces three passes (silu, mul, quantiser). */
int asq_silu_mul_quantize_fp8(const void *gate, const void *up, int x_dtype, int flags, uint8_t *xq, float *scale, int64_t M, int64_t K, void *stream);

/* Rotary position embedding of a q / k projection's output in ONE pass (round 6; caller-side glue, not a reference-path function: the reference's model wrappers run
 * HF's apply_rotary_pos_emb between their W8A8 q / k linears and the attention product -- models/llama.py:111, models/mixtral.py:75).  x [B, S, H, D] with H * D
 * contiguous and x_row_pitch elements between consecutive (b, s) rows (0 = dense H * D; a slice of a fused q || k || v output passes the fused width), out [B, S, H, D]
 * dense, cos_tab / sin_tab [S, D/2] of x_dtype, positions 0 .. S-1, "rotate_half" convention:
 *   out[.., :D/2] = dt(f32(dt(x1 * cos)) - x2 * sin),  out[.., D/2:] = dt(f32(dt(x2 * cos)) + x1 * sin)
 * -- fp16: bit-identical to torch.addcmul(x1 * cos, x2, sin, value=-1) / torch.addcmul(x2 * cos, x1, sin) on this platform (the sum is rounded ONCE to fp16).
 * D % 16 == 0 (fp32: % 8); out may alias a dense x. */
int asq_rope(const void *x, int64_t x_row_pitch, void *out, int x_dtype, const void *cos_tab, const void *sin_tab, int64_t B, int64_t S, int64_t H, int64_t D, void *stream);

/* FP8LinearDynamic experts' w1 || w3 as ONE grouped launch whose epilogue writes SiLU(w1 x) * (w3 x) (round 6; reference models/mixtral.py:99-101,142-145 with every expert
 * linear an FP8LinearDynamic, layers/nn/linear.py:413-427, easy_fp8_gemm :336-369): the fp8 counterpart of asq_linear_w8a8_grouped_gate_up.
 *   xq e4m3fn [M, K] rows sorted by group + a_scale f32 [M] (asq_quantize_act_fp8 per-token); w_gu e4m3fn [ngroups][2 F][K]: every group's w1 and w3 ROW-INTERLEAVED in
 *   blocks of 32 channels (rows 64 j .. 64 j + 31 = w1 channels 32 j .. + 31, rows 64 j + 32 .. 64 j + 63 = the same channels of w3); s_gate / s_up f32 [ngroups]: the two
 *   stacks' per-expert weight scales; out [M, F] (ASQ_F16 / ASQ_BF16) = dt(dt(silu(y_1)) * y_3) with y = dt(acc * (a_scale[m] * w_scale)) -- bit-identical to
 *   asq_linear_fp8_grouped (w1), (w3) and the SiLU * up of asq_silu_mul_quantize_fp8 with the same flags.  F % 128 == 0, K % 128 == 0. */
int asq_fp8_grouped_gate_up_supported(int64_t M, int64_t F, int64_t K, int out_dtype);
int asq_linear_fp8_grouped_gate_up(const uint8_t *xq, const uint8_t *w_gu, void *out, int out_dtype, const int32_t *group_offsets, int ngroups, int64_t M, int64_t F, int64_t K,
                                   const float *a_scale, const float *s_gate, const float *s_up, int flags, void *stream);

/* Caller-side glue for the reference's MODULE composition (round 6; not reference-path functions): the modules that sit between the W8A8 linears there -- the folded
 * RMSNorm (models/llama.py:27-37, HF LlamaRMSNorm's arithmetic) and the gated activation act_fn(gate) * up (models/llama.py:206-211) -- each as ONE pass with a floating
 * output that the next linear quantises itself; the fused (N1) forms above emit int8 instead.
 *   asq_rmsnorm:  y = dt(weight * dt(f32(x) * rsqrt(mean(f32(x)^2) + eps)));  x, y [M,K], weight [K] of x_dtype; K % 8 == 0 (fp32: % 4), K <= 16384 (8192).
 *   asq_silu_mul: out = dt(dt(silu(gate)) * up) over n elements (n % 8 == 0; fp32: % 4); flags: 0 = fixed-operation-order SiLU (oracle/n1.py), ASQ_SILU_FAST. */
int asq_rmsnorm(const void *x, int x_dtype, const void *weight, float eps, void *y, int64_t M, int64_t K, void *stream);
int asq_silu_mul(const void *gate, const void *up, int x_dtype, int flags, void *out, int64_t n, void *stream);

/* FP8E5M2Linear (linear.py:583-644): plain unscaled cast x -> e5m2 (round-to-nearest-even, IEEE-like
 * overflow to inf); the product then runs through asq_linear_fp8(..., ASQ_FP8_E5M2, ...) with unit scales.
 * (The reference's forward calls torch._scaled_mm with scale_a=None, which current PyTorch rejects;
 * restated from its intent: y = e5m2(x) . e5m2(W)^T + bias.) */
int asq_cast_e5m2(const void *x, int x_dtype, uint8_t *xq, int64_t n, void *stream);

/* ---- MX (OCP Microscaling, MXFP8 E4M3) with real block scales: an accuracy-checked opt-in beyond the reference (the reference's fp8
 * classes scale per tensor or per token, layers/nn/linear.py:336-644).  32 consecutive k share one E8M0 scale byte (value 2^(byte - 127), 0xFF = NaN):
 * scale = 2^(floor(log2 max|x|) - 8), elements = e4m3fn(x / scale) saturating at +-448 (OCP MX v1.0 section 6.3).
 * asq_quantize_mxfp8: x [M,K] f32/f16/bf16 -> xq [M,K] e4m3 bytes + scales [M,K/32]; K % 32 == 0, x / xq 16-byte aligned.
 * asq_linear_mxfp8: out[M,N] = sum_blocks (xq . wq) * 2^(x_scale + w_scale - 254) (+ bias f32 [N]) on v_mfma_scale_f32_32x32x64_f8f6f4 with the scale
 * bytes as matrix-core operands; K % 64 == 0 (the 128 x 128 tiled kernel for K % 512 == 0 and 16-byte aligned scales, a plain kernel otherwise).
 * Nothing dispatches to it: on SmoothQuant-like activations per-token e4m3 (asq_quantize_act_fp8 + asq_linear_fp8) is the more accurate AND the
 * faster path. */
int asq_quantize_mxfp8(const void *x, int x_dtype, uint8_t *xq, uint8_t *scales, int64_t M, int64_t K, void *stream);
int asq_linear_mxfp8(const uint8_t *xq, const uint8_t *x_scales, const uint8_t *wq, const uint8_t *w_scales, void *out, int out_dtype,
                     int64_t M, int64_t N, int64_t K, const float *bias, void *stream);

/* ---- introspection for tests / bench: which GEMM kernel the dispatcher picks for a shape (aligned operands):
 * "skinny" (weight streaming), "p8q" (128x128x128 tiles), "p8h" (128x256x128 tiles), "p16" (256x256x128 tiles, 8 waves, on v_mfma_i32_16x16x64_i8;
 * launches it does not carry -- 4-byte / int8 outputs, K splits -- run the same tile on gemm_i8_p8 in its 16x16x64 mode) or "generic";
 * "p16+tail" / "p8h+tail" when the call runs as a main launch + a column remainder of 128 x 128 tiles.
 * Development overrides (read once per process): ASQ_GEMM_KERNEL=generic|skinny|p8q|p8h|p16|p8|p4|p4x16 (p8 / p4: the 32x32x32-instruction kernels of
 * rounds 1-2, p4x16: four waves on the 16x16x64 instruction), ASQ_MMA=32 (p8h / p8q / grouped launches on the 32x32x32 instruction), ASQ_KSPLIT=n, ASQ_SPLITK_FIX=0 (K splits of the 128 x 128 kernel as slab launch + reduce launch again),
 * ASQ_SK_NT=1|2, ASQ_NO_TAIL=1. */
const char *asq_gemm_kernel_name(int64_t M, int64_t N, int64_t K);

#ifdef __cplusplus
}
#endif
#endif /* ASQ_HIP_H */

/*
 * petr_hip.h — C ABI of libpetr_hip.so: the MI355X (gfx950) implementation of the PETRHead hot path.
 *
 * The reference (sty61010/PETR) is pure Python and has no FFI for this path; what it calls are
 * PyTorch operators.  Each entry point below therefore names the reference call site whose
 * arithmetic it replaces (paths relative to projects/mmdet3d_plugin/ of the reference) — a
 * maintainer binds them with ctypes from the reference's own modules.
 *
 * Conventions
 *  - plain C: raw DEVICE pointers + explicit sizes/strides (in elements), no torch types;
 *  - every function enqueues work on `stream` (a hipStream_t passed as void*) and returns
 *    immediately; it never synchronises the device, never allocates, keeps no mutable global
 *    state; the caller owns all buffers including workspaces (sizes via *_workspace_bytes);
 *  - return value: 0 on success, negative petr_status on error; text via petr_last_error()
 *    (thread-local);
 *  - all tensors are fp32 unless stated; masks are uint8 (non-zero = ignore, as
 *    key_padding_mask in models/utils/petr_transformer.py:318).
 */
#ifndef PETR_HIP_H_
#define PETR_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PETR_HIP_VERSION 100 /* 0.1.0 */

typedef enum {
  PETR_OK = 0,
  PETR_ERR_INVALID = -1,   /* bad argument (shape, alignment, null pointer)      */
  PETR_ERR_UNSUPPORTED = -2, /* configuration outside what the kernels implement */
  PETR_ERR_LAUNCH = -3,    /* hipLaunch / runtime failure                        */
  PETR_ERR_WORKSPACE = -4  /* workspace too small                                */
} petr_status;

int petr_version(void);
const char* petr_last_error(void);
/* multiProcessorCount, gcnArchName ("gfx950...") of the current device; arch buffer >= 64 bytes */
int petr_device_caps(int* num_cu, char* arch, int arch_len);

/* ------------------------------------------------------------------------------------------
 * K1  camera-frustum -> LiDAR coordinate volume
 *     replaces models/dense_heads/petr_head.py:290-331 (PETRHead.position_embeding up to and
 *     including inverse_sigmoid; twin at petrv2_head.py:348-394).
 *     out    [B*N, 3*D, H, W]  channel c = 3*d + axis  (petr_head.py:330)
 *     cmask  [B, N, H, W] uint8 or NULL: (#coords outside [0,1] over D*3) > D/2  (petr_head.py:327-328),
 *            NOT yet OR-ed with the padding mask
 *     img2lidar [B*N, 16] fp32 row-major = float32(inv_fp64(lidar2img)) computed by the host as
 *            the reference does (petr_head.py:308-315)
 *     depth  [D] fp32 depth bins (petr_head.py:293-301), computed by the host
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* img2lidar;
  const float* depth;
  float* out;
  uint8_t* cmask;
  int B, N, H, W, D;
  float pad_h, pad_w;        /* img_metas[0]['pad_shape'][0][:2] */
  float range[6];            /* position_range */
  float eps;                 /* 1e-5 (petr_head.py:287) */
} petr_coords3d_args;
int petr_coords3d_fwd(const petr_coords3d_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3a SinePositionalEncoding3D.forward  (models/utils/positional_encoding.py:58-100)
 *     mask [B,N,H,W] uint8 -> out [B,N,3*F,H,W]; dim_t [F] fp32 table (positional_encoding.py:82-84)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const uint8_t* mask;
  const float* dim_t;
  float* out;
  int B, N, H, W, F;
  int normalize;
  float scale, eps, offset;
} petr_sine3d_args;
int petr_sine3d_fwd(const petr_sine3d_args* a, void* stream);

/* K3b pos2posemb3d (models/dense_heads/petr_head.py:31-43): pos [n,3] -> out [n,3*F], order (y,x,z).
 *     bwd: dpos [n,3] = d out / d pos contracted with dout [n,3*F]                         */
int petr_posemb3d_fwd(const float* pos, const float* dim_t, float* out, int n, int F, void* stream);
int petr_posemb3d_bwd(const float* pos, const float* dim_t, const float* dout, float* dpos, int n, int F,
                      void* stream);

/* ------------------------------------------------------------------------------------------
 * Dropout (training mode; reference: attn_drop 0.1 inside nn.MultiheadAttention, the residual
 * dropout_layer of both attentions models/utils/petr_transformer.py:258-265,367, and mmcv FFN's
 * ffn_drop after each of its two Linear layers).
 *     keep(row, col) = hash16(seed, site, row, col) >= round(p * 2^16) ; kept values are scaled by the matching
 *     1/(1-p) (p = 0.1 is realised as 6554/65536 = 0.100006).
 * Counter-based and stateless: a backward call given the same (seed, site) regenerates the forward's
 * mask, nothing is stored.  `site` separates the dropout layers of one forward (the executor uses
 * 8*layer + {0 self-attn P, 1 self-attn out, 2 cross-attn P, 3 cross-attn out, 4 FFN hidden, 5 FFN out}).
 * The masks cannot equal torch's Philox stream; parity tests export them (petr_dropout_mask) and hand
 * them to the oracle.  p == 0 disables the layer.                                              */
typedef struct { uint64_t seed; uint32_t site; float p; } petr_dropout;
/* keep[row*cols + col] = 1 / 0 for the element a kernel addresses as (row, col): attention P uses
 * row = (b*H + h)*Q + q, col = key; row-major activations use row = m, col = n.               */
int petr_dropout_mask(const petr_dropout* d, long rows, long cols, uint8_t* keep, void* stream);
/* The same mask packed for ONE attention call (row = bh*Q + q, col = key; BH = B*H):
 *   bits_q[(bh * nkb + kb) * 32 nqt + q]   bit j = keep(q, 32 kb + j)   query-major
 *   bits_k[(bh * nqt + qt) * 32 nkb + 32 kb + slot(c)]   bit i = keep(32 qt + i, 32 kb + c)   key-major, the 32 words of a
 *                                          key block in the order the forward's comparisons produce them:
 *                                          slot(c) = 2 ((c & 3) + 4 (c >> 3)) + ((c >> 2) & 1).  This is what petr_mha_fwd*
 *                                          leaves in its drop_bits (scalar stores of the comparison masks, no vector work)
 *                                          and petr_mha_bwd* reads (the key sits on the lane there), so that the backward
 *                                          tests a bit instead of re-hashing (~20 % of its time at 24 000 keys)
 * nqt = ceil(Q/32), nkb = ceil(L/32); each layout holds petr_dropout_bits_words(BH, Q, L) uint32_t; either may be NULL.
 * (Generating the masks with this kernel ahead of the attention calls, so that the forward tests bits too, was measured:
 * it costs what it saves - the generator is as much VALU work as the hashing it replaces.) */
size_t petr_dropout_bits_words(int BH, int Q, int L);
int petr_dropout_bits(const petr_dropout* d, int BH, int Q, int L, uint32_t* bits_q, uint32_t* bits_k, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense contraction  C[z][m, n] (+)= act( alpha * sum_k A(m,k) * B(n,k) + bias[n] + R[m,n] )
 *     replaces every nn.Conv2d(1x1) / nn.Linear / F.linear on the path:
 *     petr_head.py:220-274 (input_proj, position_encoder, adapt_pos3d, query_embedding, branches),
 *     torch.nn.MultiheadAttention in-/out-projections (petr_transformer.py:271,357-362), mmcv FFN,
 *     and their gradients (the same contraction with transposed operand layouts).
 *     fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32 products).
 *
 *     A(m,k) = a_kcontig ? a[m*lda + k] : a[k*lda + m];   B(n,k) likewise with b/ldb/b_kcontig.
 *     a2: optional addend to A with A's layout (query + query_pos, key + key_pos:
 *         petr_transformer.py:341-344); a2 row = m % a2_rows if a2_rows>0; applied only for output
 *         columns n < a2_ncols if a2_ncols>0.
 *     Two batch dimensions z = z0*nb1 + z1 with element strides for every operand.
 *     k_seg > 0: the K axis is cut into K/k_seg segments at different base addresses,
 *         k = s*k_seg + kk  ->  A(m,kk) at a + s*a_seg_stride, B(n,kk) at b + s*b_seg_stride
 *         (sums over (layer, channel), (view, pixel), (batch, token) in the gradients).
 *     C store: c[(n / c_nblk) * c_nblk_stride + m*ldc + (n % c_nblk)] if c_nblk>0 (head-split
 *         K/V layout), else c[m*ldc + n].
 *     flags: PETR_GEMM_RELU, PETR_GEMM_ACCUMULATE (C += ...), PETR_GEMM_RELU_MASK (multiply the
 *         result by (relu_mask[m,n] > 0); relu_mask indexed like R) , PETR_GEMM_SIGMOID_MUL
 *         (C = mul[m,n] * sigmoid(value); mul indexed like R; PETRv2 SELayer petrv2_head.py:48-60).
 *     split_k > 1: slice z2 of K writes its partial to c + z2*c_split_stride (bias/residual are NOT
 *         applied; reduce with petr_layernorm_fwd or petr_reduce_partials).  With PETR_GEMM_ATOMIC
 *         every slice instead ADDS into c with float atomics (weight gradients, c += ; low-order
 *         bits depend on arrival order).
 *     a_colsum: optional [M] vector receiving (atomically, +=) sum_k A(m,k): the bias gradient that
 *         belongs to a weight-gradient contraction dW = dC^T X (A = dC^T), at no extra launch.
 * ------------------------------------------------------------------------------------------ */
enum { PETR_GEMM_RELU = 1, PETR_GEMM_ACCUMULATE = 2, PETR_GEMM_RELU_MASK = 4, PETR_GEMM_SIGMOID_MUL = 8,
       PETR_GEMM_ATOMIC = 16, PETR_GEMM_STORE_BF16 = 32, PETR_GEMM_BF16 = 64, PETR_GEMM_BIAS_M = 128,
       PETR_GEMM_A_BF16 = 256, PETR_GEMM_B_BF16 = 512, PETR_GEMM_R_BF16 = 1024 };
/* PETR_GEMM_A_BF16 / _B_BF16 / _R_BF16 (with PETR_GEMM_BF16): the operand `a` / `b` / the residual-or-mask operand `r` is
 * ALREADY bf16 in memory (raw uint16_t bits behind the float pointer; ld / batch / segment strides count bf16 elements;
 * K-contiguous rows need K % 8 == 0): activations that a producing epilogue stored with PETR_GEMM_STORE_BF16, or the
 * bf16 dK / dV of petr_mha_bwd_bf16.  Half the operand bytes, no rounding pass; numerically identical to rounding the fp32
 * value on load.  a_colsum of a bf16 A sums the bf16 values. */
/* PETR_GEMM_BIAS_M: bias is indexed by the output ROW m (bias[m]) instead of the column: a convolution written with the
 * weights as the A operand, so that its result lands channel-major (NCHW) - the CPFPN neck boundary (petr_fpn.h part
 * below).  fp32 tiled kernel only. */
/* PETR_GEMM_STORE_BF16: `c` points to bf16 storage (uint16_t bits, round to nearest even) and ldc / c_bs0 / c_bs1 /
 * c_nblk_stride count bf16 elements: the K/V projections feeding petr_mha_fwd_bf16.  Tiled kernel only; excludes
 * ACCUMULATE, ATOMIC and split_k > 1.
 * PETR_GEMM_BF16: the fp32 operands are rounded to bf16 on load and multiplied on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation (what torch.autocast(bfloat16) does to an nn.Linear / 1x1 conv); B K-contiguous, A K-contiguous or (for
 * K % 32 == 0 and >= 64 output tiles) K-major, 16-byte aligned, K % 16 == 0, epilogue limited to bias / residual / ReLU / STORE_BF16 (a2 addend allowed). */
/* drop.p > 0 (only without batch dims / split_k): the activated value act(...) is dropped out with
 * (row, col) = (m, n) before it is stored (mmcv FFN: Linear, ReLU, Dropout).                   */
typedef struct {
  const float* a; long lda; int a_kcontig; long a_bs0, a_bs1;
  const float* a2; int a2_rows; int a2_ncols;
  const float* b; long ldb; int b_kcontig; long b_bs0, b_bs1;
  float* c; long ldc; long c_bs0, c_bs1; int c_nblk; long c_nblk_stride;
  const float* bias; long bias_bs0, bias_bs1;
  const float* r; long ldr; long r_bs0, r_bs1;      /* residual / relu-mask / mul operand */
  int M, N, K, nb0, nb1;
  int split_k; long c_split_stride;
  int k_seg; long a_seg_stride, b_seg_stride;
  float* a_colsum; long cs_bs0, cs_bs1;
  int flags;
  float alpha;
  petr_dropout drop;
} petr_gemm_args;
int petr_gemm(const petr_gemm_args* g, void* stream);

/* column sums: out[n] (+)= sum_m x[m*ld + n]  (bias gradients). accumulate: 0 overwrite, 1 add.
 * ws: petr_colsum_workspace_bytes(N) bytes of scratch (deterministic two-stage sum, no atomics) */
size_t petr_colsum_workspace_bytes(int N);
int petr_colsum(const float* x, long ld, int M, int N, float* out, int accumulate, void* ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (C <= 1024, multiple of 4), eps as given (nn.LayerNorm, 1e-5):
 *     z[m,:] = sum_{p<n_partials} x[p*partial_stride + m*C ..] + bias + residual[m,:]
 *     y[m,:] = (z - mean)/sqrt(var+eps) * gamma + beta ; optional ReLU; optional nan_to_num on y
 *     replaces norms.{0,1,2} / post_norm (petr_transformer.py:418-419,444; mmcv layer 'norm' op),
 *     the LayerNorms of cls_branches (petr_head.py:229) and torch.nan_to_num (petr_head.py:435).
 *     z_out / mean / rstd may be NULL (inference).
 * ------------------------------------------------------------------------------------------ */
enum { PETR_LN_RELU = 1, PETR_LN_NAN_TO_NUM = 2 };
typedef struct {
  const float* x; int n_partials; long partial_stride;
  const float* bias; const float* residual;
  const float* gamma; const float* beta;
  float* y; float* z_out; float* mean; float* rstd;
  int M, C; float eps; int flags;
  /* optional second output y2[m,:] = y[m,:] + add2[m % add2_rows,:]  (query + query_pos for the next
   * attention, petr_transformer.py:341-342, emitted while the row is still in registers) */
  float* y2; const float* add2; int add2_rows;
  /* dropout of the non-residual part: z = drop(sum partials + bias) + residual (p == 0: off)  */
  petr_dropout drop;
} petr_layernorm_args;
int petr_layernorm_fwd(const petr_layernorm_args* a, void* stream);
/* dz = LN backward given z (pre-norm input), mean, rstd, gamma, dy (and y if PETR_LN_RELU was used:
 * dy is masked by y>0). dgamma/dbeta [C] accumulated (+=) via per-block partials in `ws`
 * (petr_layernorm_bwd_workspace_bytes); ws == NULL: float atomics instead (one launch).
 * dz may alias dy. dz_accumulate: dz += result.                                            */
typedef struct {
  const float* z; const float* mean; const float* rstd; const float* gamma;
  const float* dy; const float* y;
  float* dz; float* dgamma; float* dbeta;
  void* ws;
  int M, C; int flags; int dz_accumulate;
  /* upstream gradient = sum_{p<dy_partials} dy[p*dy_partial_stride + ...] + dy_residual (both optional):
   * lets a split-K input-gradient contraction hand its slabs over without a reduce launch */
  int dy_partials; long dy_partial_stride; const float* dy_residual;
  /* optional second output dz_drop = dz * keep / (1-p): the gradient of the dropped branch of
   * z = drop(f) + residual (dz itself is the residual's gradient)                              */
  float* dz_drop; petr_dropout drop;
} petr_layernorm_bwd_args;
size_t petr_layernorm_bwd_workspace_bytes(int M, int C);
int petr_layernorm_bwd(const petr_layernorm_bwd_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-head attention core  O = softmax(scale * Q K^T + mask) V,  head_dim = 32
 *     replaces the scaled-dot-product inside torch.nn.MultiheadAttention called at
 *     models/utils/petr_transformer.py:357-362 (cross-attention, PETRMultiheadAttention) and by
 *     mmcv MultiheadAttention (self-attention).  Flash-style: scores are never written; fp32 MFMA
 *     (v_mfma_f32_32x32x2_f32) for QK^T and PV, fp32 online softmax in registers.
 *     q[b,h,i,:] at q + b*q_bs + h*q_hs + i*q_rs (+d);  k, v, o likewise.  lse [B,H,Q] (natural log)
 *     or NULL.  kpm [B,L] uint8 (non-zero = ignore) or NULL.  A fully masked row yields NaN, as
 *     the reference does (cleared later by nan_to_num).
 *     n_split>1 splits L over workgroups; partials go to `ws` (petr_mha_fwd_workspace_bytes) and a
 *     second kernel merges them.  n_split=0 lets the library choose.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* q; long q_bs, q_hs, q_rs;
  const float* k; long k_bs, k_hs, k_rs;
  const float* v; long v_bs, v_hs, v_rs;
  float* o; long o_bs, o_hs, o_rs;
  float* lse;
  const uint8_t* kpm;
  int B, H, Q, L;
  float scale;
  int n_split;
  void* ws; size_t ws_bytes;
  petr_dropout drop;   /* dropout of the attention probabilities (row = (b*H+h)*Q+q, col = key) */
  int* sched;   /* optional: B*H*ceil(Q/128) ints, ZERO on entry and left zero on exit; enables dynamic K/V-tile
                 * scheduling between the n_split workers of a query block (NULL: equal static ranges) */
  uint32_t* drop_bits;   /* out, optional with drop.p > 0: petr_dropout_bits_words(B*H, Q, L) words; the kernel leaves the
                          * dropout mask it applied there, packed key-major (the bits_k layout of petr_dropout_bits), for
                          * petr_mha_bwd's drop_bits: the backward then tests one bit per probability instead of hashing it */
  int defer_merge;       /* 1 (needs an explicit n_split and ws): with n_split > 1 the L-split partials stay in ws -
                          * o_part [n_split][B*H][Q][32] floats, then ml_part [n_split][B*H][Q][2] (running maximum in
                          * log2 units, sum) - and o / lse are NOT written: the consumer merges them
                          * (petr_attn_out_ln does, in its prologue).  n_split == 1 writes o / lse as always. */
} petr_mha_fwd_args;
size_t petr_mha_fwd_workspace_bytes(int B, int H, int Q, int L, int n_split);
int petr_mha_choose_split(int B, int H, int Q, int L);
int petr_mha_fwd(const petr_mha_fwd_args* a, void* stream);

/* bf16 K/V variant of the attention core (BASELINE.json configs 3-5: bf16 I/O, fp32 accumulate and softmax):
 *     k, v are bf16 (raw uint16_t bits) with strides in bf16 ELEMENTS, rows 16-byte aligned; q, o, lse stay fp32
 *     (q is rounded to bf16 after the scale*log2(e) pre-multiplication, P is rounded to bf16 for the second
 *     product; scores, softmax statistics and the output accumulate in fp32).  Same call site as petr_mha_fwd
 *     (petr_transformer.py:357-362).  `sched` is ignored (static key ranges).  Everything else as petr_mha_fwd. */
typedef struct {
  const float* q; long q_bs, q_hs, q_rs;
  const uint16_t* k; long k_bs, k_hs, k_rs;
  const uint16_t* v; long v_bs, v_hs, v_rs;
  float* o; long o_bs, o_hs, o_rs;
  float* lse;
  const uint8_t* kpm;
  int B, H, Q, L;
  float scale;
  int n_split;
  void* ws; size_t ws_bytes;
  petr_dropout drop;
  int* sched;
  uint32_t* drop_bits;   /* as petr_mha_fwd_args.drop_bits */
  int defer_merge;       /* as petr_mha_fwd_args.defer_merge */
} petr_mha_fwd_bf16_args;
size_t petr_mha_fwd_bf16_workspace_bytes(int B, int H, int Q, int L, int n_split);
int petr_mha_fwd_bf16_choose_split(int B, int H, int Q, int L);   /* what n_split = 0 selects */
int petr_mha_fwd_bf16(const petr_mha_fwd_bf16_args* a, void* stream);
/* ------------------------------------------------------------------------------------------
 * Attention output block of a decoder layer in ONE launch (C = 256 = 8 heads x 32):
 *     ao = merge of the attention's L-split partials (or the attention output itself)
 *     z  = drop(ao W^T + bias) + residual ,  y = LayerNorm(z) * gamma + beta ,  y2 = y + add2 (optional),
 *     out2 = y2 W2^T + bias2 (optional: the next attention's query projection)
 *   replaces, per attention of petr_transformer.py:357-367 / mmcv MultiheadAttention + the following norm of
 *   multi_atten_decoder_layer.py:204-293, FOUR launches of the 900-row chain: the partial merge, the out-projection
 *   contraction (+ residual), and the LayerNorm (+ dropout, + query_pos add).  One workgroup owns 16 full rows: merge
 *   -> LDS, 16 x 256 x 256 on v_mfma_f32_16x16x4_f32 with the weight streamed from L2, row statistics across the eight waves.
 *   The kernel reads the weights TRANSPOSED (wT [in = 256][out], k-major: the 16 lanes of a load share one 128-byte line;
 *   petr_head_fwd keeps transposed copies of the decoder weights in its workspace).
 *   a: [M, 256] attention output when n_split <= 1; else it is WRITTEN with the merged output (the backward needs it) from
 *   o_part / ml_part (layout: petr_mha_fwd_args.defer_merge), lse [B*H*Q] is written too, attn_scale = the attention
 *   dropout's 1/(1-p) (1 without).  rows m = b*Q + q.  z, mean, rstd are written when non-NULL (training needs them).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  float* a;                               /* [M, 256]: in (n_split <= 1) or out (merged attention output) */
  const float* o_part; const float* ml_part; int n_split; int B, H, Q; float attn_scale; float* lse;
  const float* wT; const float* bias;     /* [256 in, 256 out] = the nn.Linear weight transposed, [256] */
  const float* residual;                  /* [M, 256] */
  petr_dropout drop;                      /* of the projection's output, before the residual is added */
  const float* gamma; const float* beta; float eps;
  float* z; float* mean; float* rstd;     /* optional outputs: pre-norm sum [M, 256], row statistics [M] */
  float* y;                               /* [M, 256] */
  float* y2; const float* add2; int add2_rows;   /* optional: y2 = y + add2[row % add2_rows] */
  int M;
  const float* w2T; const float* bias2; float* out2;  /* optional second projection of the same rows: out2 = (y2 if
                                                       * requested, else y) w2^T + bias2, [M, 256]; w2T = w2 transposed */
} petr_attn_out_ln_args;
int petr_attn_out_ln(const petr_attn_out_ln_args* a, void* stream);

/* The LayerNorm that closes a decoder layer AND the projections that read it, one launch (C = 256):
 *     z = drop(sum_p x[p] + bias) + residual,  y = LN(z) gamma + beta,  y2 = y + add2[row % add2_rows]
 *     out2[:, 256 j .. + 255] = (j < n2_pos ? y2 : y) w2[256 j .. + 255, :]^T + bias2,   j = 0 .. n2-1,  out2 [M, 256 n2]
 *   = petr_layernorm_fwd (same prologue: split-K slabs of the FFN's second contraction, bias, dropout, residual) + the next
 *   layer's self-attention in-projection (multi_atten_decoder_layer.py:223-237: q, k from x + query_pos -> n2_pos = 2,
 *   v from x; w2 = in_proj_weight [768, 256], n2 = 3).  The kernel reads w2 TRANSPOSED: w2T [256, 256 n2] (k-major). */
typedef struct {
  const float* x; int n_partials; long partial_stride;   /* [P][M, 256] */
  const float* bias; const float* residual; petr_dropout drop;
  const float* gamma; const float* beta; float eps;
  float* z; float* mean; float* rstd; float* y;
  float* y2; const float* add2; int add2_rows;
  int M;
  const float* w2T; const float* bias2; float* out2; int n2, n2_pos;
  uint16_t* out2_bf16;                    /* optional (new LAST field): a bf16 copy of out2, same [M, 256 n2] indexing - bf16 mode's
                                           * self-attention reads its K / V rows from it (no cast pass between the two launches) */
} petr_ln_proj_args;
int petr_ln_proj(const petr_ln_proj_args* a, void* stream);

/* LayerNorm backward AND the input gradient of the linear layer behind the normalised sum's sub-layer branch, one launch
 * (C = 256): petr_layernorm_bwd semantics for dz / dz_drop / dgamma / dbeta (float atomics), then
 *     out = mask( alpha * (dz_drop if dropout else dz) w ),   out [M, 256 n2]
 *   w [256, 256 n2] = the nn.Linear weight itself (dx = dy W: already k-major for this product), relu_mask [M, 256 n2]
 *   optional (out is zeroed where relu_mask <= 0: the FFN hidden).  Replaces layernorm_bwd + the out-projection / FFN2 input
 *   gradient of multi_atten_decoder_layer.py:204-293's backward. */
typedef struct {
  const float* z; const float* mean; const float* rstd; const float* gamma;
  const float* dy; int dy_partials; long dy_partial_stride; const float* dy_residual;
  float* dz; float* dz_drop; petr_dropout drop; float* dgamma; float* dbeta;
  int M;
  const float* w; int n2; float alpha; const float* relu_mask; float* out;
  const float* pre_a; const float* pre_w;    /* optional leading product: the upstream gradient is
                                              * pre_a [M, 256 pre_n] pre_w[256 pre_n, 256] + sum of the dy slabs (dy_partials
                                              * may be 0) + dy_residual - the input gradient of the projection that read the
                                              * normalised rows (pre_w = that nn.Linear weight) */
  int pre_n;                                 /* its depth in blocks of 256 (0 = 1; at most 3: a self-attention in_proj).  With
                                              * a leading product n2 may be 0: LayerNorm backward only (w, out unused) */
} petr_ln_bwd_proj_args;
int petr_ln_bwd_proj(const petr_ln_bwd_proj_args* a, void* stream);

/* Both contractions of the decoder FFN in one launch (C = 256 channels, fp32):
 *     hidden = drop(relu(x w1^T + b1))           [M, F]   written when `hidden` is non-null (the backward reads it)
 *     part[s] = hidden[:, slice s] w2[:, slice s]^T   [M, 256], s = 0 .. n_split-1, slice s = F / n_split hidden units
 *   w1 [F, 256], w2 [256, F] are the nn.Linear weights of mmcv's FFN (reference: multi_atten_decoder_layer.py:204-293 runs
 *   `ffn` between the cross-attention's norm and the closing norm); the kernel reads their TRANSPOSES
 *   w1t [256, F] and w2t [F, 256] (k-major: a 16-lane group of a load shares one 128-byte line).  The output bias, the
 *   output dropout, the residual and the sum over the slabs belong to the normalisation that follows (petr_ln_proj /
 *   petr_layernorm_fwd with n_partials = n_split), exactly as after the split-K contraction this replaces.
 *   n_split in {1, 2, 4, 8}, F a multiple of 256 n_split; dropout element index = (row, hidden unit) as in petr_gemm. */
typedef struct {
  const float* x; const float* w1t; const float* b1; const float* w2t;
  float* hidden; float* part; long part_stride;
  petr_dropout drop;
  int M, F, n_split;
} petr_ffn_fwd_args;
int petr_ffn_fwd(const petr_ffn_fwd_args* a, void* stream);

/* The FFN's input gradient in one launch (the backward of petr_ffn_fwd's two products):
 *     d_hidden = alpha * (dy w2) where hidden > 0, else 0      [M, F]   (written: the first layer's weight gradient reads it)
 *     part[s]  = d_hidden[:, slice s] w1[slice s, :]           [M, 256], summed by the consumer (petr_ln_bwd_proj's dy slabs)
 *   dy [M, 256] = gradient of the FFN output (after the output dropout's mask), hidden [M, F] = the forward's stored hidden
 *   (post ReLU and dropout: zero exactly where no gradient flows), alpha = the hidden dropout's 1/(1-p) (0 or 1: none).
 *   w1 [F, 256], w2 [256, F]: the nn.Linear weights as stored (both products are k-major in them). */
typedef struct {
  const float* dy; const float* w2; const float* hidden; float alpha; const float* w1;
  float* d_hidden; float* part; long part_stride;
  int M, F, n_split;
} petr_ffn_bwd_args;
int petr_ffn_bwd(const petr_ffn_bwd_args* a, void* stream);

/* y[i] = bf16(x[i]) (round to nearest even), n elements, both 16-byte aligned: produces the bf16 K/V operands
 * from the fp32 projections (the tensor .to(bfloat16) an autocast reference run would do) */
int petr_cast_bf16(const float* x, uint16_t* y, long n, void* stream);

/* backward: dq, dk, dv from q,k,v,o,do,lse (same stride conventions).  dq/dk/dv are ACCUMULATED
 * (+=): the caller zero-fills them.  dq (and dk/dv when the query range is split over workgroups)
 * use float atomics, so low-order bits depend on arrival order.  delta = rowsum(do*o) goes to ws. */
typedef struct {
  const float* q; long q_bs, q_hs, q_rs;
  const float* k; long k_bs, k_hs, k_rs;
  const float* v; long v_bs, v_hs, v_rs;
  const float* o; long o_bs, o_hs, o_rs;
  const float* d_o; long do_bs, do_hs, do_rs;
  const float* lse;
  const uint8_t* kpm;
  float* dq; long dq_bs, dq_hs, dq_rs;
  float* dk; long dk_bs, dk_hs, dk_rs;
  float* dv; long dv_bs, dv_hs, dv_rs;
  int B, H, Q, L;
  float scale;
  void* ws; size_t ws_bytes;
  petr_dropout drop;   /* must equal the forward's */
  const uint32_t* drop_bits;   /* optional with drop.p > 0: the KEY-major packed mask - what the forward left in its
                                * drop_bits, or bits_k of petr_dropout_bits() for the same (drop, B*H, Q, L); NULL: re-hash */
} petr_mha_bwd_args;
size_t petr_mha_bwd_workspace_bytes(int B, int H, int Q, int L);
int petr_mha_bwd(const petr_mha_bwd_args* a, void* stream);

/* bf16 variant of the backward (gradient of petr_mha_fwd_bf16; BASELINE configs 3-5: the training step in bf16):
 *     k, v are bf16 (raw uint16_t bits, strides in bf16 ELEMENTS, rows 16-byte aligned); q, o, d_o, lse, dq, dk, dv are
 *     fp32 (rows of q / o / d_o 16-byte aligned).  q is rounded to bf16 after the scale*log2(e) pre-multiplication (as
 *     the forward does), d_o, the recomputed probabilities and ds are rounded to bf16 for the second products; LSE,
 *     delta = rowsum(d_o*o) and all five accumulators are fp32 (v_mfma_f32_32x32x16_bf16).  Same call site and the
 *     same (+=) output convention as petr_mha_bwd; no workspace.                                              */
typedef struct {
  const float* q; long q_bs, q_hs, q_rs;
  const uint16_t* k; long k_bs, k_hs, k_rs;
  const uint16_t* v; long v_bs, v_hs, v_rs;
  const float* o; long o_bs, o_hs, o_rs;
  const float* d_o; long do_bs, do_hs, do_rs;
  const float* lse;
  const uint8_t* kpm;
  float* dq; long dq_bs, dq_hs, dq_rs;
  float* dk; long dk_bs, dk_hs, dk_rs;
  float* dv; long dv_bs, dv_hs, dv_rs;
  int B, H, Q, L;
  float scale;
  void* ws; size_t ws_bytes;   /* unused (0 bytes needed); kept so that the block mirrors petr_mha_bwd_args */
  petr_dropout drop;           /* must equal the forward's */
  const uint32_t* drop_bits;   /* as petr_mha_bwd_args.drop_bits */
  int dkv_overwrite;           /* 1: dk and dv are STORED (no zero-fill by the caller, no read-modify-write; the query range is
                                * then never split over workgroups); 0: accumulated like petr_mha_bwd.  dq is always += */
  int dkv_bf16;                /* 1 (needs dkv_overwrite): dk / dv point to bf16 storage (uint16_t bits, round to nearest even),
                                * their strides count bf16 elements: what the bf16 K/V-projection backward reads
                                * (PETR_GEMM_A_BF16) */
} petr_mha_bwd_bf16_args;
size_t petr_mha_bwd_bf16_workspace_bytes(int B, int H, int Q, int L);
int petr_mha_bwd_bf16(const petr_mha_bwd_bf16_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Box epilogue (models/dense_heads/petr_head.py:441-460; petrv2_head.py:513-531):
 *     t = reg[lvl,b,q,:]; t[0:2] = sigmoid(t[0:2] + logit(ref[q,0:2])); t[4] = sigmoid(t[4] + logit(ref[q,2]));
 *     t[8:] /= mean_time_stamp (if time_div != 0); then ch 0,1,4 scaled by pc_range.
 *     reg/out [rows, code] with rows = n_lvl*B*Q (row -> q = row % Q); in-place allowed.
 *     bwd: dreg (in-place on dout allowed) and dref [Q,3] accumulated (+=).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* reg; const float* ref; float* out;
  int rows, Q, code;
  float pc_range[6];
  float time_div;  /* 0 = no division */
  float eps;       /* 1e-5, inverse_sigmoid */
} petr_bbox_args;
int petr_bbox_epilogue_fwd(const petr_bbox_args* a, void* stream);
/* out = forward OUTPUT (post-scale); dout -> dreg; dref += */
int petr_bbox_epilogue_bwd(const petr_bbox_args* a, const float* dout, float* dreg, float* dref, void* stream);

/* Opt-in per-kernel timing with HIP events on the launch stream (bench.py's roofline leg; off by
 * default).  petr_prof_begin(capacity) arms it; tagged launches (tag & 15: 1 attention forward,
 * 2 attention backward, 4 coords3d; tag & 16: cross-attention, i.e. L > Q) record begin/end events;
 * petr_prof_end synchronises them and returns elapsed milliseconds per record.                 */
int petr_prof_begin(int capacity);
int petr_prof_end(float* ms, int* tags, int cap, int* n_out);

/* Top-down step of the FPN neck (models/necks/cp_fpn.py:175-186): dst[v, c, h, w] += src[v, c, hs, ws] with the nearest
 * source pixel of F.interpolate(mode='nearest') (hs = min(floor(h * Hs / H), Hs - 1) in fp32, as torch); src is NCHW
 * [V, C, Hs, Ws], dst is addressed through element strides (dst + v*sv + c*sc + h*sh + w*sw), so it may be NCHW or the
 * interior of a zero-padded channels-last map. */
int petr_fpn_upsample_add(float* dst, long sv, long sc, long sh, long sw, const float* src, int V, int C, int H, int W,
                          int Hs, int Ws, void* stream);

/* Adjoint of petr_fpn_upsample_add (the gradient autograd sends through F.interpolate(mode='nearest') + add,
 * models/necks/cp_fpn.py:175-186): dsrc[v, c, hs, ws] (+)= sum over the destination pixels (h, w) whose nearest source pixel is
 * (hs, ws) of ddst[v*sv + c*sc + h*sh + w*sw]; dsrc NCHW [V, C, Hs, Ws], ddst addressed through element strides like the
 * forward's dst.  accumulate: 0 overwrite, 1 add.  Gather form: one thread per source element, fixed summation order. */
int petr_fpn_upsample_add_bwd(float* dsrc, const float* ddst, long sv, long sc, long sh, long sw, int V, int C, int H, int W,
                              int Hs, int Ws, int accumulate, void* stream);
/* src NCHW [V, C, H, W] -> the interior of a channels-last map dst [V, H+2, W+2, C] whose one-pixel border the caller keeps at
 * zero: the layout the neck's 3x3 contraction reads (its backward needs the output gradient there; cp_fpn.py:190-192). */
int petr_nchw_to_padded_nhwc(const float* src, float* dst, int V, int C, int H, int W, void* stream);

/* small helpers used by the host executor */
/* SELayer gate of PETRv2 (petrv2_head.py:55-60): out = x * sigmoid(u); bwd: dx = dout*sig, du = dout*x*sig*(1-sig) */
int petr_gate_fwd(const float* x, const float* u, float* out, long n, void* stream);
int petr_gate_bwd(const float* dout, const float* x, const float* u, float* dx, float* du, long n, void* stream);
/* out[m,:] = x[m,:] + e[m % e_rows,:]   (key + key_pos, petr_transformer.py:343-344) */
int petr_add_rows(const float* x, const float* e, float* out, long M, int e_rows, int C, void* stream);
/* the same sum with bf16 x and bf16 out (bf16 mode of the head: memory and key = memory + key_pos live as bf16) */
int petr_add_rows_bf16(const uint16_t* x, const float* e, uint16_t* out, long M, int e_rows, int C, void* stream);
/* key = memory + key_pos where key_pos = e1 + e2 arrives as two halves (3D position encoder petr_head.py:286-334, adapt_pos3d
 * :400-402, produced on different streams): out[m,:] = x[m,:] + e1[m,:] + e2[m,:], and e1 += e2 in place (pos_embed, the sum
 * the reference builds at :402).  _bf16: x and out are bf16 (the bf16 mode's memory / key images). */
int petr_add_rows2(const float* x, float* e1, const float* e2, float* out, long M, int C, void* stream);
int petr_add_rows2_bf16(const uint16_t* x, float* e1, const float* e2, uint16_t* out, long M, int C, void* stream);
int petr_fill(float* p, float v, long n, void* stream);
int petr_axpy(float* y, const float* x, float alpha, long n, void* stream); /* y += alpha*x */
/* out[m,:] = sum_p x[p*stride + m*C..] (+ bias) (+ residual); generic partial reducer */
int petr_reduce_partials(const float* x, int n_partials, long stride, const float* bias, const float* residual,
                         float* out, long M, int C, void* stream);
/* sum over batch copies: out[r,:] (+)= sum_b x[(b*rows + r)*C ..]   (query_pos gradient over the batch) */
int petr_reduce_batch(const float* x, int B, long rows, int C, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * Grouped weight gradients: up to PETR_WGRAD_MAX independent contractions in ONE launch,
 *     dw_i[M_i, N_i] += sum_{k < K_i} dy_i[k, m] * x_i[k, n] ,   db_i[m] += sum_k dy_i[k, m]   (db optional)
 * the parameter gradients autograd forms for the nn.Linear / nn.MultiheadAttention weights of one decoder layer
 * (petr_transformer.py:158-224: self-attention in/out projections, cross-attention query / out projections, mmcv FFN
 * layers 0 and 1) - ten small-output, 900-row contractions per layer that petr_gemm would run as ten launches of
 * K-split workgroups with float atomics.  dy_i [K_i, >= M_i] and x_i [K_i, >= N_i] are row-major activations (leading
 * dimensions lda / ldb, multiples of 4, 16-byte aligned), M_i and N_i multiples of 64; dw_i row-major with leading
 * dimension ldc.  Each 64 x 64 output tile is owned by ONE workgroup that walks the whole K range (ksplit == 1: plain
 * read-add-store, bit-reproducible) or by ksplit workgroups that add their K slices with float atomics (token-sized K).
 * fp32 on v_mfma_f32_32x32x2_f32.  Returns PETR_ERR_UNSUPPORTED (nothing launched) if an item breaks the layout rules:
 * callers fall back to petr_gemm for that item. */
#define PETR_WGRAD_MAX 16
typedef struct {
  const float* dy; long lda;
  const float* x; long ldb;
  float* dw; long ldc;
  float* db;
  int M, N, K;
  int ksplit;
} petr_wgrad_item;
int petr_wgrad_grouped(const petr_wgrad_item* items, int n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training loss on the device: PETRHead.loss / loss_single / get_targets /
 * _get_target_single (models/dense_heads/petr_head.py:470-728) with HungarianAssigner3D.assign
 * (core/bbox/assigners/hungarian_assigner_3d.py:61-143; the reference goes device -> CPU scipy ->
 * device once per level and sample), FocalLossCost + BBox3DL1Cost (core/bbox/match_costs/
 * match_cost.py:6-27), normalize_bbox (core/bbox/util.py:38-58), mmdet FocalLoss / L1Loss.
 *     cls [NL,B,Q,NC] logits, box [NL,B,Q,CS>=10]; gt_boxes [Gtot,9] = (gravity centre xyz, w, l, h,
 *     yaw, vx, vy) of all samples back to back (what petr_head.py:697-699 builds), gt_labels [Gtot],
 *     gt_offsets [B+1] ints in HOST memory (B <= 64; read at launch), Gmax = largest per-sample count (<= Q), num_pos = sum_b min(G_b,Q).
 *     losses [NL,2] = (loss_cls, loss_bbox) per decoder level (level NL-1 is 'loss_cls'/'loss_bbox',
 *     level i < NL-1 is 'd{i}.loss_*'); d_cls / d_box (optional) = gradient of the SUM of all 2*NL losses
 *     (level l's losses depend on level l's predictions only, so a caller with other output weights
 *     scales the slices); assigned [NL,B,Q] = 0 background / k = ground truth k-1 of the sample.
 *     avg_factors (optional, DEVICE, 2 floats) = (cls_avg_factor, num_total_pos) BEFORE their max(., 1) clamps, as
 *     left by the caller's cross-rank reduce_mean (petr_head.py:620-622 with sync_cls_avg_factor, :630-631 always);
 *     read by the kernels, so the caller's all-reduce needs no host round trip.  NULL: the single-process values
 *     derived from num_pos.
 *     gt_labels outside [0, NC) never become addresses: such a box costs 100 for every query, its query matches no
 *     class in the focal loss, and the 32-bit word at ws + (NL*max(Gtot,1)*Q + 2*NL) * 8 bytes is set non-zero. */
typedef struct {
  const float* cls; const float* box;
  const float* gt_boxes; const int64_t* gt_labels; const int* gt_offsets;
  int NL, B, Q, NC, CS, Gtot, Gmax; long num_pos;
  float cls_weight, bbox_weight, alpha, gamma, bg_cls_weight;   /* c5: 2.0, 0.25, 0.25, 2.0, 0.0 */
  float code_weights[10];
  float* losses; float* d_cls; float* d_box; int* assigned;
  void* ws; size_t ws_bytes;
  const float* avg_factors;
} petr_loss_args;
size_t petr_loss_workspace_bytes(int NL, int B, int Q, int Gtot);
int petr_loss_fwd_bwd(const petr_loss_args* a, void* stream);

/* Box part of NMSFreeCoder.decode_single + get_bboxes (core/bbox/coders/nms_free_coder.py:62-97,
 * core/bbox/util.py:60-87, petr_head.py:745): for the n top-k entries `index` (into the flattened
 * [Q*num_classes] scores, as torch.topk returns them) of ONE sample: boxes [n,9] = denormalised box (z at
 * the gravity centre, or moved to the bottom centre), labels [n] = index % num_classes, keep [n] = centre
 * inside post_center_range (and score > score_threshold if score_threshold >= 0; negative: the reference's score_threshold=None).                                  */
typedef struct {
  const float* bbox_preds; const int64_t* index; const float* scores;
  float* boxes; int64_t* labels; uint8_t* keep;
  int n, num_classes, code;
  float post_center_range[6]; float score_threshold;
  int bottom_center;   /* 1: z = gravity centre - h/2 (get_bboxes); 0: gravity centre (NMSFreeCoder.decode) */
} petr_decode_args;
int petr_decode_boxes(const petr_decode_args* a, void* stream);

/* NMSFreeCoder.decode_single COMPLETE on the device (core/bbox/coders/nms_free_coder.py:48-97 + petr_head.py:730-751), one
 * launch for all samples of a batch: scores = sigmoid(cls_scores[b]) ; top-k of the flattened [Q*num_classes] scores
 * (sorted, descending: torch.topk) ; labels = index % num_classes ; boxes = denormalised bbox_preds[b][index / num_classes]
 * (z at the bottom centre if bottom_center) ; keep = centre inside post_center_range (and score > score_threshold if score_threshold >= 0; negative: the reference's score_threshold=None).
 *     cls_scores [B,Q,num_classes] LOGITS, bbox_preds [B,Q,code]; outputs [B,k,...]; k <= 1024; if Q*num_classes < k the
 *     tail entries have keep = 0 and index = -1.  index = the flat index torch.topk would return.                      */
typedef struct {
  const float* cls_scores; const float* bbox_preds;
  float* boxes; float* scores; int64_t* labels; uint8_t* keep; int64_t* index;
  int B, Q, num_classes, code, k;
  float post_center_range[6]; float score_threshold;
  int bottom_center;
} petr_decode_topk_args;
int petr_decode_topk(const petr_decode_topk_args* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Whole-path executor: PETRHead.forward (petr_head.py:366-468) / PETRv2Head.forward
 * (petrv2_head.py:429-540) and its gradient as ONE call each, all kernels enqueued on `stream`.
 * Parameters live in one flat fp32 buffer; `petr_head_layout` reports offsets (in floats) of every
 * reference state_dict tensor inside it so the host can alias its nn.Parameters onto the buffer.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int B, N, C_in, H, W;        /* mlvl_feats[position_level] = [B,N,C_in,H,W]            */
  int num_query, num_layers, num_heads, embed_dims, ffn_dims, depth_num, num_classes, code_size;
  int v2, with_fpe, with_time, with_multi;   /* PETRv2Head switches (petrv2_head.py:92-95) */
  int shared_branches;          /* 1: PETRHead aliasing (petr_head.py:244-247); 0: deep copies */
  int LID;
  float depth_start;
  float position_range[6];
  float pc_range[6];
  float pad_h, pad_w;
  int has_mask;                 /* 0: padding mask known all-False (host closed form)      */
  int training;                 /* 1: keep what backward needs                              */
} petr_head_config;

/* Execution context: a few side HIP streams + an event ring, so that work off the critical path
 * (weight-gradient contractions, the position-embedding branch, the second box branch) runs
 * concurrently with the dependent chain of small decoder kernels that otherwise leaves most of the 256
 * CUs idle at B=1.  Created/destroyed by the host (the only allocating calls of the library); passing
 * ctx = NULL in petr_head_io serialises everything on the caller's stream with identical results
 * (up to float-atomic ordering in the gradients).  Fork/join are plain event record/wait pairs, so the
 * calls remain capturable into a hipGraph.                                                        */
typedef struct petr_ctx petr_ctx;
int petr_ctx_create(petr_ctx** out, int n_side_streams);
int petr_ctx_destroy(petr_ctx* ctx);
/* side stream i of the context (a hipStream_t): lets the caller put its gradient exchange on a stream the context already owns
 * instead of one more stream - HIP maps streams onto 4 hardware queues, and a collective that shares the compute stream's queue
 * stalls it at every bucket boundary (measured: 180 us per boundary at c5) */
int petr_ctx_side_stream(petr_ctx* ctx, int i, void** stream);
/* `target_stream` waits for everything enqueued so far on `main_stream` AND on every side stream of `ctx`
 * (ctx may be NULL): how a consumer of partially finished work - the gradient exchange after a backward
 * stage range that is not the last - orders itself without stalling the compute stream.          */
int petr_ctx_join_into(petr_ctx* ctx, void* main_stream, void* target_stream);

#define PETR_MAX_PARAMS 512
typedef struct {
  int count;
  long total;                          /* floats in the flat buffer */
  char name[PETR_MAX_PARAMS][96];      /* reference state_dict key  */
  long offset[PETR_MAX_PARAMS];
  int ndim[PETR_MAX_PARAMS];
  int shape[PETR_MAX_PARAMS][4];
  int alias_of[PETR_MAX_PARAMS];       /* index of the entry whose storage this key shares, or -1 */
} petr_head_layout_t;
int petr_head_layout(const petr_head_config* cfg, petr_head_layout_t* out);

typedef struct {
  const float* params;          /* flat parameter buffer                                   */
  const float* feats;           /* [B,N,C_in,H,W]                                          */
  const float* img2lidar;       /* [B*N,16]                                                */
  const float* depth;           /* [D]                                                     */
  const float* dim_t;           /* [embed_dims/2]                                          */
  const uint8_t* mask;          /* [B,N,H,W] padding mask (uint8)                          */
  float time_div;               /* mean_time_stamp (v2 with_time), else 0                  */
  float* all_cls_scores;        /* [num_layers,B,Q,num_classes]                            */
  float* all_bbox_preds;        /* [num_layers,B,Q,code_size]                              */
  void* ws; size_t ws_bytes;    /* activations + scratch (petr_head_workspace_bytes)       */
  void* ctx;                    /* petr_ctx* (side streams) or NULL: everything on `stream` */
  float dropout_p;              /* training mode: the decoder's dropout rate (reference 0.1); 0 = eval */
  uint64_t dropout_seed;        /* fresh per forward; petr_head_bwd must get the forward's value     */
  int attn_bf16;                /* 1: bf16 mode (BASELINE configs 3-5): every token-sized contraction of the forward AND
                                 * of petr_head_bwd rounds its fp32 operands to bf16 (fp32 accumulate), K/V are stored as
                                 * bf16 and the cross-attention runs petr_mha_fwd_bf16 / petr_mha_bwd_bf16; parameters,
                                 * gradients, softmax, LayerNorm and the query-sized work stay fp32.  petr_head_bwd must
                                 * get the forward's value */
  const float* memory_in;       /* NULL, or the PROJECTED memory [B, N*H*W, C] (token-major, fp32) made upstream - a neck whose
                                 * last conv absorbed input_proj (cp_fpn.py:190-192 followed by petr_head.py:390 is one 3x3 conv
                                 * with weights W_proj W_3x3): petr_head_fwd then skips input_proj and never reads `feats` (may be
                                 * NULL).  Inference only: the fold has no separate input_proj gradient, petr_head_bwd refuses. */
} petr_head_io;
size_t petr_head_workspace_bytes(const petr_head_config* cfg);
int petr_head_fwd(const petr_head_config* cfg, const petr_head_io* io, void* stream);

typedef struct {
  const float* d_cls;           /* [num_layers,B,Q,num_classes] upstream gradient          */
  const float* d_bbox;          /* [num_layers,B,Q,code_size]                              */
  float* d_params;              /* flat, same layout as params; ACCUMULATED (+=)           */
  float* d_feats;               /* [B,N,C_in,H,W] or NULL; overwritten                     */
} petr_head_grads;
/* stage_begin/stage_end select a contiguous range of backward stages so the host can interleave
 * gradient all-reduce buckets; petr_head_bwd_num_stages() stages in total; after stage s completes,
 * the gradients of flat range [petr_head_bwd_stage_range(s)] are final.  With a side-stream context
 * "completes" means: on `stream` AND on the context's side streams - `stream` itself is joined with
 * them only by the call that runs the last stage; a consumer of an earlier range uses
 * petr_ctx_join_into().                                                                        */
int petr_head_bwd_num_stages(const petr_head_config* cfg);
int petr_head_bwd_stage_range(const petr_head_config* cfg, int stage, long* begin, long* end);
int petr_head_bwd(const petr_head_config* cfg, const petr_head_io* io, const petr_head_grads* g,
                  int stage_begin, int stage_end, void* stream);

/* One prediction branch of the head in ONE launch (SURVEY K10 "branches as one kernel"; reference petr_head.py:226-247 builds
 * `cls_branches` = (Linear, LayerNorm, ReLU) x 2 + Linear and `reg_branches` = (Linear, ReLU) x 2 + Linear, applied to every
 * decoder level at :440-460): y1 = act(x W1^T + b1), y2 = act(y1 W2^T + b2), out = y2 W3^T + b3 with act = ReLU(LayerNorm(.))
 * when g1 / g2 are given (class branch) and ReLU otherwise (box branch).  A workgroup owns 32 rows for the whole chain: the two
 * 256 x 256 products stream their k-major weights (w1t / w2t: TRANSPOSED copies [in][out]) past rows that never leave LDS, the
 * last n_out-wide product (n_out <= 16; w3 in nn.Linear layout [n_out][256]; NULL: stop after y2 - PETRv2's RegLayer heads
 * follow) is a dot product per output.  What the backward reads is written on the way when the pointer is given: h = pre-norm
 * rows, y = activations, per-row mean / rstd.  `groups` > 1: PETRv2's deep-copied branches (petrv2_head.py:294-307) - group g
 * owns rows [g rows, (g + 1) rows) and the parameters at + g * param_gs (transposed weights: + g * wt_gs).  C = 256 only.   */
typedef struct {
  const float* x;                                       /* [groups * rows, 256]                                    */
  const float* w1t; const float* b1; const float* g1; const float* be1;
  const float* w2t; const float* b2; const float* g2; const float* be2;
  const float* w3; const float* b3;
  long param_gs, wt_gs;
  float* h1; float* y1; float* h2; float* y2;           /* optional [groups * rows, 256] each                      */
  float* mean1; float* rstd1; float* mean2; float* rstd2;   /* optional [groups * rows] (class branch)             */
  float* out; int n_out;                                /* [groups * rows, n_out]                                  */
  int rows, groups;
  float eps;                                            /* LayerNorm epsilon                                       */
} petr_branch_fwd_args;
int petr_branch_fwd(const petr_branch_fwd_args* a, void* stream);

/* Input-gradient chain of one prediction branch in ONE launch (the backward of petr_branch_fwd; what autograd does for
 * petr_head.py:226-247): d_y2 = d_out W3 (or given: w3 = NULL, PETRv2's RegLayer heads produce it), d_h2 = act'(d_y2),
 * d_y1 = d_h2 W2, d_h1 = act'(d_y1), d_x = d_h1 W1, with act' = ReLU mask (y > 0) followed - when g1 / g2 are given - by the
 * LayerNorm backward (dz = rstd (g gamma - mean(g gamma) - xhat mean(g gamma xhat))), whose dgamma / dbeta are ADDED with float
 * atomics.  Weights in nn.Linear layout ([out][in] is k-major for an input gradient: nothing is transposed).  d_h2 / d_h1 are
 * written for the weight gradients (dW2 = d_h2^T y1, dW1 = d_h1^T x: petr_wgrad_grouped), d_x is overwritten; dW3 / db3 can
 * be formed here (dw3, db3).  32 rows per
 * workgroup, groups as in petr_branch_fwd.  C = 256 only.                                                              */
typedef struct {
  const float* d_out; int n_out; const float* w3;      /* [groups * rows, n_out], [n_out, 256]; or                    */
  const float* d_y2;                                    /* [groups * rows, 256] when w3 == NULL                        */
  const float* y2; const float* h2; const float* mean2; const float* rstd2; const float* g2;
  const float* w2;
  const float* y1; const float* h1; const float* mean1; const float* rstd1; const float* g1;
  const float* w1;
  long param_gs;
  float* d_h2; float* d_h1; float* d_x;                 /* [groups * rows, 256] each                                   */
  float* dg2; float* dbe2; float* dg1; float* dbe1;     /* += ; group stride param_gs (class branch)                   */
  int rows, groups;
  float* dw3; float* db3;                               /* optional, += (float atomics), group stride param_gs: the last
                                                         * Linear's parameter gradients d_out^T y2 [n_out, 256] and the
                                                         * column sums of d_out [n_out] - a 10 x 256 x 5 400 contraction is
                                                         * nothing for the workgroups that hold both operands anyway       */
} petr_branch_bwd_args;
int petr_branch_bwd(const petr_branch_bwd_args* a, void* stream);

/* PETRv2's RegLayer task heads (petrv2_head.py:81-95: `heads` = 5 x (Linear, ReLU, Linear -> group_reg_dims (2,1,3,2,2)), outputs
 * concatenated), the SECOND Linear of all heads in one launch: out[r, cols[t] + o] = h_t[r,:] . W2_t[o,:] + b2_t[o], o < dims[t].
 * As five contractions with 1-3 output columns they were scalar-load launches of 12-14 us each at the end of the forward and
 * 18-40 us each (input gradient) + 20-40 us each (weight gradient) at the start of the backward.  h / d_h: [groups][heads][rows][256];
 * head t of group g reads its parameters at + g * param_gs + t * head_stride ([dims[t]][256] weights, [dims[t]] bias).
 * _bwd: d_h = (h > 0) * (d_out_t W2_t)  (the ReLU in front of the Linear folded in), dW2 / db2 ADDED with float atomics.   */
typedef struct {
  const float* h; const float* w2; const float* b2;
  long param_gs, head_stride;
  float* out; int ld_out;                                /* [groups * rows, ld_out]                                 */
  int rows, groups, heads;
  int dims[8]; int cols[8];
} petr_task_heads_fwd_args;
int petr_task_heads_fwd(const petr_task_heads_fwd_args* a, void* stream);
typedef struct {
  const float* d_out; int ld_out;
  const float* h; const float* w2;
  long param_gs, head_stride;
  float* d_h; float* dw2; float* db2;
  int rows, groups, heads;
  int dims[8]; int cols[8];
} petr_task_heads_bwd_args;
int petr_task_heads_bwd(const petr_task_heads_bwd_args* a, void* stream);

/* named views into the forward workspace, for tests and for the per-module Python API
 * ("memory", "pos_embed", "query_embed", "outs_dec", "coords3d", "sine", "k_all", "v_all", ...)   */
int petr_head_ws_view(const petr_head_config* cfg, const char* name, long* offset_floats, long* numel);

#ifdef __cplusplus
}
#endif
#endif /* PETR_HIP_H_ */

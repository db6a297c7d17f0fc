// Attention output block of a decoder layer in one launch (petr_attn_out_ln, petr_hip.h):
//   ao = merge of the attention's L-split partials          (was: mha_combine_kernel)
//   z  = drop(ao W^T + bias) + residual                     (was: the out-projection contraction, gemm.hip)
//   y  = LayerNorm(z) gamma + beta,  y2 = y + add2          (was: layernorm_fwd_kernel)
// reference: the tail of nn.MultiheadAttention (out_proj) called at petr_transformer.py:357-367 / mmcv
// MultiheadAttention (identity + dropout_layer(proj_drop(out))) and the `norm` that follows it in
// multi_atten_decoder_layer.py:204-293.  These were four dependent links of the B = 1 chain (6 + 12 + 5 us and their
// launch gaps for 0.12 GFLOP); one workgroup of eight waves now owns 16 FULL rows:
//   prologue   the 16 x 256 attention rows are merged from the split partials (or read), normalised, written back as the
//              attention output the backward needs, and kept in LDS;
//   product    wave w computes columns 32 w .. + 31 on v_mfma_f32_16x16x4_f32 (exact fp32): per 16-deep K group a lane
//              takes one float4 of the A row block from LDS and four float2 of the TRANSPOSED weight straight from L2
//              (WStreamT below: lane (n, q) holds k = 16 g + 4 q .. + 3 of columns 2 n, 2 n + 1; step j multiplies element j
//              of both); half of K is in flight under the MFMAs of the other half, and the next product's first half
//              follows without a gap;
//   epilogue   bias, dropout, residual; row sums of the 16 rows across the 16 lanes of a row group (xor shuffles) and
//              the eight waves (LDS), two passes (mean, then centred squares) like layernorm_fwd_kernel;
//   optional   a second 256 x 256 projection of the normalised rows (+ the query_pos addend): the NEXT attention's query
//              projection (petr_transformer.py:341-362: q = (x + query_pos) Wq^T + bq) - the rows go back to LDS, the same
//              product runs again.  One more link of the chain in the same launch.
// 57 workgroups at 900 rows: 128 32-cycle MFMAs per wave and product, two waves per SIMD (3.4 us) are the floor of a product.
#include "wstream.h"

namespace {

constexpr float AO_LN2 = 0.6931471805599453f;

struct AoParams {
  petr_attn_out_ln_args a;
  DropDev drop;
};

__global__ __launch_bounds__(512) void attn_out_ln_kernel(const AoParams p) {
  __shared__ __attribute__((aligned(16))) float As[AO_ROWS * AO_PITCH];
  // row-block images in the [row][256] layout the global side is read / written in: the residual and the query_pos addend
  // come in with two float4 per thread, requested at the very start; z, y, y2 leave the same way.  (The accumulator layout
  // - a lane holds 4 rows x 2 columns - would make all of that 4-byte accesses issued behind the product.)
  __shared__ __attribute__((aligned(16))) float Rs[AO_ROWS * AO_PITCH];     // residual in, z out
  __shared__ __attribute__((aligned(16))) float Ys[AO_ROWS * AO_PITCH];     // y out
  __shared__ __attribute__((aligned(16))) float Ps[AO_ROWS * AO_PITCH];     // add2 in, y2 out
  __shared__ float red[2][8][AO_ROWS];
  const petr_attn_out_ln_args& a = p.a;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m0 = blockIdx.x * AO_ROWS;
  const int tr = t >> 5, tc = 8 * (t & 31);                 // this thread's piece of a row-block image
  const int tm = min(m0 + tr, a.M - 1);
  float4 rres[2], radd[2];
  rres[0] = rres[1] = radd[0] = radd[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.residual) {
    const float4* src = reinterpret_cast<const float4*>(a.residual + (long)tm * AO_C + tc);
    rres[0] = src[0]; rres[1] = src[1];
  }
  if (a.y2) {
    const float4* src = reinterpret_cast<const float4*>(a.add2 + (long)(a.add2_rows > 0 ? tm % a.add2_rows : tm) * AO_C + tc);
    radd[0] = src[0]; radd[1] = src[1];
  }

  // W fragments of the first half of K: requested before the prologue so that their latency runs under the merge
  const int nl = lane & 15, q4 = lane >> 4;
  const float* w0 = a.wT + 32 * wave;                      // this wave's 32 columns of wT [256][256]
  const uint32_t lo = WStreamT<1>::lane_off(lane, AO_C);
  WStreamT<1> ws;
  ws.first(w0, AO_C, lo);

  // ---- prologue: the A row block (thread: row t >> 5, head (t >> 2) & 7, 8 channels 8 (t & 3) .. + 7) ----
  {
    const int r = t >> 5, hd = (t >> 2) & 7, d0 = 8 * (t & 3);
    const int m = min(m0 + r, a.M - 1);
    float4 v0, v1;
    if (a.n_split > 1) {
      const int b = m / a.Q, q = m - b * a.Q;
      const long rows = (long)a.B * a.H * a.Q;
      const long row = ((long)b * a.H + hd) * a.Q + q;
      float M = -INFINITY, L = 0.f;
      v0 = v1 = make_float4(0.f, 0.f, 0.f, 0.f);
      auto add = [&](const float2 ml, const float4 x0, const float4 x1) {
        // M == -inf (every key of the row masked): exp2(-inf - -inf) = NaN, propagating like the reference
        const float wgt = __builtin_amdgcn_exp2f(ml.x - M);
        L += ml.y * wgt;
        v0.x += x0.x * wgt; v0.y += x0.y * wgt; v0.z += x0.z * wgt; v0.w += x0.w * wgt;
        v1.x += x1.x * wgt; v1.y += x1.y * wgt; v1.z += x1.z * wgt; v1.w += x1.w * wgt;
      };
      if (a.n_split <= 8) {        // block-uniform: every partial of the row requested before the first is used
        float2 ml[8];
        float4 x0[8], x1[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int sc = min(s, a.n_split - 1);
          ml[s] = *reinterpret_cast<const float2*>(a.ml_part + ((long)sc * rows + row) * 2);
          const float4* o = reinterpret_cast<const float4*>(a.o_part + ((long)sc * rows + row) * 32 + d0);
          x0[s] = o[0]; x1[s] = o[1];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) M = s < a.n_split ? fmaxf(M, ml[s].x) : M;
#pragma unroll
        for (int s = 0; s < 8; ++s)
          if (s < a.n_split) add(ml[s], x0[s], x1[s]);
      } else {
        for (int s = 0; s < a.n_split; ++s) M = fmaxf(M, a.ml_part[((long)s * rows + row) * 2]);
        for (int s = 0; s < a.n_split; ++s) {
          const float4* o = reinterpret_cast<const float4*>(a.o_part + ((long)s * rows + row) * 32 + d0);
          add(*reinterpret_cast<const float2*>(a.ml_part + ((long)s * rows + row) * 2), o[0], o[1]);
        }
      }
      const float inv = a.attn_scale / L;
      v0.x *= inv; v0.y *= inv; v0.z *= inv; v0.w *= inv;
      v1.x *= inv; v1.y *= inv; v1.z *= inv; v1.w *= inv;
      if (m0 + r < a.M) {
        float4* dst = reinterpret_cast<float4*>(a.a + (long)m * AO_C + 32 * hd + d0);
        dst[0] = v0; dst[1] = v1;
        if (a.lse && (t & 3) == 0) a.lse[row] = (M + log2f(L)) * AO_LN2;
      }
    } else {
      const float4* src = reinterpret_cast<const float4*>(a.a + (long)m * AO_C + 32 * hd + d0);
      v0 = src[0]; v1 = src[1];
    }
    float4* ls = reinterpret_cast<float4*>(As + r * AO_PITCH + 32 * hd + d0);
    ls[0] = v0; ls[1] = v1;
  }
  {
    float4* rs = reinterpret_cast<float4*>(Rs + tr * AO_PITCH + tc);
    rs[0] = rres[0]; rs[1] = rres[1];
    float4* ps = reinterpret_cast<float4*>(Ps + tr * AO_PITCH + tc);
    ps[0] = radd[0]; ps[1] = radd[1];
  }

  // ---- product: wave = 16 rows x 32 columns, K = 256 in 16 groups of 16 ----
  __syncthreads();
  f32x4 acc[1][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
  const float* arow = As + nl * AO_PITCH + 4 * q4;
  // second projection (optional): the first half of its weight follows the first product's in the stream, its latency runs
  // under the epilogue
  const float* v0 = a.w2T ? a.w2T + 32 * wave : nullptr;
  ws.run(arow, AO_PITCH, w0, AO_C, lo, v0, AO_C, lo, acc);
  const f32x4 acc0 = acc[0][0], acc1 = acc[0][1];

  // ---- epilogue: lane holds rows 4 q4 + i (i = 0..3), columns 32 wave + 2 nl (acc0) and + 1 (acc1) ----
  const int c0 = 32 * wave + 2 * nl, c1 = c0 + 1;
  const float bia0 = a.bias ? a.bias[c0] : 0.f, bia1 = a.bias ? a.bias[c1] : 0.f;
  float z0[4], z1[4], part[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + 4 * q4 + i;
    const int mc = min(m, a.M - 1);
    float u0 = acc0[i] + bia0, u1 = acc1[i] + bia1;
    if (p.drop.thr) {       // dropout of the sub-layer output, before the identity is added (as layernorm_fwd_kernel)
      const uint32_t rk = drop_row_key(p.drop, (uint32_t)mc);
      u0 = drop_keep(rk, (uint32_t)c0, p.drop.thr) ? u0 * p.drop.scale : 0.f;
      u1 = drop_keep(rk, (uint32_t)c1, p.drop.thr) ? u1 * p.drop.scale : 0.f;
    }
    u0 += Rs[(4 * q4 + i) * AO_PITCH + c0];          // zeros without a residual
    u1 += Rs[(4 * q4 + i) * AO_PITCH + c1];
    z0[i] = u0; z1[i] = u1;
    part[i] = u0 + u1;
  }
  auto row_reduce = [&](float (&v)[4], int slot) -> void {      // v[i] <- sum over the 256 columns of row 4 q4 + i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x = v[i];
      x += __shfl_xor(x, 1, 64);
      x += __shfl_xor(x, 2, 64);
      x += __shfl_xor(x, 4, 64);
      x += __shfl_xor(x, 8, 64);
      if (nl == 0) red[slot][wave][4 * q4 + i] = x;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float x = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) x += red[slot][w][4 * q4 + i];
      v[i] = x;
    }
  };
  row_reduce(part, 0);
  float mean[4], sq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    mean[i] = part[i] * (1.f / AO_C);
    const float d0 = z0[i] - mean[i], d1 = z1[i] - mean[i];
    sq[i] = d0 * d0 + d1 * d1;
  }
  row_reduce(sq, 1);
  const float g0 = a.gamma[c0], g1 = a.gamma[c1], be0 = a.beta[c0], be1 = a.beta[c1];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + 4 * q4 + i;
    if (m >= a.M) continue;
    const float rstd = 1.f / sqrtf(sq[i] * (1.f / AO_C) + a.eps);
    const float y0 = (z0[i] - mean[i]) * rstd * g0 + be0, y1 = (z1[i] - mean[i]) * rstd * g1 + be1;
    const int ro = (4 * q4 + i) * AO_PITCH;
    // every wave is past its reads of Rs / Ps (this lane's own elements: read above, written here) and of As (two barriers ago)
    const float u0 = y0 + Ps[ro + c0], u1 = y1 + Ps[ro + c1];     // + zeros without add2
    Rs[ro + c0] = z0[i]; Rs[ro + c1] = z1[i];
    Ys[ro + c0] = y0; Ys[ro + c1] = y1;
    Ps[ro + c0] = u0; Ps[ro + c1] = u1;
    As[ro + c0] = u0; As[ro + c1] = u1;                           // operand rows of the second projection
    if (wave == 0 && nl == 0) {
      if (a.mean) a.mean[m] = mean[i];
      if (a.rstd) a.rstd[m] = rstd;
    }
  }
  __syncthreads();
  if (m0 + tr < a.M) {             // the images leave as rows: two 16-byte stores per thread and output
    const long o = (long)(m0 + tr) * AO_C + tc;
    const int lo = tr * AO_PITCH + tc;
    if (a.z) {
      reinterpret_cast<float4*>(a.z + o)[0] = *reinterpret_cast<const float4*>(Rs + lo);
      reinterpret_cast<float4*>(a.z + o)[1] = *reinterpret_cast<const float4*>(Rs + lo + 4);
    }
    reinterpret_cast<float4*>(a.y + o)[0] = *reinterpret_cast<const float4*>(Ys + lo);
    reinterpret_cast<float4*>(a.y + o)[1] = *reinterpret_cast<const float4*>(Ys + lo + 4);
    if (a.y2) {
      reinterpret_cast<float4*>(a.y2 + o)[0] = *reinterpret_cast<const float4*>(Ps + lo);
      reinterpret_cast<float4*>(a.y2 + o)[1] = *reinterpret_cast<const float4*>(Ps + lo + 4);
    }
  }
  // ---- second projection: out2 = (y2, or y) W2^T + bias2 - the next attention's query projection of the same rows ----
  if (!a.w2T) return;
  acc[0][0] = acc[0][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  ws.run(arow, AO_PITCH, v0, AO_C, lo, nullptr, AO_C, lo, acc);
  const float bb0 = a.bias2 ? a.bias2[c0] : 0.f, bb1 = a.bias2 ? a.bias2[c1] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + 4 * q4 + i;
    if (m >= a.M) continue;
    *reinterpret_cast<float2*>(a.out2 + (long)m * AO_C + c0) = make_float2(acc[0][0][i] + bb0, acc[0][1][i] + bb1);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// petr_ln_proj: the LayerNorm that closes a decoder layer - sum of the FFN's split-K slabs + bias, dropout, residual, norm,
// query_pos add (layernorm_fwd_kernel's job) - AND the projection(s) that read its result: the next layer's self-attention
// in-projection (multi_atten_decoder_layer.py:223-237: q, k from x + query_pos, v from x).  grid = (row blocks of 16,
// column blocks of 256): every workgroup re-derives the 16 normalised rows (16 x 256 x n_partials loads - nothing next to
// a 256-deep product), block 0 writes them, each multiplies them by its own 256 rows of W.  One launch instead of two, and
// the projection's operand never leaves the chip.
// ---------------------------------------------------------------------------------------------------------------
struct LpParams {
  petr_ln_proj_args a;
  DropDev drop;
};

// half-wave sum: the 32 threads that share a row of a 16 x 256 row image (thread = row t >> 5, columns 8 (t & 31) .. + 7)
__device__ __forceinline__ float row32_sum(float x) {
  x += __shfl_xor(x, 1, 64);
  x += __shfl_xor(x, 2, 64);
  x += __shfl_xor(x, 4, 64);
  x += __shfl_xor(x, 8, 64);
  x += __shfl_xor(x, 16, 64);
  return x;
}

__global__ __launch_bounds__(512) void ln_proj_kernel(const LpParams p) {
  __shared__ __attribute__((aligned(16))) float As[AO_ROWS * AO_PITCH];
  const petr_ln_proj_args& a = p.a;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m0 = blockIdx.x * AO_ROWS, jb = blockIdx.y;
  const int nl = lane & 15, q4 = lane >> 4;
  const long ld2 = 256L * a.n2;
  const float* w0 = a.w2T + 256 * jb + 32 * wave;            // this wave's 32 columns of w2T [256][256 n2]
  const uint32_t lo = WStreamT<1>::lane_off(lane, ld2);
  WStreamT<1> ws;
  ws.first(w0, ld2, lo);
  // ---- the normalisation in the ROW-IMAGE layout: thread = (row t >> 5, 8 consecutive columns) - every global access is a
  // 16-byte one, a row's statistics are sums over one half wave (no LDS, no barrier) ----
  const int tr = t >> 5, tc = 8 * (t & 31);
  const int m = m0 + tr, mc = min(m, a.M - 1);
  float v[8];
  {
    const float4* x = reinterpret_cast<const float4*>(a.x + (long)mc * AO_C + tc);
    float4 s0 = x[0], s1 = x[1];
    for (int sp = 1; sp < a.n_partials; ++sp) {
      const float4* xs = reinterpret_cast<const float4*>(a.x + (long)sp * a.partial_stride + (long)mc * AO_C + tc);
      const float4 u0 = xs[0], u1 = xs[1];
      s0.x += u0.x; s0.y += u0.y; s0.z += u0.z; s0.w += u0.w;
      s1.x += u1.x; s1.y += u1.y; s1.z += u1.z; s1.w += u1.w;
    }
    v[0] = s0.x; v[1] = s0.y; v[2] = s0.z; v[3] = s0.w; v[4] = s1.x; v[5] = s1.y; v[6] = s1.z; v[7] = s1.w;
  }
  auto ld8 = [&](const float* ptr, float (&o)[8]) {
    const float4 u0 = reinterpret_cast<const float4*>(ptr)[0], u1 = reinterpret_cast<const float4*>(ptr)[1];
    o[0] = u0.x; o[1] = u0.y; o[2] = u0.z; o[3] = u0.w; o[4] = u1.x; o[5] = u1.y; o[6] = u1.z; o[7] = u1.w;
  };
  auto st8 = [&](float* ptr, const float (&o)[8]) {
    reinterpret_cast<float4*>(ptr)[0] = make_float4(o[0], o[1], o[2], o[3]);
    reinterpret_cast<float4*>(ptr)[1] = make_float4(o[4], o[5], o[6], o[7]);
  };
  float gam[8], bet[8], tmp[8];
  ld8(a.gamma + tc, gam);
  ld8(a.beta + tc, bet);
  if (a.bias) {
    ld8(a.bias + tc, tmp);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += tmp[j];
  }
  if (p.drop.thr) {
    const uint32_t rk = drop_row_key(p.drop, (uint32_t)mc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t hsh = drop_pair_hash(rk, (uint32_t)(tc + 2 * j) >> 1);
      v[2 * j] = (uint16_t)hsh >= (uint16_t)p.drop.thr ? v[2 * j] * p.drop.scale : 0.f;
      v[2 * j + 1] = (uint16_t)(hsh >> 16) >= (uint16_t)p.drop.thr ? v[2 * j + 1] * p.drop.scale : 0.f;
    }
  }
  if (a.residual) {
    ld8(a.residual + (long)mc * AO_C + tc, tmp);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] += tmp[j];
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) sum += v[j];
  const float mean = row32_sum(sum) * (1.f / AO_C);
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) sq += (v[j] - mean) * (v[j] - mean);
  const float rstd = 1.f / sqrtf(row32_sum(sq) * (1.f / AO_C) + a.eps);
  float y[8], y2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) y2[j] = y[j] = (v[j] - mean) * rstd * gam[j] + bet[j];
  if (a.add2) {
    ld8(a.add2 + (long)(a.add2_rows > 0 ? mc % a.add2_rows : mc) * AO_C + tc, tmp);
#pragma unroll
    for (int j = 0; j < 8; ++j) y2[j] = y[j] + tmp[j];
  }
  if (jb == 0 && m < a.M) {
    const long o = (long)m * AO_C + tc;
    if (a.z) st8(a.z + o, v);
    st8(a.y + o, y);
    if (a.y2) st8(a.y2 + o, y2);
    if ((t & 31) == 0) {
      if (a.mean) a.mean[m] = mean;
      if (a.rstd) a.rstd[m] = rstd;
    }
  }
  st8(As + tr * AO_PITCH + tc, jb < a.n2_pos ? y2 : y);
  __syncthreads();
  f32x4 acc[1][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
  const float* arow = As + nl * AO_PITCH + 4 * q4;
  ws.run(arow, AO_PITCH, w0, ld2, lo, nullptr, ld2, lo, acc);
  const int n0 = 256 * jb + 32 * wave + 2 * nl;
  const float bb0 = a.bias2 ? a.bias2[n0] : 0.f, bb1 = a.bias2 ? a.bias2[n0 + 1] : 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mo = m0 + 4 * q4 + i;
    if (mo >= a.M) continue;
    const float o0 = acc[0][0][i] + bb0, o1 = acc[0][1][i] + bb1;
    *reinterpret_cast<float2*>(a.out2 + (long)mo * ld2 + n0) = make_float2(o0, o1);
    if (a.out2_bf16) {
      typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
      const bf2 pk = {(__bf16)o0, (__bf16)o1};
      *reinterpret_cast<uint32_t*>(a.out2_bf16 + (long)mo * ld2 + n0) = __builtin_bit_cast(uint32_t, pk);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// petr_ln_bwd_proj: LayerNorm backward (petr_layernorm_bwd's job: dz = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma,
// with the split-K slab / identity-path prologue, the dropped copy of dz for the sub-layer branch, dgamma / dbeta by float
// atomics) AND the input gradient of the linear layer behind that branch - out = (dz_drop or dz) W, read through the
// transposed weight wT[K_in][256] - in one launch.  Same shape as ln_proj_kernel: grid = (row blocks of 16, blocks of 256
// output columns), every workgroup re-derives the 16 gradient rows, block 0 writes them and adds the column sums.
// Three of the eleven links of a decoder layer's backward chain (LN2 + FFN2, LN1 + cross out-proj, LN0 + self out-proj).
// ---------------------------------------------------------------------------------------------------------------
struct LbParams {
  petr_ln_bwd_proj_args a;
  DropDev drop;
};

__global__ __launch_bounds__(512) void ln_bwd_proj_kernel(const LbParams p) {
  __shared__ __attribute__((aligned(16))) float As[AO_ROWS * AO_PITCH];
  __shared__ __attribute__((aligned(16))) float Cs[2][AO_ROWS * AO_PITCH];   // dy xhat and dy of the block: column sums for dgamma / dbeta
  const petr_ln_bwd_proj_args& a = p.a;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int m0 = blockIdx.x * AO_ROWS, jb = blockIdx.y;
  const int nl = lane & 15, q4 = lane >> 4;
  const long ld2 = 256L * a.n2;
  const float* w0 = a.w ? a.w + 256 * jb + 32 * wave : nullptr;   // this wave's 32 columns of W [256][256 n2]: dx = dz W
  const uint32_t lo = WStreamT<1>::lane_off(lane, ld2), lop = WStreamT<1>::lane_off(lane, AO_C);
  WStreamT<1> ws;
  const int tr = t >> 5, tc = 8 * (t & 31);                  // row-image layout: thread = (row, 8 consecutive columns)
  const int m = m0 + tr, mc = min(m, a.M - 1);
  auto ld8 = [&](const float* ptr, float (&o)[8]) {
    const float4 u0 = reinterpret_cast<const float4*>(ptr)[0], u1 = reinterpret_cast<const float4*>(ptr)[1];
    o[0] = u0.x; o[1] = u0.y; o[2] = u0.z; o[3] = u0.w; o[4] = u1.x; o[5] = u1.y; o[6] = u1.z; o[7] = u1.w;
  };
  auto st8 = [&](float* ptr, const float (&o)[8]) {
    reinterpret_cast<float4*>(ptr)[0] = make_float4(o[0], o[1], o[2], o[3]);
    reinterpret_cast<float4*>(ptr)[1] = make_float4(o[4], o[5], o[6], o[7]);
  };
  float d[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) d[j] = 0.f;
  // optional leading product: dy = pre_a pre_w (+ dy_residual) - the input gradient of the projection that consumed the
  // normalised rows (the cross-attention's query projection), computed here instead of by a launch of its own; its
  // accumulators go through LDS into the row-image layout the LayerNorm backward below works in
  if (a.pre_a) {
    // pre_w [256 pre_n][256]: d(input) = pre_a pre_w, K = 256 pre_n in blocks of 256 - the (at most three) operand blocks sit in
    // As and the two column-sum images, which are not needed before the products are over
    const int pre_n = a.pre_n > 0 ? a.pre_n : 1;
    const float* u0 = a.pre_w + 32 * wave;
    ws.first(u0, AO_C, lop);
    for (int kb = 0; kb < pre_n; ++kb) {
      float blk[8];
      ld8(a.pre_a + (long)mc * AO_C * pre_n + 256 * kb + tc, blk);
      st8((kb == 0 ? As : Cs[kb - 1]) + tr * AO_PITCH + tc, blk);
    }
    __syncthreads();
    f32x4 pre[1][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
    for (int kb = 0; kb < pre_n; ++kb) {
      const float* blk = (kb == 0 ? As : Cs[kb - 1]) + nl * AO_PITCH + 4 * q4;
      const bool last = kb + 1 == pre_n;
      // the next block's weight - or the trailing product's - follows in the stream
      const float* wn = last ? (a.n2 > 0 ? w0 : nullptr) : u0 + (long)256 * (kb + 1) * AO_C;
      ws.run(blk, AO_PITCH, u0 + (long)256 * kb * AO_C, AO_C, lop, wn, last ? ld2 : (long)AO_C, last ? lo : lop, pre);
    }
    __syncthreads();                       // every wave is past its reads of the operand blocks
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float2*>(As + (4 * q4 + i) * AO_PITCH + 32 * wave + 2 * nl) = make_float2(pre[0][0][i], pre[0][1][i]);
    __syncthreads();
    ld8(As + tr * AO_PITCH + tc, d);
    // (As is rewritten below by the thread that just read the same elements)
  } else if (a.n2 > 0) {
    ws.first(w0, ld2, lo);
  }
  float tmp[8];
  for (int sp = 0; sp < a.dy_partials; ++sp) {
    ld8(a.dy + (long)sp * a.dy_partial_stride + (long)mc * AO_C + tc, tmp);
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] += tmp[j];
  }
  if (a.dy_residual) {
    ld8(a.dy_residual + (long)mc * AO_C + tc, tmp);
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] += tmp[j];
  }
  if (m >= a.M) {                                             // padding rows add nothing to the column sums
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = 0.f;
  }
  const float mean = a.mean[mc], rstd = a.rstd[mc];
  float xh[8], gm[8], g[8];
  ld8(a.z + (long)mc * AO_C + tc, xh);
  ld8(a.gamma + tc, gm);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    xh[j] = (xh[j] - mean) * rstd;
    g[j] = d[j] * gm[j];
    s1 += g[j];
    s2 += g[j] * xh[j];
  }
  const float m1 = row32_sum(s1) * (1.f / AO_C), m2 = row32_sum(s2) * (1.f / AO_C);
  float dz[8], e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) e[j] = dz[j] = rstd * (g[j] - m1 - xh[j] * m2);
  if (p.drop.thr) {                         // gradient of the dropped branch of z = drop(f) + residual
    const uint32_t rk = drop_row_key(p.drop, (uint32_t)mc);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t hsh = drop_pair_hash(rk, (uint32_t)(tc + 2 * j) >> 1);
      e[2 * j] = (uint16_t)hsh >= (uint16_t)p.drop.thr ? dz[2 * j] * p.drop.scale : 0.f;
      e[2 * j + 1] = (uint16_t)(hsh >> 16) >= (uint16_t)p.drop.thr ? dz[2 * j + 1] * p.drop.scale : 0.f;
    }
  }
  const bool writer = jb == 0;
  if (writer && m < a.M) {
    const long o = (long)m * AO_C + tc;
    st8(a.dz + o, dz);
    if (a.dz_drop) st8(a.dz_drop + o, e);
  }
  st8(As + tr * AO_PITCH + tc, e);
  const bool col_sums = writer && (a.dgamma || a.dbeta);
  if (col_sums) {
#pragma unroll
    for (int j = 0; j < 8; ++j) tmp[j] = d[j] * xh[j];
    st8(Cs[0] + tr * AO_PITCH + tc, tmp);
    st8(Cs[1] + tr * AO_PITCH + tc, d);
  }
  __syncthreads();
  if (col_sums) {                           // thread = (quantity t >> 8, column t & 255): the 16 rows' sum, one float atomic
    const float* cs = Cs[t >> 8] + (t & 255);
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < AO_ROWS; ++r) acc += cs[r * AO_PITCH];
    float* dst = (t >> 8) ? a.dbeta : a.dgamma;
    if (dst) atomicAdd(dst + (t & 255), acc);
  }
  if (a.n2 <= 0) return;                    // LayerNorm backward (behind the leading product) only
  f32x4 acc[1][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
  const float* arow = As + nl * AO_PITCH + 4 * q4;
  ws.run(arow, AO_PITCH, w0, ld2, lo, nullptr, ld2, lo, acc);
  const int n0 = 256 * jb + 32 * wave + 2 * nl;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mo = m0 + 4 * q4 + i;
    if (mo >= a.M) continue;
    float v0 = acc[0][0][i] * a.alpha, v1 = acc[0][1][i] * a.alpha;
    if (a.relu_mask) {
      const float2 rm = *reinterpret_cast<const float2*>(a.relu_mask + (long)mo * ld2 + n0);
      v0 = rm.x > 0.f ? v0 : 0.f;
      v1 = rm.y > 0.f ? v1 : 0.f;
    }
    *reinterpret_cast<float2*>(a.out + (long)mo * ld2 + n0) = make_float2(v0, v1);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// petr_ffn_fwd: both contractions of the FFN (mmcv FFN, SURVEY A.5: W2 drop(relu(W1 x + b1))) in one launch.  The hidden
// activation of a 16-row block never round-trips through HBM between the two products (it is written once, for the
// backward, when asked for).  A workgroup owns 32 rows and ONE slice of F / n_split hidden units: per 256 hidden units it
// multiplies the rows by that block of W1 (wave w: hidden columns 32 w .. + 31), applies bias / ReLU / dropout, parks the
// block in LDS, and multiplies it by the matching 256 columns of W2 into accumulators that live across the blocks.  The
// slice's partial sum leaves as slab `slice` of `part` - the layout the split-K FFN2 contraction wrote, summed (with the
// bias, the dropout and the residual) by the LayerNorm that follows (petr_ln_proj / petr_layernorm_fwd).
// XCD placement: workgroup ids are dealt round-robin to the 8 XCDs, so slice = f(id % 8) pins a slice's 2 x F/n_split x 256
// weights to 8 / n_split L2s instead of streaming all 4 MB of W1 and W2 through every L2.
// ---------------------------------------------------------------------------------------------------------------
struct FfParams {
  petr_ffn_fwd_args a;
  DropDev drop;
  int nrb;          // row blocks
  const float* mask;   // backward (petr_ffn_bwd): the forward's hidden [M, F]; the first product's result is zeroed where it is <= 0
  float alpha;         //   and scaled by the hidden dropout's 1 / (1 - p) elsewhere, instead of bias / ReLU / dropout
};

constexpr int FF_ROWS = 32;

template <bool BWD>
__global__ __launch_bounds__(512) void ffn_fwd_kernel(const FfParams p) {
  __shared__ __attribute__((aligned(16))) float As[FF_ROWS * AO_PITCH];     // the x rows
  __shared__ __attribute__((aligned(16))) float Hs[FF_ROWS * AO_PITCH];     // one block of 256 hidden units
  const petr_ffn_fwd_args& a = p.a;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nl = lane & 15, q4 = lane >> 4;
  const int rep = 8 / a.n_split;                     // XCDs per slice (n_split in {1, 2, 4, 8})
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int slice = xcd / rep;
  const int rb = seq * rep + (xcd - slice * rep);
  if (rb >= p.nrb) return;                            // block-uniform
  const int m0 = rb * FF_ROWS;
  const int hb = a.F / a.n_split;                     // hidden units of the slice, a multiple of 256
  const int nb = hb >> 8;
  const int h0 = slice * hb;
  const int tr = t >> 4, tc = 16 * (t & 15);          // this thread's piece of a 32 x 256 row image: 16 floats
  const int c0 = 32 * wave + 2 * nl;                  // this lane's two columns of a 256-wide block: c0 (acc[.][0]), c0 + 1 (acc[.][1])

  const float* w1 = a.w1t + h0 + 32 * wave;                    // w1t [256][F]: k = channel, columns = hidden units
  const float* w2 = a.w2t + (long)h0 * AO_C + 32 * wave;       // w2t [F][256]: k = hidden unit of the slice
  const uint32_t lo1 = WStreamT<2>::lane_off(lane, a.F), lo2 = WStreamT<2>::lane_off(lane, AO_C);
  WStreamT<2> ws;
  ws.first(w1, a.F, lo1);
  {
    const float4* src = reinterpret_cast<const float4*>(a.x + (long)min(m0 + tr, a.M - 1) * AO_C + tc);
    float4* ls = reinterpret_cast<float4*>(As + tr * AO_PITCH + tc);
    const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
    ls[0] = v0; ls[1] = v1; ls[2] = v2; ls[3] = v3;
  }
  uint32_t rk[2][4];
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int i = 0; i < 4; ++i) rk[rg][i] = p.drop.thr ? drop_row_key(p.drop, (uint32_t)min(m0 + 16 * rg + 4 * q4 + i, a.M - 1)) : 0u;
  float2 bia = a.b1 ? *reinterpret_cast<const float2*>(a.b1 + h0 + c0) : make_float2(0.f, 0.f);
  __syncthreads();
  const float* arow = As + nl * AO_PITCH + 4 * q4;
  const float* hrow = Hs + nl * AO_PITCH + 4 * q4;
  f32x4 out[2][2];
#pragma unroll
  for (int rg = 0; rg < 2; ++rg) out[rg][0] = out[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int b = 0; b < nb; ++b) {
    f32x4 acc[2][2];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) acc[rg][0] = acc[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* w2b = w2 + (long)256 * b * AO_C;
    float2 msk[2][4];
    if (BWD) {          // the forward's hidden at this lane's accumulator elements, requested ahead of the product
#pragma unroll
      for (int rg = 0; rg < 2; ++rg)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          msk[rg][i] = *reinterpret_cast<const float2*>(p.mask + (long)min(m0 + 16 * rg + 4 * q4 + i, a.M - 1) * a.F + h0 + 256 * b + c0);
    }
    ws.run(arow, AO_PITCH, w1 + 256 * b, a.F, lo1, w2b, AO_C, lo2, acc);
    const float2 bc = bia;
    if (b + 1 < nb && a.b1) bia = *reinterpret_cast<const float2*>(a.b1 + h0 + 256 * (b + 1) + c0);
    const int f0 = h0 + 256 * b + c0;                   // hidden unit of acc[.][0]; acc[.][1]: f0 + 1 (the same hash pair)
    if (b) __syncthreads();                             // every wave is past the previous block's reads of Hs
#pragma unroll
    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float u0, u1;
        if (BWD) {
          u0 = msk[rg][i].x > 0.f ? acc[rg][0][i] * p.alpha : 0.f;
          u1 = msk[rg][i].y > 0.f ? acc[rg][1][i] * p.alpha : 0.f;
        } else {
          u0 = fmaxf(acc[rg][0][i] + bc.x, 0.f);
          u1 = fmaxf(acc[rg][1][i] + bc.y, 0.f);
        }
        if (!BWD && p.drop.thr) {
          const uint32_t hsh = drop_pair_hash(rk[rg][i], (uint32_t)f0 >> 1);
          u0 = (uint16_t)hsh >= (uint16_t)p.drop.thr ? u0 * p.drop.scale : 0.f;
          u1 = (uint16_t)(hsh >> 16) >= (uint16_t)p.drop.thr ? u1 * p.drop.scale : 0.f;
        }
        *reinterpret_cast<float2*>(Hs + (16 * rg + 4 * q4 + i) * AO_PITCH + c0) = make_float2(u0, u1);
      }
    __syncthreads();
    if (a.hidden && m0 + tr < a.M) {                    // the block leaves as rows: four 16-byte stores per thread
      float4* dst = reinterpret_cast<float4*>(a.hidden + (long)(m0 + tr) * a.F + h0 + 256 * b + tc);
      const float4* hs = reinterpret_cast<const float4*>(Hs + tr * AO_PITCH + tc);
      dst[0] = hs[0]; dst[1] = hs[1]; dst[2] = hs[2]; dst[3] = hs[3];
    }
    ws.run(hrow, AO_PITCH, w2b, AO_C, lo2, b + 1 < nb ? w1 + 256 * (b + 1) : nullptr, a.F, lo1, out);
  }
  float* dst = a.part + (long)slice * a.part_stride;
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + 16 * rg + 4 * q4 + i;
      if (m >= a.M) continue;
      *reinterpret_cast<float2*>(dst + (long)m * AO_C + c0) = make_float2(out[rg][0][i], out[rg][1][i]);
    }
}
}  // namespace

extern "C" int petr_ffn_fwd(const petr_ffn_fwd_args* ap, void* stream) {
  PETR_CHECK(ap && ap->x && ap->w1t && ap->w2t && ap->part && ap->M > 0 && ap->F > 0, PETR_ERR_INVALID, "ffn_fwd: bad arguments");
  const petr_ffn_fwd_args& a = *ap;
  PETR_CHECK(a.n_split == 1 || a.n_split == 2 || a.n_split == 4 || a.n_split == 8, PETR_ERR_UNSUPPORTED,
             "ffn_fwd: n_split=%d not in {1, 2, 4, 8}", a.n_split);
  PETR_CHECK(a.F % (256 * a.n_split) == 0, PETR_ERR_UNSUPPORTED, "ffn_fwd: F=%d must be a multiple of 256 * n_split", a.F);
  PETR_CHECK(aligned16(a.x) && aligned16(a.w1t) && aligned16(a.w2t) && (!a.hidden || aligned16(a.hidden)) && aligned16(a.part) &&
                 (!a.b1 || aligned16(a.b1)) && !(a.part_stride & 3),
             PETR_ERR_INVALID, "ffn_fwd: x / w1t / w2t / b1 / hidden / part must be 16-byte aligned");
  PETR_CHECK(a.n_split == 1 || a.part_stride >= (long)a.M * AO_C, PETR_ERR_INVALID, "ffn_fwd: part_stride too small");
  PETR_CHECK(a.drop.p >= 0.f && a.drop.p < 1.f, PETR_ERR_INVALID, "ffn_fwd: dropout p=%g outside [0,1)", (double)a.drop.p);
  FfParams p;
  p.a = a;
  p.drop = make_drop(a.drop);
  p.nrb = (int)cdiv(a.M, FF_ROWS);
  const int rep = 8 / a.n_split;
  const long grid = cdiv(p.nrb, rep) * 8;
  p.mask = nullptr; p.alpha = 1.f;
  hipLaunchKernelGGL(ffn_fwd_kernel<false>, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, p);
  PETR_LAUNCH_CHECK("ffn_fwd");
  return PETR_OK;
}

// the FFN's input gradient: the same two chained products with the weights as stored (dh = dy W2 is k-major in W2 [256, F],
// dx = dh W1 in W1 [F, 256]) and the ReLU / dropout mask in place of bias, ReLU and dropout
extern "C" int petr_ffn_bwd(const petr_ffn_bwd_args* ap, void* stream) {
  PETR_CHECK(ap && ap->dy && ap->w2 && ap->hidden && ap->w1 && ap->d_hidden && ap->part && ap->M > 0 && ap->F > 0, PETR_ERR_INVALID,
             "ffn_bwd: bad arguments");
  const petr_ffn_bwd_args& b = *ap;
  PETR_CHECK(b.n_split == 1 || b.n_split == 2 || b.n_split == 4 || b.n_split == 8, PETR_ERR_UNSUPPORTED,
             "ffn_bwd: n_split=%d not in {1, 2, 4, 8}", b.n_split);
  PETR_CHECK(b.F % (256 * b.n_split) == 0, PETR_ERR_UNSUPPORTED, "ffn_bwd: F=%d must be a multiple of 256 * n_split", b.F);
  PETR_CHECK(aligned16(b.dy) && aligned16(b.w2) && aligned16(b.w1) && aligned16(b.hidden) && aligned16(b.d_hidden) && aligned16(b.part) &&
                 !(b.part_stride & 3),
             PETR_ERR_INVALID, "ffn_bwd: every operand must be 16-byte aligned");
  PETR_CHECK(b.n_split == 1 || b.part_stride >= (long)b.M * AO_C, PETR_ERR_INVALID, "ffn_bwd: part_stride too small");
  FfParams p;
  memset(&p.a, 0, sizeof p.a);
  p.a.x = b.dy; p.a.w1t = b.w2; p.a.b1 = nullptr; p.a.w2t = b.w1; p.a.hidden = b.d_hidden; p.a.part = b.part;
  p.a.part_stride = b.part_stride; p.a.M = b.M; p.a.F = b.F; p.a.n_split = b.n_split;
  p.drop = DropDev{0u, 0u, 0u, 1.f};
  p.nrb = (int)cdiv(b.M, FF_ROWS);
  p.mask = b.hidden; p.alpha = b.alpha == 0.f ? 1.f : b.alpha;
  const int rep = 8 / b.n_split;
  const long grid = cdiv(p.nrb, rep) * 8;
  hipLaunchKernelGGL(ffn_fwd_kernel<true>, dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, p);
  PETR_LAUNCH_CHECK("ffn_bwd");
  return PETR_OK;
}

extern "C" int petr_ln_bwd_proj(const petr_ln_bwd_proj_args* ap, void* stream) {
  PETR_CHECK(ap && ap->z && ap->mean && ap->rstd && ap->gamma && ap->dz && ap->M > 0 &&
                 ((ap->n2 > 0 && ap->w && ap->out) || (ap->n2 == 0 && ap->pre_a)) &&
                 ((ap->dy && ap->dy_partials > 0) || (ap->pre_a && ap->pre_w && ap->dy_partials == 0)),
             PETR_ERR_INVALID, "ln_bwd_proj: bad arguments");
  PETR_CHECK(ap->pre_n >= 0 && ap->pre_n <= 3, PETR_ERR_UNSUPPORTED, "ln_bwd_proj: pre_n=%d outside 0..3", ap->pre_n);
  PETR_CHECK(!ap->pre_a || (ap->pre_w && aligned16(ap->pre_a) && aligned16(ap->pre_w)), PETR_ERR_INVALID,
             "ln_bwd_proj: pre_a needs pre_w, both 16-byte aligned");
  const petr_ln_bwd_proj_args& a = *ap;
  PETR_CHECK((!a.w || aligned16(a.w)) && (!a.out || aligned16(a.out)) && (!a.relu_mask || aligned16(a.relu_mask)), PETR_ERR_INVALID,
             "ln_bwd_proj: w / out / relu_mask must be 16-byte aligned");
  PETR_CHECK(aligned16(a.z) && aligned16(a.gamma) && aligned16(a.dz) && (!a.dy || (aligned16(a.dy) && !(a.dy_partial_stride & 3))) &&
                 (!a.dy_residual || aligned16(a.dy_residual)) && (!a.dz_drop || aligned16(a.dz_drop)),
             PETR_ERR_INVALID, "ln_bwd_proj: every [.., 256] operand must be 16-byte aligned");
  PETR_CHECK(a.drop.p >= 0.f && a.drop.p < 1.f, PETR_ERR_INVALID, "ln_bwd_proj: dropout p=%g outside [0,1)", (double)a.drop.p);
  PETR_CHECK(!(a.drop.p > 0.f) || a.dz_drop, PETR_ERR_INVALID, "ln_bwd_proj: dropout needs dz_drop");
  LbParams p;
  p.a = a;
  p.drop = make_drop(a.drop);
  if (a.alpha == 0.f) p.a.alpha = 1.f;
  hipLaunchKernelGGL(ln_bwd_proj_kernel, dim3((unsigned)cdiv(a.M, AO_ROWS), (unsigned)(a.n2 > 0 ? a.n2 : 1)), dim3(512), 0, (hipStream_t)stream, p);
  PETR_LAUNCH_CHECK("ln_bwd_proj");
  return PETR_OK;
}

extern "C" int petr_ln_proj(const petr_ln_proj_args* ap, void* stream) {
  PETR_CHECK(ap && ap->x && ap->gamma && ap->beta && ap->y && ap->w2T && ap->out2 && ap->M > 0 && ap->n_partials > 0 && ap->n2 > 0,
             PETR_ERR_INVALID, "ln_proj: bad arguments");
  const petr_ln_proj_args& a = *ap;
  PETR_CHECK(aligned16(a.w2T) && aligned16(a.out2), PETR_ERR_INVALID, "ln_proj: w2T / out2 must be 16-byte aligned");
  PETR_CHECK(aligned16(a.x) && !(a.partial_stride & 3) && aligned16(a.gamma) && aligned16(a.beta) && aligned16(a.y) &&
                 (!a.bias || aligned16(a.bias)) && (!a.residual || aligned16(a.residual)) && (!a.z || aligned16(a.z)) &&
                 (!a.y2 || aligned16(a.y2)) && (!a.add2 || aligned16(a.add2)),
             PETR_ERR_INVALID, "ln_proj: every [.., 256] operand must be 16-byte aligned");
  PETR_CHECK(!a.y2 || a.add2, PETR_ERR_INVALID, "ln_proj: y2 without add2");
  PETR_CHECK(a.n2_pos >= 0 && a.n2_pos <= a.n2 && (a.n2_pos == 0 || a.add2), PETR_ERR_INVALID, "ln_proj: n2_pos needs add2");
  PETR_CHECK(a.drop.p >= 0.f && a.drop.p < 1.f, PETR_ERR_INVALID, "ln_proj: dropout p=%g outside [0,1)", (double)a.drop.p);
  LpParams p;
  p.a = a;
  p.drop = make_drop(a.drop);
  hipLaunchKernelGGL(ln_proj_kernel, dim3((unsigned)cdiv(a.M, AO_ROWS), (unsigned)a.n2), dim3(512), 0, (hipStream_t)stream, p);
  PETR_LAUNCH_CHECK("ln_proj");
  return PETR_OK;
}

extern "C" int petr_attn_out_ln(const petr_attn_out_ln_args* ap, void* stream) {
  PETR_CHECK(ap && ap->a && ap->wT && ap->gamma && ap->beta && ap->y && ap->M > 0, PETR_ERR_INVALID, "attn_out_ln: null pointer");
  const petr_attn_out_ln_args& a = *ap;
  PETR_CHECK(aligned16(a.a) && aligned16(a.wT), PETR_ERR_INVALID, "attn_out_ln: a / wT must be 16-byte aligned");
  if (a.n_split > 1) {
    PETR_CHECK(a.o_part && a.ml_part && a.B > 0 && a.H == 8 && a.Q > 0 && a.M == a.B * a.Q && aligned16(a.o_part) &&
                   (((uintptr_t)a.ml_part) & 7) == 0,
               PETR_ERR_INVALID, "attn_out_ln: merging needs o_part / ml_part, H == 8 and M == B * Q");
  }
  PETR_CHECK(!a.y2 || a.add2, PETR_ERR_INVALID, "attn_out_ln: y2 without add2");
  PETR_CHECK(!a.w2T || (a.out2 && aligned16(a.w2T) && aligned16(a.out2)), PETR_ERR_INVALID,
             "attn_out_ln: w2T needs out2 (both 16-byte aligned)");
  PETR_CHECK(a.drop.p >= 0.f && a.drop.p < 1.f, PETR_ERR_INVALID, "attn_out_ln: dropout p=%g outside [0,1)", (double)a.drop.p);
  AoParams p;
  p.a = a;
  p.drop = make_drop(a.drop);
  if (!(a.attn_scale > 0.f)) p.a.attn_scale = 1.f;
  hipLaunchKernelGGL(attn_out_ln_kernel, dim3((unsigned)cdiv(a.M, AO_ROWS)), dim3(512), 0, (hipStream_t)stream, p);
  PETR_LAUNCH_CHECK("attn_out_ln");
  return PETR_OK;
}

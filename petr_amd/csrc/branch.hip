// petr_branch_fwd (petr_hip.h): one prediction branch of the head - Linear, [LayerNorm,] ReLU, Linear, [LayerNorm,] ReLU, Linear
// (reference petr_head.py:226-247, applied to all decoder levels at :440-460) - in one launch.  Before: three contractions and
// two LayerNorm launches per branch over 5 400 rows, each a link of the chain that closes the forward (89 us at c5 for
// 1.4 GFLOP).  A workgroup of eight waves now owns 32 rows for the whole chain, as petr_ffn_fwd does for the FFN: wave w computes columns 32 w .. + 31 of a 256 x 256 product on v_mfma_f32_16x16x4_f32 with the k-major
// weight streamed from L2 (WStreamT, wstream.h) and the second weight's first half already in flight under the first epilogue;
// the activations go LDS -> LDS, global memory sees them only as the copies the backward asks for.
#include "wstream.h"

namespace {

struct BrParams {
  petr_branch_fwd_args a;
  int nrb;        // row blocks per group
};

// 32 rows per workgroup (two 16-row groups share every weight fragment): with 16 rows the 338 workgroups of a 5 400-row branch
// each streamed both 256 KB weights - 173 MB from L2 per launch, 48-58 us, bound by the L2, not by the matrix cores.
constexpr int BR_RG = 2, BR_ROWS = 16 * BR_RG;

__global__ __launch_bounds__(512) void branch_fwd_kernel(const BrParams p) {
  __shared__ __attribute__((aligned(16))) float As[BR_ROWS * AO_PITCH];     // x rows, then y2 rows
  __shared__ __attribute__((aligned(16))) float Hs[BR_ROWS * AO_PITCH];     // y1 rows
  __shared__ float red[2][8][BR_ROWS];
  const petr_branch_fwd_args& a = p.a;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = blockIdx.x / p.nrb, rb = blockIdx.x - grp * p.nrb;
  const int m0 = rb * BR_ROWS;                                  // first row of the block inside its group
  const long row0 = (long)grp * a.rows;                         // first row of the group
  const int tr = t >> 4, tc = 16 * (t & 15);                    // this thread's piece of a 32 x 256 row image: 16 floats
  const int nl = lane & 15, q4 = lane >> 4;
  const long pg = (long)grp * a.param_gs, wg = (long)grp * a.wt_gs;

  const float* w1 = a.w1t + wg + 32 * wave;
  const float* w2 = a.w2t + wg + 32 * wave;
  const uint32_t lo = WStreamT<BR_RG>::lane_off(lane, AO_C);
  WStreamT<BR_RG> ws;
  ws.first(w1, AO_C, lo);
  {
    const float4* src = reinterpret_cast<const float4*>(a.x + (row0 + min(m0 + tr, a.rows - 1)) * AO_C + tc);
    const float4 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
    float4* ls = reinterpret_cast<float4*>(As + tr * AO_PITCH + tc);
    ls[0] = v0; ls[1] = v1; ls[2] = v2; ls[3] = v3;
  }
  __syncthreads();

  const int c0 = 32 * wave + 2 * nl;               // this lane's two columns c0, c0 + 1; rows 16 rg + 4 q4 + i
  auto row_reduce = [&](float (&v)[BR_RG][4], int slot) -> void {     // v[rg][i] <- sum over the 256 columns of its row
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = v[rg][i];
        x += __shfl_xor(x, 1, 64);
        x += __shfl_xor(x, 2, 64);
        x += __shfl_xor(x, 4, 64);
        x += __shfl_xor(x, 8, 64);
        if (nl == 0) red[slot][wave][16 * rg + 4 * q4 + i] = x;
      }
    __syncthreads();
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) x += red[slot][w][16 * rg + 4 * q4 + i];
        v[rg][i] = x;
      }
  };
  // one (Linear, [LayerNorm,] ReLU) stage: rows of `src` times w, activation rows into `dst`; wn = the next stage's weight
  auto stage = [&](const float* src, float* dst, const float* w, const float* wn, const float* bias, const float* gamma,
                   const float* beta, float* h_save, float* y_save, float* mean_save, float* rstd_save) {
    f32x4 acc[BR_RG][2];
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg) acc[rg][0] = acc[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    ws.run(src + nl * AO_PITCH + 4 * q4, AO_PITCH, w, AO_C, lo, wn, AO_C, lo, acc);
    const float2 bb = *reinterpret_cast<const float2*>(bias + c0);
    float u0[BR_RG][4], u1[BR_RG][4];
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
      for (int i = 0; i < 4; ++i) { u0[rg][i] = acc[rg][0][i] + bb.x; u1[rg][i] = acc[rg][1][i] + bb.y; }
    if (gamma) {        // kernel-uniform
      float part[BR_RG][4], mean[BR_RG][4], sq[BR_RG][4];
#pragma unroll
      for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[rg][i] = u0[rg][i] + u1[rg][i];
      row_reduce(part, 0);
#pragma unroll
      for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          mean[rg][i] = part[rg][i] * (1.f / AO_C);
          const float d0 = u0[rg][i] - mean[rg][i], d1 = u1[rg][i] - mean[rg][i];
          sq[rg][i] = d0 * d0 + d1 * d1;
        }
      row_reduce(sq, 1);
      const float2 gg = *reinterpret_cast<const float2*>(gamma + c0), be = *reinterpret_cast<const float2*>(beta + c0);
#pragma unroll
      for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float rstd = 1.f / sqrtf(sq[rg][i] * (1.f / AO_C) + a.eps);
          const int rl = 16 * rg + 4 * q4 + i, m = m0 + rl;
          *reinterpret_cast<float2*>(dst + rl * AO_PITCH + c0) =
              make_float2(fmaxf((u0[rg][i] - mean[rg][i]) * rstd * gg.x + be.x, 0.f), fmaxf((u1[rg][i] - mean[rg][i]) * rstd * gg.y + be.y, 0.f));
          if (m < a.rows) {
            // the pre-norm rows leave from the accumulators: 16 lanes of a row cover one 128-byte line
            if (h_save) *reinterpret_cast<float2*>(h_save + (row0 + m) * AO_C + c0) = make_float2(u0[rg][i], u1[rg][i]);
            if (wave == 0 && nl == 0) {
              if (mean_save) mean_save[row0 + m] = mean[rg][i];
              if (rstd_save) rstd_save[row0 + m] = rstd;
            }
          }
        }
    } else {
#pragma unroll
      for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<float2*>(dst + (16 * rg + 4 * q4 + i) * AO_PITCH + c0) = make_float2(fmaxf(u0[rg][i], 0.f), fmaxf(u1[rg][i], 0.f));
    }
    __syncthreads();
    if (y_save && m0 + tr < a.rows) {       // the activations leave as rows: four 16-byte stores per thread
      float4* o = reinterpret_cast<float4*>(y_save + (row0 + m0 + tr) * AO_C + tc);
      const float4* li = reinterpret_cast<const float4*>(dst + tr * AO_PITCH + tc);
      o[0] = li[0]; o[1] = li[1]; o[2] = li[2]; o[3] = li[3];
    }
  };
  stage(As, Hs, w1, w2, a.b1 + pg, a.g1 ? a.g1 + pg : nullptr, a.g1 ? a.be1 + pg : nullptr, a.h1, a.y1, a.mean1, a.rstd1);
  stage(Hs, As, w2, nullptr, a.b2 + pg, a.g2 ? a.g2 + pg : nullptr, a.g2 ? a.be2 + pg : nullptr, a.h2, a.y2, a.mean2, a.rstd2);
  if (!a.w3) return;
  // out = y2 W3^T + b3: n_out <= 16 outputs per row, one dot product of 256 per thread (32 rows x n_out threads)
  const int r = t & 31, o = t >> 5;
  if (o < a.n_out && m0 + r < a.rows) {
    const float4* wr = reinterpret_cast<const float4*>(a.w3 + pg + (long)o * AO_C);
    const float4* yr = reinterpret_cast<const float4*>(As + r * AO_PITCH);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
    for (int k = 0; k < AO_C / 4; ++k) {
      const float4 w = wr[k], y = yr[k];
      s0 += w.x * y.x; s1 += w.y * y.y; s2 += w.z * y.z; s3 += w.w * y.w;
    }
    a.out[(row0 + m0 + r) * a.n_out + o] = (s0 + s1) + (s2 + s3) + (a.b3 ? a.b3[pg + o] : 0.f);
  }
}
}  // namespace

extern "C" int petr_branch_fwd(const petr_branch_fwd_args* ap, void* stream) {
  PETR_CHECK(ap && ap->x && ap->w1t && ap->b1 && ap->w2t && ap->b2 && ap->rows > 0 && ap->groups > 0, PETR_ERR_INVALID,
             "branch_fwd: bad arguments");
  const petr_branch_fwd_args& a = *ap;
  PETR_CHECK((a.g1 != nullptr) == (a.be1 != nullptr) && (a.g2 != nullptr) == (a.be2 != nullptr), PETR_ERR_INVALID,
             "branch_fwd: LayerNorm weight and bias come together");
  PETR_CHECK(!a.w3 || (a.out && a.n_out > 0 && a.n_out <= 16), PETR_ERR_UNSUPPORTED, "branch_fwd: n_out=%d outside 1..16", a.n_out);
  PETR_CHECK(aligned16(a.x) && aligned16(a.w1t) && aligned16(a.w2t) && aligned16(a.h1) && aligned16(a.y1) && aligned16(a.h2) &&
                 aligned16(a.y2) && (!a.w3 || aligned16(a.w3)) && !(a.param_gs & 3) && !(a.wt_gs & 3),
             PETR_ERR_INVALID, "branch_fwd: x / weights / saved rows must be 16-byte aligned (group strides multiples of 4)");
  BrParams p;
  p.a = a;
  p.nrb = (int)cdiv(a.rows, BR_ROWS);
  hipLaunchKernelGGL(branch_fwd_kernel, dim3((unsigned)(p.nrb * a.groups)), dim3(512), 0, (hipStream_t)stream, p);
  PETR_LAUNCH_CHECK("branch_fwd");
  return PETR_OK;
}

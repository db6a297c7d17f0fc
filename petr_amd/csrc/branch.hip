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

// ---------------------------------------------------------------------------------------------------------------
// petr_branch_bwd: the input-gradient chain of a branch, same shape as the forward - 32 rows per workgroup, two streamed 256 x 256
// products (W2, W1 as stored: [out][in] is k-major for d_in = d_out W), everything between them in the accumulator layout
// (lane: rows 16 rg + 4 q4 + i, columns c0, c0 + 1): ReLU mask from the saved activation, LayerNorm backward with row sums over
// the 16 lanes of a row and the eight waves (one barrier for both sums), dgamma / dbeta as column sums over the lane's 8 rows and
// the four lane groups of the wave (a wave covers all 32 rows of its 32 columns) -> one float atomic per column and workgroup.
// ---------------------------------------------------------------------------------------------------------------
struct BbParams {
  petr_branch_bwd_args a;
  int nrb;
};

__global__ __launch_bounds__(512) void branch_bwd_kernel(const BbParams p) {
  __shared__ __attribute__((aligned(16))) float As[BR_ROWS * AO_PITCH];     // d_h2 rows
  __shared__ __attribute__((aligned(16))) float Hs[BR_ROWS * AO_PITCH];     // d_h1 rows
  __shared__ float red[2][8][BR_ROWS];
  __shared__ float Ds[BR_ROWS][17];                                         // the block's d_out rows (n_out <= 16)
  const petr_branch_bwd_args& a = p.a;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = blockIdx.x / p.nrb, rb = blockIdx.x - grp * p.nrb;
  const int m0 = rb * BR_ROWS;
  const long row0 = (long)grp * a.rows;
  const int nl = lane & 15, q4 = lane >> 4;
  const long pg = (long)grp * a.param_gs;
  const int c0 = 32 * wave + 2 * nl;

  const float* w2 = a.w2 + pg + 32 * wave;
  const float* w1 = a.w1 + pg + 32 * wave;
  const uint32_t lo = WStreamT<BR_RG>::lane_off(lane, AO_C);
  WStreamT<BR_RG> ws;
  ws.first(w2, AO_C, lo);

  // ---- d_y2 in the accumulator layout ----
  float u0[BR_RG][4], u1[BR_RG][4];
  if (a.w3) {
    if (t < BR_ROWS * 16) {
      const int r = t >> 4, o = t & 15;
      Ds[r][o] = (o < a.n_out) ? a.d_out[(row0 + min(m0 + r, a.rows - 1)) * a.n_out + o] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
      for (int i = 0; i < 4; ++i) u0[rg][i] = u1[rg][i] = 0.f;
    for (int o = 0; o < a.n_out; ++o) {
      const float2 w = *reinterpret_cast<const float2*>(a.w3 + pg + (long)o * AO_C + c0);
#pragma unroll
      for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float dv = Ds[16 * rg + 4 * q4 + i][o];
          u0[rg][i] += dv * w.x; u1[rg][i] += dv * w.y;
        }
    }
  } else {
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float2 v = *reinterpret_cast<const float2*>(a.d_y2 + (row0 + min(m0 + 16 * rg + 4 * q4 + i, a.rows - 1)) * AO_C + c0);
        u0[rg][i] = v.x; u1[rg][i] = v.y;
      }
  }

  // ---- the last Linear's parameter gradients from the operands this workgroup holds anyway: dW3[o][c] += sum_r d_out[r][o] y2[r][c]
  //      (lane: its 8 rows x 2 columns, then the four lane groups of the wave), db3[o] += sum_r d_out[r][o] ----
  if (a.w3 && a.dw3) {
    float2 yv[BR_RG][4];
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + 16 * rg + 4 * q4 + i;
        const float2 y = *reinterpret_cast<const float2*>(a.y2 + (row0 + min(m, a.rows - 1)) * AO_C + c0);
        yv[rg][i] = m < a.rows ? y : make_float2(0.f, 0.f);
      }
    for (int o = 0; o < a.n_out; ++o) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float dv = Ds[16 * rg + 4 * q4 + i][o];
          s0 += dv * yv[rg][i].x; s1 += dv * yv[rg][i].y;
        }
      s0 += __shfl_xor(s0, 16, 64); s1 += __shfl_xor(s1, 16, 64);
      s0 += __shfl_xor(s0, 32, 64); s1 += __shfl_xor(s1, 32, 64);
      if (q4 == 0) {
        atomicAdd(a.dw3 + pg + (long)o * AO_C + c0, s0);
        atomicAdd(a.dw3 + pg + (long)o * AO_C + c0 + 1, s1);
      }
    }
    if (a.db3 && t < a.n_out) {
      float sb = 0.f;
      for (int r = 0; r < BR_ROWS; ++r) sb += (m0 + r < a.rows) ? Ds[r][t] : 0.f;
      atomicAdd(a.db3 + pg + t, sb);
    }
  }

  // gradient through ReLU [and LayerNorm] of one stage, in place: (u0, u1) = d_y -> d_h; rows beyond the group end become zero
  auto act_bwd = [&](const float* y_save, const float* h_save, const float* mean_save, const float* rstd_save, const float* gamma,
                     float* dgamma, float* dbeta, float* dst, float* dh_save) {
    float xh0[BR_RG][4], xh1[BR_RG][4], rs[BR_RG][4];
    float cg0 = 0.f, cg1 = 0.f, cb0 = 0.f, cb1 = 0.f;
    float2 gm = make_float2(1.f, 1.f);
    if (gamma) gm = *reinterpret_cast<const float2*>(gamma + c0);
    float s1[BR_RG][4], s2[BR_RG][4];
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + 16 * rg + 4 * q4 + i;
        const bool ok = m < a.rows;
        const long ro = (row0 + min(m, a.rows - 1)) * AO_C + c0;
        const float2 y = *reinterpret_cast<const float2*>(y_save + ro);
        float g0 = (ok && y.x > 0.f) ? u0[rg][i] : 0.f, g1 = (ok && y.y > 0.f) ? u1[rg][i] : 0.f;
        if (gamma) {
          const float2 hv = *reinterpret_cast<const float2*>(h_save + ro);
          const float mean = mean_save[row0 + min(m, a.rows - 1)], rstd = rstd_save[row0 + min(m, a.rows - 1)];
          xh0[rg][i] = (hv.x - mean) * rstd; xh1[rg][i] = (hv.y - mean) * rstd; rs[rg][i] = rstd;
          cg0 += g0 * xh0[rg][i]; cg1 += g1 * xh1[rg][i]; cb0 += g0; cb1 += g1;
          g0 *= gm.x; g1 *= gm.y;
          s1[rg][i] = g0 + g1;
          s2[rg][i] = g0 * xh0[rg][i] + g1 * xh1[rg][i];
        }
        u0[rg][i] = g0; u1[rg][i] = g1;
      }
    if (gamma) {
      // both row sums behind one barrier
#pragma unroll
      for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x = s1[rg][i], z = s2[rg][i];
          x += __shfl_xor(x, 1, 64); z += __shfl_xor(z, 1, 64);
          x += __shfl_xor(x, 2, 64); z += __shfl_xor(z, 2, 64);
          x += __shfl_xor(x, 4, 64); z += __shfl_xor(z, 4, 64);
          x += __shfl_xor(x, 8, 64); z += __shfl_xor(z, 8, 64);
          if (nl == 0) { red[0][wave][16 * rg + 4 * q4 + i] = x; red[1][wave][16 * rg + 4 * q4 + i] = z; }
        }
      __syncthreads();
#pragma unroll
      for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x = 0.f, z = 0.f;
#pragma unroll
          for (int w = 0; w < 8; ++w) { x += red[0][w][16 * rg + 4 * q4 + i]; z += red[1][w][16 * rg + 4 * q4 + i]; }
          const float m1 = x * (1.f / AO_C), m2 = z * (1.f / AO_C);
          u0[rg][i] = rs[rg][i] * (u0[rg][i] - m1 - xh0[rg][i] * m2);
          u1[rg][i] = rs[rg][i] * (u1[rg][i] - m1 - xh1[rg][i] * m2);
        }
      // column sums over the wave's 32 rows: the lane's 8, then the four lane groups
      cg0 += __shfl_xor(cg0, 16, 64); cg1 += __shfl_xor(cg1, 16, 64); cb0 += __shfl_xor(cb0, 16, 64); cb1 += __shfl_xor(cb1, 16, 64);
      cg0 += __shfl_xor(cg0, 32, 64); cg1 += __shfl_xor(cg1, 32, 64); cb0 += __shfl_xor(cb0, 32, 64); cb1 += __shfl_xor(cb1, 32, 64);
      if (q4 == 0) {
        if (dgamma) { atomicAdd(dgamma + c0, cg0); atomicAdd(dgamma + c0 + 1, cg1); }
        if (dbeta) { atomicAdd(dbeta + c0, cb0); atomicAdd(dbeta + c0 + 1, cb1); }
      }
    }
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = 16 * rg + 4 * q4 + i, m = m0 + rl;
        const float2 v = make_float2(u0[rg][i], u1[rg][i]);
        *reinterpret_cast<float2*>(dst + rl * AO_PITCH + c0) = v;
        if (dh_save && m < a.rows) *reinterpret_cast<float2*>(dh_save + (row0 + m) * AO_C + c0) = v;
      }
    __syncthreads();
  };
  act_bwd(a.y2, a.h2, a.mean2, a.rstd2, a.g2 ? a.g2 + pg : nullptr, a.dg2 ? a.dg2 + pg : nullptr, a.dbe2 ? a.dbe2 + pg : nullptr, As, a.d_h2);
  {
    f32x4 acc[BR_RG][2];
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg) acc[rg][0] = acc[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    ws.run(As + nl * AO_PITCH + 4 * q4, AO_PITCH, w2, AO_C, lo, w1, AO_C, lo, acc);
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
      for (int i = 0; i < 4; ++i) { u0[rg][i] = acc[rg][0][i]; u1[rg][i] = acc[rg][1][i]; }
  }
  act_bwd(a.y1, a.h1, a.mean1, a.rstd1, a.g1 ? a.g1 + pg : nullptr, a.dg1 ? a.dg1 + pg : nullptr, a.dbe1 ? a.dbe1 + pg : nullptr, Hs, a.d_h1);
  {
    f32x4 acc[BR_RG][2];
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg) acc[rg][0] = acc[rg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    ws.run(Hs + nl * AO_PITCH + 4 * q4, AO_PITCH, w1, AO_C, lo, nullptr, AO_C, lo, acc);
#pragma unroll
    for (int rg = 0; rg < BR_RG; ++rg)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + 16 * rg + 4 * q4 + i;
        if (m < a.rows) *reinterpret_cast<float2*>(a.d_x + (row0 + m) * AO_C + c0) = make_float2(acc[rg][0][i], acc[rg][1][i]);
      }
  }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// petr_task_heads_fwd / _bwd (petr_hip.h): the 1-3 column Linear that ends each of PETRv2's five RegLayer heads.  Vector work, not
// matrix work: workgroup = 32 rows of one (group, head), thread = (row t >> 3, 32-channel slice t & 7): eight lanes of a row read
// its 1 KB contiguously, dot products are finished with three xor shuffles.
// ---------------------------------------------------------------------------------------------------------------
namespace {
constexpr int TH_MAXD = 4;      // outputs per head (reference: <= 3)

__global__ __launch_bounds__(256) void task_heads_fwd_kernel(const petr_task_heads_fwd_args a) {
  const int t = threadIdx.x, r = t >> 3, part = t & 7;
  const int gh = blockIdx.y, grp = gh / a.heads, hd = gh - grp * a.heads;
  const int m = blockIdx.x * 32 + r;
  const int nd = a.dims[hd];
  const float* hrow = a.h + ((long)gh * a.rows + min(m, a.rows - 1)) * 256 + 32 * part;
  const float* w = a.w2 + (long)grp * a.param_gs + (long)hd * a.head_stride + 32 * part;
  float4 hv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) hv[i] = reinterpret_cast<const float4*>(hrow)[i];
  for (int o = 0; o < nd; ++o) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 wv = reinterpret_cast<const float4*>(w + (long)o * 256)[i];
      s += hv[i].x * wv.x + hv[i].y * wv.y + hv[i].z * wv.z + hv[i].w * wv.w;
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (part == 0 && m < a.rows)
      a.out[((long)grp * a.rows + m) * a.ld_out + a.cols[hd] + o] = s + (a.b2 ? a.b2[(long)grp * a.param_gs + (long)hd * a.head_stride + o] : 0.f);
  }
}

__global__ __launch_bounds__(256) void task_heads_bwd_kernel(const petr_task_heads_bwd_args a) {
  // a workgroup walks row blocks blockIdx.x, + gridDim.x, .. of its (group, head) and keeps its share of dW2 / db2 in registers:
  // one pass over the rows, gridDim.x float atomics per parameter at the end (one atomic per 32-row block and element - the first
  // version - queued 116 adders on each of 2 560 addresses: 312 us)
  __shared__ float red[4][TH_MAXD][256 + 8];
  const int t = threadIdx.x, r = t >> 3, part = t & 7, wave = t >> 6;
  const int gh = blockIdx.y, grp = gh / a.heads, hd = gh - grp * a.heads;
  const int nd = a.dims[hd];
  const long prow = (long)grp * a.param_gs + (long)hd * a.head_stride;
  const float* w = a.w2 + prow + 32 * part;
  float4 acc[TH_MAXD][8];
  float bacc[TH_MAXD];
#pragma unroll
  for (int o = 0; o < TH_MAXD; ++o) {
    bacc[o] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[o][i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int nrb = (a.rows + 31) >> 5;
  for (int rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
    const int m = rb * 32 + r;
    const bool ok = m < a.rows;
    const long hoff = ((long)gh * a.rows + min(m, a.rows - 1)) * 256 + 32 * part;
    float dv[TH_MAXD];
#pragma unroll
    for (int o = 0; o < TH_MAXD; ++o)
      dv[o] = (ok && o < nd) ? a.d_out[((long)grp * a.rows + m) * a.ld_out + a.cols[hd] + o] : 0.f;
    float4 hv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) hv[i] = reinterpret_cast<const float4*>(a.h + hoff)[i];
    if (a.d_h) {       // input gradient of the Linear with the ReLU in front of it folded in
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int o = 0; o < TH_MAXD; ++o)
          if (o < nd) {
            const float4 wv = reinterpret_cast<const float4*>(w + (long)o * 256)[i];
            g.x += dv[o] * wv.x; g.y += dv[o] * wv.y; g.z += dv[o] * wv.z; g.w += dv[o] * wv.w;
          }
        g.x = hv[i].x > 0.f ? g.x : 0.f; g.y = hv[i].y > 0.f ? g.y : 0.f; g.z = hv[i].z > 0.f ? g.z : 0.f; g.w = hv[i].w > 0.f ? g.w : 0.f;
        if (ok) reinterpret_cast<float4*>(a.d_h + hoff)[i] = g;
      }
    }
#pragma unroll
    for (int o = 0; o < TH_MAXD; ++o) {
      bacc[o] += dv[o];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[o][i].x += dv[o] * hv[i].x; acc[o][i].y += dv[o] * hv[i].y; acc[o][i].z += dv[o] * hv[i].z; acc[o][i].w += dv[o] * hv[i].w;
      }
    }
  }
  if (!a.dw2 && !a.db2) return;
  // the 8 rows of a wave by xor shuffles (lane = 8 row + part), the four waves through LDS
#pragma unroll
  for (int o = 0; o < TH_MAXD; ++o) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float4 p = acc[o][i];
#pragma unroll
      for (int sh = 8; sh < 64; sh <<= 1) {
        p.x += __shfl_xor(p.x, sh, 64); p.y += __shfl_xor(p.y, sh, 64); p.z += __shfl_xor(p.z, sh, 64); p.w += __shfl_xor(p.w, sh, 64);
      }
      if ((t & 63) < 8) *reinterpret_cast<float4*>(&red[wave][o][32 * part + 4 * i]) = p;
    }
    float sb = part == 0 ? bacc[o] : 0.f;
#pragma unroll
    for (int sh = 8; sh < 64; sh <<= 1) sb += __shfl_xor(sb, sh, 64);
    if ((t & 63) == 0) red[wave][o][256] = sb;
  }
  __syncthreads();
  for (int o = 0; o < nd; ++o) {
    if (a.dw2) atomicAdd(a.dw2 + prow + (long)o * 256 + t, red[0][o][t] + red[1][o][t] + red[2][o][t] + red[3][o][t]);
    if (a.db2 && t == 0) atomicAdd(a.db2 + prow + o, red[0][o][256] + red[1][o][256] + red[2][o][256] + red[3][o][256]);
  }
}
}  // namespace

static int task_heads_check(const float* h, const float* w2, long param_gs, long head_stride, int rows, int groups, int heads, const int* dims,
                            const int* cols, int ld_out, const char* who) {
  PETR_CHECK(h && w2 && rows > 0 && groups > 0 && heads > 0 && heads <= 8 && ld_out > 0, PETR_ERR_INVALID, "%s: bad arguments", who);
  PETR_CHECK(aligned16(h) && aligned16(w2) && !(param_gs & 3) && !(head_stride & 3), PETR_ERR_INVALID,
             "%s: h / w2 must be 16-byte aligned, parameter strides multiples of 4", who);
  for (int t = 0; t < heads; ++t)
    PETR_CHECK(dims[t] >= 1 && dims[t] <= TH_MAXD && cols[t] >= 0 && cols[t] + dims[t] <= ld_out, PETR_ERR_UNSUPPORTED,
               "%s: head %d: dims=%d (1..%d), cols=%d", who, t, dims[t], TH_MAXD, cols[t]);
  PETR_CHECK((long)groups * heads <= 65535, PETR_ERR_UNSUPPORTED, "%s: too many (group, head) pairs", who);
  return PETR_OK;
}

extern "C" int petr_task_heads_fwd(const petr_task_heads_fwd_args* ap, void* stream) {
  PETR_CHECK(ap && ap->out, PETR_ERR_INVALID, "task_heads_fwd: null pointer");
  const petr_task_heads_fwd_args& a = *ap;
  if (int rc = task_heads_check(a.h, a.w2, a.param_gs, a.head_stride, a.rows, a.groups, a.heads, a.dims, a.cols, a.ld_out, "task_heads_fwd")) return rc;
  hipLaunchKernelGGL(task_heads_fwd_kernel, dim3((unsigned)cdiv(a.rows, 32), (unsigned)(a.groups * a.heads)), dim3(256), 0,
                     (hipStream_t)stream, a);
  PETR_LAUNCH_CHECK("task_heads_fwd");
  return PETR_OK;
}

extern "C" int petr_task_heads_bwd(const petr_task_heads_bwd_args* ap, void* stream) {
  PETR_CHECK(ap && ap->d_out && (ap->d_h || ap->dw2), PETR_ERR_INVALID, "task_heads_bwd: null pointer");
  const petr_task_heads_bwd_args& a = *ap;
  if (int rc = task_heads_check(a.h, a.w2, a.param_gs, a.head_stride, a.rows, a.groups, a.heads, a.dims, a.cols, a.ld_out, "task_heads_bwd")) return rc;
  PETR_CHECK(!a.d_h || aligned16(a.d_h), PETR_ERR_INVALID, "task_heads_bwd: d_h must be 16-byte aligned");
  // row-block chunks per (group, head): enough workgroups to fill the chip, few adders per parameter
  const long nrb = cdiv(a.rows, 32);
  long chunks = cdiv(512, (long)a.groups * a.heads);
  if (chunks > nrb) chunks = nrb;
  hipLaunchKernelGGL(task_heads_bwd_kernel, dim3((unsigned)chunks, (unsigned)(a.groups * a.heads)), dim3(256), 0, (hipStream_t)stream, a);
  PETR_LAUNCH_CHECK("task_heads_bwd");
  return PETR_OK;
}

extern "C" int petr_branch_bwd(const petr_branch_bwd_args* ap, void* stream) {
  PETR_CHECK(ap && ap->y2 && ap->w2 && ap->y1 && ap->w1 && ap->d_x && ap->rows > 0 && ap->groups > 0 &&
                 ((ap->w3 && ap->d_out) || (!ap->w3 && ap->d_y2)),
             PETR_ERR_INVALID, "branch_bwd: bad arguments");
  const petr_branch_bwd_args& a = *ap;
  PETR_CHECK(!a.w3 || (a.n_out > 0 && a.n_out <= 16), PETR_ERR_UNSUPPORTED, "branch_bwd: n_out=%d outside 1..16", a.n_out);
  PETR_CHECK((!a.g2 || (a.h2 && a.mean2 && a.rstd2)) && (!a.g1 || (a.h1 && a.mean1 && a.rstd1)), PETR_ERR_INVALID,
             "branch_bwd: a LayerNorm stage needs its pre-norm rows and statistics");
  PETR_CHECK(aligned16(a.w2) && aligned16(a.w1) && aligned16(a.y2) && aligned16(a.y1) && aligned16(a.h2) && aligned16(a.h1) &&
                 aligned16(a.d_h2) && aligned16(a.d_h1) && aligned16(a.d_x) && aligned16(a.d_y2) && (!a.w3 || aligned16(a.w3)) &&
                 !(a.param_gs & 3),
             PETR_ERR_INVALID, "branch_bwd: every [.., 256] operand must be 16-byte aligned (group stride a multiple of 4)");
  BbParams p;
  p.a = a;
  p.nrb = (int)cdiv(a.rows, BR_ROWS);
  hipLaunchKernelGGL(branch_bwd_kernel, dim3((unsigned)(p.nrb * a.groups)), dim3(512), 0, (hipStream_t)stream, p);
  PETR_LAUNCH_CHECK("branch_bwd");
  return PETR_OK;
}

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

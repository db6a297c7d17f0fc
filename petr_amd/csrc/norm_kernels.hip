// Row-wise / elementwise kernels of the PETRHead path (all HBM- or latency-bound):
// LayerNorm forward/backward with fused partial-sum + bias + residual prologue, column sums for
// bias gradients, the box-decoding epilogue and its gradient, and small reducers.
#include "common.h"

namespace {

constexpr int LN_MAXV = 4;  // float4 chunks per lane: C <= 1024

// ------------------------------------------------------------------------------------------
// LayerNorm forward: one wave per row, row kept in registers (two-pass mean / variance).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float nan_to_num_f(float v) {
  if (v != v) return 0.f;
  if (v == INFINITY) return 3.402823466e+38f;
  if (v == -INFINITY) return -3.402823466e+38f;
  return v;
}

__global__ __launch_bounds__(256) void layernorm_fwd_kernel(petr_layernorm_args a) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.M) return;
  const int nv = a.C >> 2;
  DropDev dd;
  __builtin_memcpy(&dd, &a.drop, sizeof dd);   // petr_layernorm_fwd() stored the derived keys here
  const uint32_t rk = dd.thr ? drop_row_key(dd, (uint32_t)row) : 0u;
  float4 v[LN_MAXV];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int i4 = lane + 64 * j;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < nv) {
      x = reinterpret_cast<const float4*>(a.x + (size_t)row * a.C)[i4];
      for (int p = 1; p < a.n_partials; ++p) {
        const float4 y = reinterpret_cast<const float4*>(a.x + (size_t)p * a.partial_stride + (size_t)row * a.C)[i4];
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      }
      if (a.bias) {
        const float4 y = reinterpret_cast<const float4*>(a.bias)[i4];
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      }
      if (dd.thr) {   // dropout of the sub-layer output, before the identity is added
        x.x = drop_keep(rk, 4 * i4, dd.thr) ? x.x * dd.scale : 0.f;
        x.y = drop_keep(rk, 4 * i4 + 1, dd.thr) ? x.y * dd.scale : 0.f;
        x.z = drop_keep(rk, 4 * i4 + 2, dd.thr) ? x.z * dd.scale : 0.f;
        x.w = drop_keep(rk, 4 * i4 + 3, dd.thr) ? x.w * dd.scale : 0.f;
      }
      if (a.residual) {
        const float4 y = reinterpret_cast<const float4*>(a.residual + (size_t)row * a.C)[i4];
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      }
      if (a.z_out) reinterpret_cast<float4*>(a.z_out + (size_t)row * a.C)[i4] = x;
      sum += (x.x + x.y) + (x.z + x.w);
    }
    v[j] = x;
  }
  const float mean = wave_sum(sum) / (float)a.C;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    if (lane + 64 * j < nv) {
      const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
      sq += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float var = wave_sum(sq) / (float)a.C;
  const float rstd = 1.f / sqrtf(var + a.eps);
  if (lane == 0) {
    if (a.mean) a.mean[row] = mean;
    if (a.rstd) a.rstd[row] = rstd;
  }
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int i4 = lane + 64 * j;
    if (i4 < nv) {
      const float4 gm = reinterpret_cast<const float4*>(a.gamma)[i4];
      const float4 bt = reinterpret_cast<const float4*>(a.beta)[i4];
      float4 y;
      y.x = (v[j].x - mean) * rstd * gm.x + bt.x;
      y.y = (v[j].y - mean) * rstd * gm.y + bt.y;
      y.z = (v[j].z - mean) * rstd * gm.z + bt.z;
      y.w = (v[j].w - mean) * rstd * gm.w + bt.w;
      if (a.flags & PETR_LN_RELU) {
        y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
      }
      if (a.flags & PETR_LN_NAN_TO_NUM) {
        y.x = nan_to_num_f(y.x); y.y = nan_to_num_f(y.y); y.z = nan_to_num_f(y.z); y.w = nan_to_num_f(y.w);
      }
      reinterpret_cast<float4*>(a.y + (size_t)row * a.C)[i4] = y;
      if (a.y2) {
        const int r2 = a.add2_rows > 0 ? row % a.add2_rows : row;
        const float4 e = reinterpret_cast<const float4*>(a.add2 + (size_t)r2 * a.C)[i4];
        reinterpret_cast<float4*>(a.y2 + (size_t)row * a.C)[i4] = make_float4(y.x + e.x, y.y + e.y, y.z + e.z, y.w + e.w);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward.  Each block walks rows block-stride, every lane owns fixed columns, so
// dgamma/dbeta accumulate in registers; per-block partials -> ws, reduced by a second kernel
// (deterministic, no atomics).
// ------------------------------------------------------------------------------------------
constexpr int LNB_BLOCKS = 256;

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(petr_layernorm_bwd_args a, int nblocks) {
  __shared__ float red[2][4][LN_MAXV * 256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = a.C >> 2;
  float4 dg[LN_MAXV], db[LN_MAXV];
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) dg[j] = db[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  DropDev dd;
  __builtin_memcpy(&dd, &a.drop, sizeof dd);   // petr_layernorm_bwd() stored the derived keys here
  for (int row = blockIdx.x * 4 + wave; row < a.M; row += nblocks * 4) {
    const float mean = a.mean[row], rstd = a.rstd[row];
    float4 xh[LN_MAXV], g[LN_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int i4 = lane + 64 * j;
      xh[j] = g[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i4 < nv) {
        const float4 z = reinterpret_cast<const float4*>(a.z + (size_t)row * a.C)[i4];
        float4 dy = reinterpret_cast<const float4*>(a.dy + (size_t)row * a.C)[i4];
        for (int p = 1; p < a.dy_partials; ++p) {   // split-K slabs of the producing contraction
          const float4 t = reinterpret_cast<const float4*>(a.dy + (size_t)p * a.dy_partial_stride + (size_t)row * a.C)[i4];
          dy.x += t.x; dy.y += t.y; dy.z += t.z; dy.w += t.w;
        }
        if (a.dy_residual) {
          const float4 t = reinterpret_cast<const float4*>(a.dy_residual + (size_t)row * a.C)[i4];
          dy.x += t.x; dy.y += t.y; dy.z += t.z; dy.w += t.w;
        }
        if (a.flags & PETR_LN_RELU) {
          const float4 y = reinterpret_cast<const float4*>(a.y + (size_t)row * a.C)[i4];
          dy.x = y.x > 0.f ? dy.x : 0.f; dy.y = y.y > 0.f ? dy.y : 0.f;
          dy.z = y.z > 0.f ? dy.z : 0.f; dy.w = y.w > 0.f ? dy.w : 0.f;
        }
        const float4 gm = reinterpret_cast<const float4*>(a.gamma)[i4];
        xh[j] = make_float4((z.x - mean) * rstd, (z.y - mean) * rstd, (z.z - mean) * rstd, (z.w - mean) * rstd);
        g[j] = make_float4(dy.x * gm.x, dy.y * gm.y, dy.z * gm.z, dy.w * gm.w);
        s1 += (g[j].x + g[j].y) + (g[j].z + g[j].w);
        s2 += (g[j].x * xh[j].x + g[j].y * xh[j].y) + (g[j].z * xh[j].z + g[j].w * xh[j].w);
        dg[j].x += dy.x * xh[j].x; dg[j].y += dy.y * xh[j].y; dg[j].z += dy.z * xh[j].z; dg[j].w += dy.w * xh[j].w;
        db[j].x += dy.x; db[j].y += dy.y; db[j].z += dy.z; db[j].w += dy.w;
      }
    }
    const float m1 = wave_sum(s1) / (float)a.C, m2 = wave_sum(s2) / (float)a.C;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int i4 = lane + 64 * j;
      if (i4 < nv) {
        float4 d;
        d.x = rstd * (g[j].x - m1 - xh[j].x * m2);
        d.y = rstd * (g[j].y - m1 - xh[j].y * m2);
        d.z = rstd * (g[j].z - m1 - xh[j].z * m2);
        d.w = rstd * (g[j].w - m1 - xh[j].w * m2);
        float4* dst = reinterpret_cast<float4*>(a.dz + (size_t)row * a.C) + i4;
        if (a.dz_accumulate) { const float4 o = *dst; d.x += o.x; d.y += o.y; d.z += o.z; d.w += o.w; }
        *dst = d;
        if (a.dz_drop) {   // gradient of the dropped branch of z = drop(f) + residual
          const uint32_t rk = drop_row_key(dd, (uint32_t)row);
          float4 e;
          e.x = drop_keep(rk, 4 * i4, dd.thr) ? d.x * dd.scale : 0.f;
          e.y = drop_keep(rk, 4 * i4 + 1, dd.thr) ? d.y * dd.scale : 0.f;
          e.z = drop_keep(rk, 4 * i4 + 2, dd.thr) ? d.z * dd.scale : 0.f;
          e.w = drop_keep(rk, 4 * i4 + 3, dd.thr) ? d.w * dd.scale : 0.f;
          reinterpret_cast<float4*>(a.dz_drop + (size_t)row * a.C)[i4] = e;
        }
      }
    }
  }
  // cross-wave reduction of the column partials
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int i4 = lane + 64 * j;
    reinterpret_cast<float4*>(red[0][wave])[i4] = dg[j];
    reinterpret_cast<float4*>(red[1][wave])[i4] = db[j];
  }
  __syncthreads();
  float* ws = reinterpret_cast<float*>(a.ws);
  for (int cidx = threadIdx.x; cidx < a.C; cidx += 256) {
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { sg += red[0][w][cidx]; sb += red[1][w][cidx]; }
    if (ws) {
      ws[(size_t)blockIdx.x * 2 * a.C + cidx] = sg;
      ws[(size_t)blockIdx.x * 2 * a.C + a.C + cidx] = sb;
    } else {
      if (a.dgamma) atomicAdd(a.dgamma + cidx, sg);
      if (a.dbeta) atomicAdd(a.dbeta + cidx, sb);
    }
  }
}

__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* ws, int nblocks, int C, float* dgamma,
                                                                    float* dbeta) {
  const int cidx = blockIdx.x * blockDim.x + threadIdx.x;
  if (cidx >= 2 * C) return;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += ws[(size_t)b * 2 * C + cidx];
  if (cidx < C) { if (dgamma) dgamma[cidx] += s; }
  else { if (dbeta) dbeta[cidx - C] += s; }
}

// ------------------------------------------------------------------------------------------
// Column sums (bias gradients): 32 row-chunks x 64-column stripes -> partials -> final add.
// ------------------------------------------------------------------------------------------
constexpr int CS_CHUNKS = 32;

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long ld, int M, int N, float* ws) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int per = (M + CS_CHUNKS - 1) / CS_CHUNKS;
  const int r0 = blockIdx.y * per, r1 = min(M, r0 + per);
  float s = 0.f;
  if (n < N)
    for (int r = r0 + wave; r < r1; r += 4) s += x[(size_t)r * ld + n];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && n < N) ws[(size_t)blockIdx.y * N + n] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

__global__ __launch_bounds__(256) void colsum_final_kernel(const float* ws, int N, float* out, int accumulate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int c = 0; c < CS_CHUNKS; ++c) s += ws[(size_t)c * N + n];
  out[n] = accumulate ? out[n] + s : s;
}

// ------------------------------------------------------------------------------------------
// Box epilogue (petr_head.py:441-460)
// ------------------------------------------------------------------------------------------
struct BboxParams {
  const float* reg; const float* ref; float* out;
  int rows, Q, code;
  float lo[3], span[3];
  float time_div, eps;
};

__device__ __forceinline__ float inv_sigmoid_dev(float x, float eps) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  return logf(fmaxf(x, eps) / fmaxf(1.f - x, eps));
}

__global__ __launch_bounds__(256) void bbox_fwd_kernel(BboxParams p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)p.rows * p.code) return;
  const int row = (int)(idx / p.code), ch = (int)(idx - (long)row * p.code);
  const int q = row % p.Q;
  float t = p.reg[idx];
  const int ax = ch == 0 ? 0 : (ch == 1 ? 1 : (ch == 4 ? 2 : -1));
  if (ax >= 0) {
    t += inv_sigmoid_dev(p.ref[q * 3 + ax], p.eps);
    t = 1.f / (1.f + expf(-t));
    t = t * p.span[ax] + p.lo[ax];
  } else if (ch >= 8 && p.time_div != 0.f) {
    t = t / p.time_div;
  }
  p.out[idx] = t;
}

// thread per (q, ch): loops over the rows/Q (level, batch) copies so that dref needs no atomics
__global__ __launch_bounds__(256) void bbox_bwd_kernel(BboxParams p, const float* dout, float* dreg, float* dref) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= p.Q * p.code) return;
  const int q = idx / p.code, ch = idx - q * p.code;
  const int ax = ch == 0 ? 0 : (ch == 1 ? 1 : (ch == 4 ? 2 : -1));
  const int copies = p.rows / p.Q;
  float racc = 0.f;
  for (int cpy = 0; cpy < copies; ++cpy) {
    const long off = ((long)cpy * p.Q + q) * p.code + ch;
    float g = dout[off];
    if (ax >= 0) {
      const float s = (p.out[off] - p.lo[ax]) / p.span[ax];
      g = g * p.span[ax] * s * (1.f - s);
      racc += g;
    } else if (ch >= 8 && p.time_div != 0.f) {
      g = g / p.time_div;
    }
    dreg[off] = g;
  }
  if (ax >= 0 && dref) {
    const float r = p.ref[q * 3 + ax];
    // d/dr log(clamp(r,eps)/clamp(1-r,eps)) on the un-clamped interior; 0 where a clamp is active
    float d = 0.f;
    if (r > 0.f && r < 1.f) {
      if (r > p.eps) d += 1.f / r;
      if (1.f - r > p.eps) d += 1.f / (1.f - r);
    }
    dref[q * 3 + ax] += racc * d;
  }
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, float v, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ __launch_bounds__(256) void axpy_kernel(float* y, const float* x, float alpha, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += alpha * x[i];
}
__global__ __launch_bounds__(256) void add_rows_kernel(const float4* x, const float4* e, float4* out, long n4, long e_n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 a = x[i], b = e[i % e_n4];
  out[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
// the same sum with a bf16 x and a bf16 result (bf16 mode: memory is stored as bf16 by input_proj, key = memory + pos
// is only ever read by bf16 contractions); 8 elements per thread
__global__ __launch_bounds__(256) void add_rows_bf16_kernel(const uint4* x, const float4* e, uint4* out, long n8, long e_n8) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const uint4 a = x[i];
  const long j = i % e_n8;
  const float4 b0 = e[2 * j], b1 = e[2 * j + 1];
  auto lo = [](uint32_t w) { return __uint_as_float(w << 16); };
  auto hi = [](uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); };
  auto pk = [](float u, float v) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    b2 o = {(__bf16)u, (__bf16)v};
    return __builtin_bit_cast(uint32_t, o);
  };
  out[i] = make_uint4(pk(lo(a.x) + b0.x, hi(a.x) + b0.y), pk(lo(a.y) + b0.z, hi(a.y) + b0.w),
                      pk(lo(a.z) + b1.x, hi(a.z) + b1.y), pk(lo(a.w) + b1.z, hi(a.w) + b1.w));
}
// SELayer gate of PETRv2 (petrv2_head.py:55-60): out = x * sigmoid(u)
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* x, const float* u, float* out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[i] * (1.f / (1.f + expf(-u[i])));
}
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* dout, const float* x, const float* u, float* dx, float* du,
                                                        long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float sg = 1.f / (1.f + expf(-u[i]));
  const float g = dout[i];
  dx[i] = g * sg;
  du[i] = g * x[i] * sg * (1.f - sg);
}
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* x, int np, long stride, const float* bias,
                                                               const float* residual, float* out, long M, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C) return;
  float s = x[i];
  for (int p = 1; p < np; ++p) s += x[(size_t)p * stride + i];
  if (bias) s += bias[i % C];
  if (residual) s += residual[i];
  out[i] = s;
}
__global__ __launch_bounds__(256) void reduce_batch_kernel(const float* x, int B, long n, float* out, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += x[(size_t)b * n + i];
  out[i] = accumulate ? out[i] + s : s;
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void dropout_mask_kernel(DropDev d, long rows, long cols, uint8_t* keep) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const long row = i / cols, col = i - row * cols;
  keep[i] = d.thr == 0u || drop_keep(drop_row_key(d, (uint32_t)row), (uint32_t)col, d.thr) ? 1 : 0;
}
}  // namespace

namespace {
// Packed attention-dropout masks (see petr_dropout_bits in petr_hip.h).  One wave = one 32 query x 32 key tile: lane
// (q = lane & 31, half = lane >> 5) evaluates the 16 keys 16 half .. + 15 of its row with eight pair hashes (the same
// function, hence the same mask, as drop_keep); the per-key ballots are the key-major words, the per-lane bit strings
// the query-major ones.
__global__ __launch_bounds__(256) void dropout_bits_kernel(DropDev d, int BH, int Q, int L, int nqt, int nkb, uint32_t* bits_q,
                                                           uint32_t* bits_k) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kb_groups = (nkb + 3) >> 2;
  int w = blockIdx.x;
  const int kbg = w % kb_groups;
  w /= kb_groups;
  const int qt = w % nqt, bh = w / nqt;
  const int kb = 4 * kbg + wave;
  if (kb >= nkb) return;                                  // wave-uniform
  const int q = 32 * qt + (lane & 31), half = lane >> 5;
  const uint32_t rk = drop_row_key(d, (uint32_t)(bh * Q + min(q, Q - 1)));
  uint32_t mine = 0u, colw = 0u;                            // this lane's 16 key bits; (lanes 0..31) the word of key `lane`
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const uint32_t col = (uint32_t)(32 * kb + 16 * half + 2 * j);
    const uint32_t hsh = drop_pair_hash(rk, col >> 1);
    const bool k0 = (hsh & 0xFFFFu) >= d.thr, k1 = (hsh >> 16) >= d.thr;
    mine |= (k0 ? 1u : 0u) << (2 * j) | (k1 ? 1u : 0u) << (2 * j + 1);
    const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1);       // low word: half 0 (key 2j), high word: half 1 (16 + 2j)
    colw = (lane == 2 * j) ? (uint32_t)b0 : colw;
    colw = (lane == 16 + 2 * j) ? (uint32_t)(b0 >> 32) : colw;
    colw = (lane == 2 * j + 1) ? (uint32_t)b1 : colw;
    colw = (lane == 16 + 2 * j + 1) ? (uint32_t)(b1 >> 32) : colw;
  }
  const uint32_t other = __shfl_xor(mine, 32, 64);
  if (lane < 32) {
    const long tile = ((long)bh * nkb + kb) * (32L * nqt) + q;                 // query-major: [bh][kb][q]
    if (bits_q) bits_q[tile] = mine | (other << 16);
    if (bits_k) bits_k[((long)bh * nqt + qt) * (32L * nkb) + 32 * kb + petr_bits_slot(lane)] = colw;   // key-major: [bh][qt][block][slot]
  }
}
}  // namespace

extern "C" size_t petr_dropout_bits_words(int BH, int Q, int L) {
  return (size_t)BH * (size_t)cdiv(L, 32) * 32 * (size_t)cdiv(Q, 32);
}

extern "C" int petr_dropout_bits(const petr_dropout* d, int BH, int Q, int L, uint32_t* bits_q, uint32_t* bits_k, void* stream) {
  PETR_CHECK(d && (bits_q || bits_k) && BH > 0 && Q > 0 && L > 0, PETR_ERR_INVALID, "dropout_bits: bad arguments");
  PETR_CHECK(d->p > 0.f && d->p < 1.f, PETR_ERR_INVALID, "dropout_bits: p=%g outside (0,1)", (double)d->p);
  PETR_CHECK((long)BH * Q < (1L << 32), PETR_ERR_UNSUPPORTED, "dropout_bits: row index needs B*H*Q < 2^32");
  const int nqt = (int)cdiv(Q, 32), nkb = (int)cdiv(L, 32);
  const long blocks = (long)BH * nqt * cdiv(nkb, 4);
  PETR_CHECK(blocks < (1L << 31), PETR_ERR_UNSUPPORTED, "dropout_bits: grid too large");
  hipLaunchKernelGGL(dropout_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, make_drop(*d), BH, Q, L, nqt,
                     nkb, bits_q, bits_k);
  PETR_LAUNCH_CHECK("dropout_bits");
  return PETR_OK;
}

extern "C" int petr_dropout_mask(const petr_dropout* d, long rows, long cols, uint8_t* keep, void* stream) {
  PETR_CHECK(d && keep && rows > 0 && cols > 0 && rows < (1L << 32) && cols < (1L << 32), PETR_ERR_INVALID,
             "dropout_mask: bad arguments");
  PETR_CHECK(d->p >= 0.f && d->p < 1.f, PETR_ERR_INVALID, "dropout_mask: p=%g outside [0,1)", (double)d->p);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)cdiv(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream,
                     make_drop(*d), rows, cols, keep);
  PETR_LAUNCH_CHECK("dropout_mask");
  return PETR_OK;
}

extern "C" int petr_layernorm_fwd(const petr_layernorm_args* a, void* stream) {
  PETR_CHECK(a && a->x && a->gamma && a->beta && a->y, PETR_ERR_INVALID, "layernorm: null pointer");
  PETR_CHECK(a->M > 0 && a->C > 0 && (a->C & 3) == 0 && a->C <= 256 * LN_MAXV, PETR_ERR_UNSUPPORTED,
             "layernorm: C=%d must be a multiple of 4 and <= 1024", a->C);
  PETR_CHECK(aligned16(a->x) && aligned16(a->y) && aligned16(a->gamma) && aligned16(a->beta) &&
                 (!a->bias || aligned16(a->bias)) && (!a->residual || aligned16(a->residual)) &&
                 (!a->z_out || aligned16(a->z_out)) && (a->partial_stride & 3) == 0 &&
                 (!a->y2 || (a->add2 && aligned16(a->y2) && aligned16(a->add2))),
             PETR_ERR_INVALID, "layernorm: pointers must be 16-byte aligned");
  petr_layernorm_args p = *a;
  if (p.n_partials <= 0) p.n_partials = 1;
  {
    static_assert(sizeof(petr_dropout) == sizeof(DropDev), "petr_dropout / DropDev must have the same size");
    PETR_CHECK(a->drop.p >= 0.f && a->drop.p < 1.f, PETR_ERR_INVALID, "layernorm: dropout p=%g outside [0,1)", (double)a->drop.p);
    const DropDev dd = make_drop(a->drop);   // the kernel reads p.drop as a DropDev (same 16 bytes)
    memcpy(&p.drop, &dd, sizeof dd);
  }
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)cdiv(a->M, 4)), dim3(256), 0, (hipStream_t)stream, p);
  PETR_LAUNCH_CHECK("layernorm_fwd");
  return PETR_OK;
}

extern "C" size_t petr_layernorm_bwd_workspace_bytes(int M, int C) {
  (void)M;
  return (size_t)LNB_BLOCKS * 2 * C * sizeof(float);
}

extern "C" int petr_layernorm_bwd(const petr_layernorm_bwd_args* a, void* stream) {
  PETR_CHECK(a && a->z && a->mean && a->rstd && a->gamma && a->dy && a->dz, PETR_ERR_INVALID,
             "layernorm_bwd: null pointer");
  PETR_CHECK(a->M > 0 && a->C > 0 && (a->C & 3) == 0 && a->C <= 256 * LN_MAXV, PETR_ERR_UNSUPPORTED,
             "layernorm_bwd: C=%d unsupported", a->C);
  PETR_CHECK(!(a->flags & PETR_LN_RELU) || a->y, PETR_ERR_INVALID, "layernorm_bwd: ReLU flag needs y");
  int nblocks = (int)((cdiv(a->M, 4) < LNB_BLOCKS) ? cdiv(a->M, 4) : LNB_BLOCKS);
  if (!a->ws && nblocks > 128) nblocks = 128;   // atomics form: fewer, longer blocks -> fewer atomic adds (64 / 128 / 256 A/B: 128 best)
  hipStream_t s = (hipStream_t)stream;
  petr_layernorm_bwd_args p = *a;
  {
    PETR_CHECK(!a->dz_drop || (a->drop.p > 0.f && a->drop.p < 1.f && aligned16(a->dz_drop)), PETR_ERR_INVALID,
               "layernorm_bwd: dz_drop needs 0 < p < 1 and a 16-byte aligned pointer");
    const DropDev dd = make_drop(a->drop);
    memcpy(&p.drop, &dd, sizeof dd);
  }
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblocks), dim3(256), 0, s, p, nblocks);
  PETR_LAUNCH_CHECK("layernorm_bwd");
  if (a->ws && (a->dgamma || a->dbeta)) {
    hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3((unsigned)cdiv(2 * a->C, 256)), dim3(256), 0, s,
                       (const float*)a->ws, nblocks, a->C, a->dgamma, a->dbeta);
    PETR_LAUNCH_CHECK("layernorm_bwd_reduce");
  }
  return PETR_OK;
}

extern "C" size_t petr_colsum_workspace_bytes(int N) { return (size_t)CS_CHUNKS * N * sizeof(float); }

extern "C" int petr_colsum(const float* x, long ld, int M, int N, float* out, int accumulate, void* ws, void* stream) {
  PETR_CHECK(x && out && ws && M > 0 && N > 0, PETR_ERR_INVALID, "colsum: bad argument");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)cdiv(N, 64), CS_CHUNKS), dim3(256), 0, s, x, ld, M, N, (float*)ws);
  PETR_LAUNCH_CHECK("colsum");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, s, (const float*)ws, N, out,
                     accumulate);
  PETR_LAUNCH_CHECK("colsum_final");
  return PETR_OK;
}

static BboxParams make_bbox(const petr_bbox_args* a) {
  BboxParams p;
  p.reg = a->reg; p.ref = a->ref; p.out = a->out;
  p.rows = a->rows; p.Q = a->Q; p.code = a->code;
  for (int i = 0; i < 3; ++i) {
    p.lo[i] = a->pc_range[i];
    p.span[i] = (float)((double)a->pc_range[i + 3] - (double)a->pc_range[i]);
  }
  p.time_div = a->time_div;
  p.eps = a->eps;
  return p;
}

extern "C" int petr_bbox_epilogue_fwd(const petr_bbox_args* a, void* stream) {
  PETR_CHECK(a && a->reg && a->ref && a->out, PETR_ERR_INVALID, "bbox: null pointer");
  PETR_CHECK(a->rows > 0 && a->Q > 0 && a->rows % a->Q == 0 && a->code >= 5, PETR_ERR_INVALID, "bbox: bad shape");
  BboxParams p = make_bbox(a);
  hipLaunchKernelGGL(bbox_fwd_kernel, dim3((unsigned)cdiv((long)a->rows * a->code, 256)), dim3(256), 0,
                     (hipStream_t)stream, p);
  PETR_LAUNCH_CHECK("bbox_fwd");
  return PETR_OK;
}

extern "C" int petr_bbox_epilogue_bwd(const petr_bbox_args* a, const float* dout, float* dreg, float* dref, void* stream) {
  PETR_CHECK(a && a->out && a->ref && dout && dreg, PETR_ERR_INVALID, "bbox_bwd: null pointer");
  PETR_CHECK(a->rows > 0 && a->Q > 0 && a->rows % a->Q == 0 && a->code >= 5, PETR_ERR_INVALID, "bbox_bwd: bad shape");
  BboxParams p = make_bbox(a);
  hipLaunchKernelGGL(bbox_bwd_kernel, dim3((unsigned)cdiv((long)a->Q * a->code, 256)), dim3(256), 0,
                     (hipStream_t)stream, p, dout, dreg, dref);
  PETR_LAUNCH_CHECK("bbox_bwd");
  return PETR_OK;
}

extern "C" int petr_fill(float* p, float v, long n, void* stream) {
  PETR_CHECK(p && n >= 0, PETR_ERR_INVALID, "fill: bad argument");
  if (n == 0) return PETR_OK;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, v, n);
  PETR_LAUNCH_CHECK("fill");
  return PETR_OK;
}

extern "C" int petr_axpy(float* y, const float* x, float alpha, long n, void* stream) {
  PETR_CHECK(y && x && n >= 0, PETR_ERR_INVALID, "axpy: bad argument");
  if (n == 0) return PETR_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, y, x, alpha, n);
  PETR_LAUNCH_CHECK("axpy");
  return PETR_OK;
}

extern "C" int petr_add_rows(const float* x, const float* e, float* out, long M, int e_rows, int C, void* stream) {
  PETR_CHECK(x && e && out && M > 0 && C > 0 && (C & 3) == 0 && aligned16(x) && aligned16(e) && aligned16(out),
             PETR_ERR_INVALID, "add_rows: bad argument");
  const long n4 = M * C / 4, e_n4 = (long)(e_rows > 0 ? e_rows : M) * C / 4;
  hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                     (const float4*)e, (float4*)out, n4, e_n4);
  PETR_LAUNCH_CHECK("add_rows");
  return PETR_OK;
}

extern "C" int petr_add_rows_bf16(const uint16_t* x, const float* e, uint16_t* out, long M, int e_rows, int C, void* stream) {
  PETR_CHECK(x && e && out && M > 0 && C > 0 && (C & 7) == 0 && aligned16(x) && aligned16(e) && aligned16(out),
             PETR_ERR_INVALID, "add_rows_bf16: bad argument");
  const long n8 = M * C / 8, e_n8 = (long)(e_rows > 0 ? e_rows : M) * C / 8;
  hipLaunchKernelGGL(add_rows_bf16_kernel, dim3((unsigned)cdiv(n8, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x,
                     (const float4*)e, (uint4*)out, n8, e_n8);
  PETR_LAUNCH_CHECK("add_rows_bf16");
  return PETR_OK;
}

// key = memory + (pos_a + pos_b) with the sum pos_a += pos_b left in place: the two halves of the key position embedding
// (3D position encoder, adapt_pos3d) come from different streams and are only joined here (head.hip forward)
__global__ __launch_bounds__(256) void add_rows2_kernel(const float4* x, float4* e1, const float4* e2, float4* out, long n4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 a = x[i], b = e1[i], c = e2[i];
  const float4 pe = make_float4(b.x + c.x, b.y + c.y, b.z + c.z, b.w + c.w);
  e1[i] = pe;
  out[i] = make_float4(a.x + pe.x, a.y + pe.y, a.z + pe.z, a.w + pe.w);
}
__global__ __launch_bounds__(256) void add_rows2_bf16_kernel(const uint4* x, float4* e1, const float4* e2, uint4* out, long n8) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const uint4 a = x[i];
  float4 b0 = e1[2 * i], b1 = e1[2 * i + 1];
  const float4 c0 = e2[2 * i], c1 = e2[2 * i + 1];
  b0 = make_float4(b0.x + c0.x, b0.y + c0.y, b0.z + c0.z, b0.w + c0.w);
  b1 = make_float4(b1.x + c1.x, b1.y + c1.y, b1.z + c1.z, b1.w + c1.w);
  e1[2 * i] = b0;
  e1[2 * i + 1] = b1;
  auto lo = [](uint32_t w) { return __uint_as_float(w << 16); };
  auto hi = [](uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); };
  auto pk = [](float u, float v) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 o = {(__bf16)u, (__bf16)v};
    return __builtin_bit_cast(uint32_t, o);
  };
  uint4 o;
  o.x = pk(lo(a.x) + b0.x, hi(a.x) + b0.y);
  o.y = pk(lo(a.y) + b0.z, hi(a.y) + b0.w);
  o.z = pk(lo(a.z) + b1.x, hi(a.z) + b1.y);
  o.w = pk(lo(a.w) + b1.z, hi(a.w) + b1.w);
  out[i] = o;
}

extern "C" int petr_add_rows2(const float* x, float* e1, const float* e2, float* out, long M, int C, void* stream) {
  PETR_CHECK(x && e1 && e2 && out && M > 0 && C > 0 && (C & 3) == 0 && aligned16(x) && aligned16(e1) && aligned16(e2) && aligned16(out),
             PETR_ERR_INVALID, "add_rows2: bad argument");
  const long n4 = M * C / 4;
  hipLaunchKernelGGL(add_rows2_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (float4*)e1,
                     (const float4*)e2, (float4*)out, n4);
  PETR_LAUNCH_CHECK("add_rows2");
  return PETR_OK;
}

extern "C" int petr_add_rows2_bf16(const uint16_t* x, float* e1, const float* e2, uint16_t* out, long M, int C, void* stream) {
  PETR_CHECK(x && e1 && e2 && out && M > 0 && C > 0 && (C & 7) == 0 && aligned16(x) && aligned16(e1) && aligned16(e2) && aligned16(out),
             PETR_ERR_INVALID, "add_rows2_bf16: bad argument");
  const long n8 = M * C / 8;
  hipLaunchKernelGGL(add_rows2_bf16_kernel, dim3((unsigned)cdiv(n8, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (float4*)e1,
                     (const float4*)e2, (uint4*)out, n8);
  PETR_LAUNCH_CHECK("add_rows2_bf16");
  return PETR_OK;
}

extern "C" int petr_gate_fwd(const float* x, const float* u, float* out, long n, void* stream) {
  PETR_CHECK(x && u && out && n > 0, PETR_ERR_INVALID, "gate_fwd: bad argument");
  hipLaunchKernelGGL(gate_fwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, u, out, n);
  PETR_LAUNCH_CHECK("gate_fwd");
  return PETR_OK;
}

extern "C" int petr_gate_bwd(const float* dout, const float* x, const float* u, float* dx, float* du, long n, void* stream) {
  PETR_CHECK(dout && x && u && dx && du && n > 0, PETR_ERR_INVALID, "gate_bwd: bad argument");
  hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dout, x, u, dx, du, n);
  PETR_LAUNCH_CHECK("gate_bwd");
  return PETR_OK;
}

extern "C" int petr_reduce_partials(const float* x, int n_partials, long stride, const float* bias,
                                    const float* residual, float* out, long M, int C, void* stream) {
  PETR_CHECK(x && out && M > 0 && C > 0 && n_partials > 0, PETR_ERR_INVALID, "reduce_partials: bad argument");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)cdiv(M * C, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     n_partials, stride, bias, residual, out, M, C);
  PETR_LAUNCH_CHECK("reduce_partials");
  return PETR_OK;
}

extern "C" int petr_reduce_batch(const float* x, int B, long rows, int C, float* out, int accumulate, void* stream) {
  PETR_CHECK(x && out && B > 0 && rows > 0 && C > 0, PETR_ERR_INVALID, "reduce_batch: bad argument");
  hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, x, B,
                     rows * C, out, accumulate);
  PETR_LAUNCH_CHECK("reduce_batch");
  return PETR_OK;
}


// ---- FPN top-down step (reference models/necks/cp_fpn.py:175-186): dst += nearest-upsampled src ----
__global__ __launch_bounds__(256) void fpn_upsample_add_kernel(float* dst, long sv, long sc, long sh, long sw, const float* src,
                                                              int V, int C, int H, int W, int Hs, int Ws, int c_fast) {
  const long n = (long)V * C * H * W;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  int v, c, h, w;
  if (c_fast) {      // channels-last destination: consecutive threads = consecutive channels
    c = (int)(idx % C); long r = idx / C; w = (int)(r % W); r /= W; h = (int)(r % H); v = (int)(r / H);
  } else {
    w = (int)(idx % W); long r = idx / W; h = (int)(r % H); r /= H; c = (int)(r % C); v = (int)(r / C);
  }
  // torch nearest: src = min(int(floorf(dst * scale)), in - 1), scale = float(in) / out
  const float sch = (float)Hs / (float)H, scw = (float)Ws / (float)W;
  const int hs = min((int)floorf((float)h * sch), Hs - 1), ws = min((int)floorf((float)w * scw), Ws - 1);
  dst[(long)v * sv + (long)c * sc + (long)h * sh + (long)w * sw] += src[(((long)v * C + c) * Hs + hs) * Ws + ws];
}

extern "C" int petr_fpn_upsample_add(float* dst, long sv, long sc, long sh, long sw, const float* src, int V, int C, int H,
                                     int W, int Hs, int Ws, void* stream) {
  PETR_CHECK(dst && src && V > 0 && C > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0, PETR_ERR_INVALID, "fpn_upsample_add: bad arguments");
  const long n = (long)V * C * H * W;
  hipLaunchKernelGGL(fpn_upsample_add_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dst, sv, sc, sh,
                     sw, src, V, C, H, W, Hs, Ws, sc == 1 ? 1 : 0);
  PETR_LAUNCH_CHECK("fpn_upsample_add");
  return PETR_OK;
}

// ---- adjoint of the FPN top-down step: dsrc[v,c,hs,ws] (+)= sum of ddst over the destination pixels whose nearest source
//      pixel is (hs, ws) (the gradient autograd sends through F.interpolate(mode='nearest') + add, cp_fpn.py:175-186) ----
__global__ __launch_bounds__(256) void fpn_upsample_add_bwd_kernel(float* dsrc, const float* ddst, long sv, long sc, long sh, long sw,
                                                                  int V, int C, int H, int W, int Hs, int Ws, int accumulate,
                                                                  int c_fast) {
  const long n = (long)V * C * Hs * Ws;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  int v, c, hs, ws;
  if (c_fast) {      // channels-last gradient map: consecutive threads read consecutive channels
    c = (int)(idx % C); long r = idx / C; ws = (int)(r % Ws); r /= Ws; hs = (int)(r % Hs); v = (int)(r / Hs);
  } else {
    ws = (int)(idx % Ws); long r = idx / Ws; hs = (int)(r % Hs); r /= Hs; c = (int)(r % C); v = (int)(r / C);
  }
  // the forward's map, evaluated exactly as fpn_upsample_add_kernel does: src(h) = min(int(floorf(h * scale)), in - 1)
  const float sch = (float)Hs / (float)H, scw = (float)Ws / (float)W;
  auto first_of = [](int s, float sc_, int n_in, int n_out) {      // smallest destination index that maps to a source >= s
    int d = (int)floorf((float)s / sc_) - 1;
    if (d < 0) d = 0;
    while (d < n_out && min((int)floorf((float)d * sc_), n_in - 1) < s) ++d;
    return d;
  };
  const int h0 = first_of(hs, sch, Hs, H), w0 = first_of(ws, scw, Ws, W);
  float acc = 0.f;
  for (int h = h0; h < H && min((int)floorf((float)h * sch), Hs - 1) == hs; ++h)
    for (int w = w0; w < W && min((int)floorf((float)w * scw), Ws - 1) == ws; ++w)
      acc += ddst[(long)v * sv + (long)c * sc + (long)h * sh + (long)w * sw];
  float* o = dsrc + (((long)v * C + c) * Hs + hs) * Ws + ws;
  *o = accumulate ? *o + acc : acc;
}

extern "C" int petr_fpn_upsample_add_bwd(float* dsrc, const float* ddst, long sv, long sc, long sh, long sw, int V, int C, int H,
                                         int W, int Hs, int Ws, int accumulate, void* stream) {
  PETR_CHECK(dsrc && ddst && V > 0 && C > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0, PETR_ERR_INVALID, "fpn_upsample_add_bwd: bad arguments");
  const long n = (long)V * C * Hs * Ws;
  hipLaunchKernelGGL(fpn_upsample_add_bwd_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, dsrc, ddst, sv,
                     sc, sh, sw, V, C, H, W, Hs, Ws, accumulate, sc == 1 ? 1 : 0);
  PETR_LAUNCH_CHECK("fpn_upsample_add_bwd");
  return PETR_OK;
}

// ---- NCHW [V, C, H, W] -> interior of a zero-bordered channels-last map [V, H+2, W+2, C] (the layout the 3x3 contraction of
//      the neck reads: its backward needs the OUTPUT gradient in that layout).  32 x 32 (channel, pixel) tiles through LDS:
//      reads coalesced along w, writes coalesced along c.  The border is the caller's (zero-filled once). ----
__global__ __launch_bounds__(256) void nchw_to_padded_nhwc_kernel(const float* src, float* dst, int C, int H, int W) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
  const int w0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int vh = blockIdx.z, v = vh / H, h = vh - v * H;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, w = w0 + tx;
    tile[ty + 8 * i][tx] = (c < C && w < W) ? src[(((long)v * C + c) * H + h) * W + w] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int w = w0 + ty + 8 * i, c = c0 + tx;
    if (c < C && w < W) dst[(((long)v * (H + 2) + h + 1) * (W + 2) + w + 1) * C + c] = tile[tx][ty + 8 * i];
  }
}

extern "C" int petr_nchw_to_padded_nhwc(const float* src, float* dst, int V, int C, int H, int W, void* stream) {
  PETR_CHECK(src && dst && V > 0 && C > 0 && H > 0 && W > 0, PETR_ERR_INVALID, "nchw_to_padded_nhwc: bad arguments");
  PETR_CHECK((long)V * H <= 65535 && cdiv(C, 32) <= 65535, PETR_ERR_UNSUPPORTED, "nchw_to_padded_nhwc: V * H and C / 32 must be <= 65535");
  hipLaunchKernelGGL(nchw_to_padded_nhwc_kernel, dim3((unsigned)cdiv(W, 32), (unsigned)cdiv(C, 32), (unsigned)(V * H)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, C, H, W);
  PETR_LAUNCH_CHECK("nchw_to_padded_nhwc");
  return PETR_OK;
}

// Position-embedding producers of the PETRHead path (HBM-bound, integer index generation -> fp32):
//   K1  camera-frustum -> LiDAR coordinate volume      (reference petr_head.py:290-331)
//   K3a SinePositionalEncoding3D features              (reference positional_encoding.py:58-100)
//   K3b pos2posemb3d for the object queries            (reference petr_head.py:31-43)
// Output layouts are the reference's ([B*N, C, H, W]); the GEMM consumes them directly as an
// "M-contiguous" A operand, so no permute copy exists anywhere on the path.
#include "common.h"

// ------------------------------------------------------------------------------------------
// K1.  One thread = one depth bin d of 4 consecutive pixels of one view; writes 3 float4
// (x,y,z channels 3d..3d+2).  blockIdx.y = view so the 4x4 inverse projection is wave-uniform
// and is held in SGPRs (scalar loads through the constant cache) instead of LDS.
// Arithmetic follows the reference op for op in fp32.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float logit_clamped(float x, float eps) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float x1 = fmaxf(x, eps);
  const float x2 = fmaxf(1.f - x, eps);
  return logf(x1 / x2);
}

struct Coords3dParams {
  const float* img2lidar;
  const float* depth;
  float* out;
  uint8_t* cmask;
  int N, H, W, D;
  float pad_h, pad_w;
  float lo[3], span[3];
  float eps;
};

__device__ __forceinline__ void frustum_point(const float* __restrict__ m, float u, float v, float dv, float eps,
                                              const float lo[3], const float span[3], float nrm[3]) {
  const float s = fmaxf(dv, eps);
  const float p0 = __fmul_rn(u, s), p1 = __fmul_rn(v, s), p2 = dv;
  // The reference multiplies [4x4] by [4x1] with torch.matmul (petr_head.py:317-319): a k-ordered sum of individually
  // rounded fp32 products, no fused multiply-add (bmm's small-matrix path).  The same sequence here - every product and
  // every sum rounded on its own - makes the normalised coordinates, and with them the > 1 / < 0 tests behind
  // coords_mask (:327-328), come out bit for bit like the reference's; a contracted fma chain flipped 1-2 mask pixels per
  // sample whose coordinate sits within an ulp of the range border.
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float x = __fmul_rn(m[4 * a + 0], p0);
    x = __fadd_rn(x, __fmul_rn(m[4 * a + 1], p1));
    x = __fadd_rn(x, __fmul_rn(m[4 * a + 2], p2));
    x = __fadd_rn(x, m[4 * a + 3]);             // p3 = 1: the product is exact
    nrm[a] = __fdiv_rn(__fsub_rn(x, lo[a]), span[a]);
  }
}

__global__ __launch_bounds__(256) void coords3d_kernel(Coords3dParams p) {
  const int HW = p.H * p.W;
  const int quads = HW >> 2;
  const int bn = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= quads * p.D) return;
  const int d = idx / quads;
  const int quad = idx - d * quads;
  const float* __restrict__ m = p.img2lidar + (size_t)bn * 16;  // uniform -> s_load
  const float dv = p.depth[d];
  float o[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int hw = quad * 4 + j;
    const int h = hw / p.W, w = hw - h * p.W;
    const float u = __fdiv_rn(__fmul_rn((float)w, p.pad_w), (float)p.W);
    const float v = __fdiv_rn(__fmul_rn((float)h, p.pad_h), (float)p.H);
    float nrm[3];
    frustum_point(m, u, v, dv, p.eps, p.lo, p.span, nrm);
#pragma unroll
    for (int a = 0; a < 3; ++a) o[a][j] = logit_clamped(nrm[a], p.eps);
  }
  float* base = p.out + ((size_t)bn * 3 * p.D + 3 * d) * HW + quad * 4;
#pragma unroll
  for (int a = 0; a < 3; ++a)
    *reinterpret_cast<float4*>(base + (size_t)a * HW) = make_float4(o[a][0], o[a][1], o[a][2], o[a][3]);
}

// scalar-tail variant for H*W not a multiple of 4 (never the case in the BASELINE shapes)
__global__ __launch_bounds__(256) void coords3d_scalar_kernel(Coords3dParams p) {
  const int HW = p.H * p.W;
  const int bn = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= HW * p.D) return;
  const int d = idx / HW;
  const int hw = idx - d * HW;
  const int h = hw / p.W, w = hw - h * p.W;
  const float* __restrict__ m = p.img2lidar + (size_t)bn * 16;
  float nrm[3];
  frustum_point(m, __fdiv_rn(__fmul_rn((float)w, p.pad_w), (float)p.W), __fdiv_rn(__fmul_rn((float)h, p.pad_h), (float)p.H),
                p.depth[d], p.eps, p.lo, p.span, nrm);
  float* base = p.out + ((size_t)bn * 3 * p.D + 3 * d) * HW + hw;
#pragma unroll
  for (int a = 0; a < 3; ++a) base[(size_t)a * HW] = logit_clamped(nrm[a], p.eps);
}

// coords_mask (petr_head.py:327-328): returned by position_embeding() but discarded by
// PETRHead.forward (:397), so it is a separate, optional kernel off the hot path.
__global__ __launch_bounds__(256) void coords3d_mask_kernel(Coords3dParams p) {
  const int HW = p.H * p.W;
  const int bn = blockIdx.y;
  const int hw = blockIdx.x * blockDim.x + threadIdx.x;
  if (hw >= HW) return;
  const int h = hw / p.W, w = hw - h * p.W;
  const float* __restrict__ m = p.img2lidar + (size_t)bn * 16;
  const float u = __fdiv_rn(__fmul_rn((float)w, p.pad_w), (float)p.W);
  const float v = __fdiv_rn(__fmul_rn((float)h, p.pad_h), (float)p.H);
  int cnt = 0;
  for (int d = 0; d < p.D; ++d) {
    float nrm[3];
    frustum_point(m, u, v, p.depth[d], p.eps, p.lo, p.span, nrm);
#pragma unroll
    for (int a = 0; a < 3; ++a) cnt += (nrm[a] > 1.0f) | (nrm[a] < 0.0f);
  }
  p.cmask[(size_t)bn * HW + hw] = ((float)cnt > (float)p.D * 0.5f) ? 1 : 0;
}

extern "C" int petr_coords3d_fwd(const petr_coords3d_args* a, void* stream) {
  PETR_CHECK(a && a->img2lidar && a->depth && (a->out || a->cmask), PETR_ERR_INVALID, "coords3d: null pointer");
  PETR_CHECK(a->B > 0 && a->N > 0 && a->H > 0 && a->W > 0 && a->D > 0, PETR_ERR_INVALID, "coords3d: bad shape");
  Coords3dParams p;
  p.img2lidar = a->img2lidar;
  p.depth = a->depth;
  p.out = a->out;
  p.cmask = a->cmask;
  p.N = a->N; p.H = a->H; p.W = a->W; p.D = a->D;
  p.pad_h = a->pad_h; p.pad_w = a->pad_w;
  for (int i = 0; i < 3; ++i) {
    p.lo[i] = a->range[i];
    p.span[i] = (float)((double)a->range[i + 3] - (double)a->range[i]);
  }
  p.eps = a->eps;
  const int HW = a->H * a->W, BN = a->B * a->N;
  hipStream_t s = (hipStream_t)stream;
  if (a->out) {
    if ((HW & 3) == 0 && aligned16(a->out)) {
      dim3 grid((unsigned)cdiv((long)(HW / 4) * a->D, 256), BN);
      hipEvent_t ev0, ev1;
      petr_prof_claim(PETR_PROF_COORDS3D, &ev0, &ev1);
      hipExtLaunchKernelGGL(coords3d_kernel, grid, dim3(256), 0, s, ev0, ev1, 0, p);
    } else {
      dim3 grid((unsigned)cdiv((long)HW * a->D, 256), BN);
      hipLaunchKernelGGL(coords3d_scalar_kernel, grid, dim3(256), 0, s, p);
    }
    PETR_LAUNCH_CHECK("coords3d");
  }
  if (a->cmask) {
    dim3 grid((unsigned)cdiv(HW, 256), BN);
    hipLaunchKernelGGL(coords3d_mask_kernel, grid, dim3(256), 0, s, p);
    PETR_LAUNCH_CHECK("coords3d_mask");
  }
  return PETR_OK;
}

// ------------------------------------------------------------------------------------------
// K3a.  Block = 64 consecutive pixels of one (b, n) view.  Phase 1: 64x3 threads compute the
// normalised cumulative-count embeddings (integer cumsum of the non-masked positions along
// N / H / W, exactly as the reference) into LDS.  Phase 2: every thread emits (sin, cos) pairs
// for its pixel over a strided set of channel pairs; stores are coalesced along the pixel axis.
// ------------------------------------------------------------------------------------------
struct Sine3dParams {
  const uint8_t* mask;
  const float* dim_t;
  float* out;
  int B, N, H, W, F;
  int normalize;
  float scale, eps, offset;
};

__global__ __launch_bounds__(256) void sine3d_kernel(Sine3dParams p) {
  __shared__ float emb[3][64];
  const int HW = p.H * p.W;
  const int bn = blockIdx.y;
  const int b = bn / p.N, n = bn - b * p.N;
  const int pix0 = blockIdx.x * 64;
  const int t = threadIdx.x;
  if (t < 192) {
    const int axis = t >> 6;  // 0: n, 1: y, 2: x  (concat order positional_encoding.py:99)
    const int hw = pix0 + (t & 63);
    if (hw < HW) {
      const int h = hw / p.W, w = hw - h * p.W;
      int cum = 0, tot = 0;
      if (p.mask == nullptr) {
        if (axis == 0) { cum = n + 1; tot = p.N; }
        else if (axis == 1) { cum = h + 1; tot = p.H; }
        else { cum = w + 1; tot = p.W; }
      } else {
        const uint8_t* mb = p.mask + (size_t)b * p.N * HW;
        if (axis == 0) {
          for (int i = 0; i < p.N; ++i) { const int v = mb[(size_t)i * HW + hw] ? 0 : 1; tot += v; if (i <= n) cum += v; }
        } else if (axis == 1) {
          for (int i = 0; i < p.H; ++i) { const int v = mb[(size_t)n * HW + i * p.W + w] ? 0 : 1; tot += v; if (i <= h) cum += v; }
        } else {
          for (int i = 0; i < p.W; ++i) { const int v = mb[(size_t)n * HW + h * p.W + i] ? 0 : 1; tot += v; if (i <= w) cum += v; }
        }
      }
      float e = (float)cum;
      if (p.normalize) e = (e + p.offset) / ((float)tot + p.eps) * p.scale;
      emb[axis][t & 63] = e;
    }
  }
  __syncthreads();
  const int pix = t & 63;
  const int hw = pix0 + pix;
  if (hw >= HW) return;
  const int half = p.F >> 1;           // channel pairs per axis
  const int pairs = 3 * half;
  float* ob = p.out + (size_t)bn * 3 * p.F * HW + hw;
  for (int pr = t >> 6; pr < pairs; pr += 4) {
    const int axis = pr / half;
    const int k = pr - axis * half;
    const float e = emb[axis][pix];
    // positional_encoding.py:90-98 stacks (sin(even), cos(odd)) on a NEW dim 4 of a 5-D tensor and
    // flattens [2][F/2]: the F/2 sines come first, then the F/2 cosines (NOT interleaved, unlike
    // pos2posemb3d which stacks on the last dim).
    const float s = sinf(e / p.dim_t[2 * k]);
    const float c = cosf(e / p.dim_t[2 * k + 1]);
    ob[(size_t)(axis * p.F + k) * HW] = s;
    ob[(size_t)(axis * p.F + half + k) * HW] = c;
  }
}

// Without a padding mask the three embeddings are functions of ONE coordinate each (e_n of the view, e_y of the row, e_x
// of the column), so a row of W pixels needs 64 + 64 + 64 W sine / cosine pairs instead of 192 W: one workgroup per
// (view, row) evaluates them once into LDS ([pair][w], conflict-free both ways) and then only streams the 3 F channel rows
// out.  Same expressions on the same arguments as sine3d_kernel, hence the same bits; 12x fewer sinf / cosf calls - the
// general kernel spent 40 - 55 us on them, on the critical path of the position-embedding phase.
__global__ __launch_bounds__(256) void sine3d_rows_kernel(Sine3dParams p) {
  extern __shared__ float tab[];                 // [2 (sin, cos)][half pairs][W + 1] for the x axis, then 2 x 2 x half for n / y
  const int half = p.F >> 1;
  const int WP = p.W + 1;
  float* xs = tab;
  float* xc = tab + half * WP;
  float* ny = tab + 2 * half * WP;               // [axis 0 / 1][sin / cos][half]
  const int bn = blockIdx.y, h = blockIdx.x;
  const int b = bn / p.N, n = bn - b * p.N;
  (void)b;
  const int t = threadIdx.x;
  auto emb = [&](int cum, int tot) {
    float e = (float)cum;
    if (p.normalize) e = (e + p.offset) / ((float)tot + p.eps) * p.scale;
    return e;
  };
  // sin and cos of one pair share their argument whenever dim_t[2k] == dim_t[2k+1] (always, for the reference's dim_t):
  // sincosf pays ONE range reduction; its two results are those of sinf / cosf (same reduction, same polynomials)
  auto pair = [&](float e, int k, float* sn, float* cs) {
    const float d0 = p.dim_t[2 * k], d1 = p.dim_t[2 * k + 1];
    if (d0 == d1) {
      sincosf(e / d0, sn, cs);
    } else {
      *sn = sinf(e / d0);
      *cs = cosf(e / d1);
    }
  };
  // blockIdx.z cuts the pairs into slices (the evaluations are a dependent instruction chain of ~0.5 us each at one wave per
  // SIMD: 4 per thread instead of 16)
  const int kper = (half + gridDim.z - 1) / gridDim.z;
  const int k_lo = blockIdx.z * kper, k_hi = min(half, k_lo + kper);
  for (int k = k_lo + (t >> 6); k < k_hi; k += 4)        // wave-uniform pair index, lanes over the columns
    for (int w = t & 63; w < p.W; w += 64) pair(emb(w + 1, p.W), k, xs + k * WP + w, xc + k * WP + w);
  for (int i = t; i < 2 * (k_hi - k_lo); i += 256) {
    const int axis = i >= (k_hi - k_lo) ? 1 : 0, k = k_lo + i - axis * (k_hi - k_lo);
    pair(axis == 0 ? emb(n + 1, p.N) : emb(h + 1, p.H), k, ny + (2 * axis) * half + k, ny + (2 * axis + 1) * half + k);
  }
  __syncthreads();
  const int HW = p.H * p.W;
  float* ob = p.out + (size_t)bn * 3 * p.F * HW + (size_t)h * p.W;
  // channel (axis, sc, k) = axis * F + sc * half + k: the F/2 sines first, then the F/2 cosines (see sine3d_kernel);
  // one wave per channel row, lanes over the columns: no integer division in the streaming loop
  const int kn = k_hi - k_lo;
  for (int j = t >> 6; j < 6 * kn; j += 4) {             // j -> (axis, sc, k) of this slice
    const int as = j / kn, k = k_lo + j - as * kn;
    const int axis = as >> 1, sc = as & 1;
    const int ch = axis * p.F + sc * half + k;
    const float* src = (sc ? xc : xs) + k * WP;
    const float nyv = axis < 2 ? ny[(2 * axis + sc) * half + k] : 0.f;
    for (int w = t & 63; w < p.W; w += 64) ob[(size_t)ch * HW + w] = axis == 2 ? src[w] : nyv;
  }
}

// The same tables written PLANE-wise: one workgroup per (pair k, view) evaluates sin / cos of the W column, the H row and the
// one view argument of its pair - one sincosf per thread - and streams its six channel planes [H][W] out as 16-byte stores
// (a plane of the NCHW output is contiguous; sine3d_rows_kernel writes a W-float segment per channel and row: 48 us for the
// 37 MB of the 6 x 40 x 100 shape).  Needs W % 4 == 0 and W + H + 1 <= 256; same expressions on the same arguments, same bits.
__global__ __launch_bounds__(256) void sine3d_planes_kernel(Sine3dParams p) {
  __shared__ __attribute__((aligned(16))) float xs[256], xc[256];
  __shared__ float ys[256], yc[256], nsc[2];
  const int half = p.F >> 1;
  const int k = blockIdx.x, bn = blockIdx.y;
  const int n = bn % p.N;
  const int t = threadIdx.x;
  auto emb = [&](int cum, int tot) {
    float e = (float)cum;
    if (p.normalize) e = (e + p.offset) / ((float)tot + p.eps) * p.scale;
    return e;
  };
  {
    float e = 0.f;
    float *sn = nullptr, *cs = nullptr;
    if (t < p.W) { e = emb(t + 1, p.W); sn = xs + t; cs = xc + t; }
    else if (t < p.W + p.H) { e = emb(t - p.W + 1, p.H); sn = ys + (t - p.W); cs = yc + (t - p.W); }
    else if (t == p.W + p.H) { e = emb(n + 1, p.N); sn = nsc; cs = nsc + 1; }
    if (sn) {
      const float d0 = p.dim_t[2 * k], d1 = p.dim_t[2 * k + 1];
      if (d0 == d1) {
        sincosf(e / d0, sn, cs);
      } else {
        *sn = sinf(e / d0);
        *cs = cosf(e / d1);
      }
    }
  }
  __syncthreads();
  const int HW = p.H * p.W, n4 = HW >> 2;
  float* ob = p.out + (size_t)bn * 3 * p.F * HW;
  // channel (axis, sc, k) = axis * F + sc * half + k (see sine3d_kernel)
  float4* pl[6];
#pragma unroll
  for (int as = 0; as < 6; ++as) pl[as] = reinterpret_cast<float4*>(ob + (size_t)((as >> 1) * p.F + (as & 1) * half + k) * HW);
  const float4 cn0 = make_float4(nsc[0], nsc[0], nsc[0], nsc[0]), cn1 = make_float4(nsc[1], nsc[1], nsc[1], nsc[1]);
  const int dh = 1024 / p.W, dw = 1024 - dh * p.W;         // one step of the loop advances 1024 pixels
  int h = (4 * t) / p.W, w = 4 * t - h * p.W;
  for (int i4 = t; i4 < n4; i4 += 256) {
    const float y0 = ys[h], y1 = yc[h];
    pl[0][i4] = cn0;
    pl[1][i4] = cn1;
    pl[2][i4] = make_float4(y0, y0, y0, y0);
    pl[3][i4] = make_float4(y1, y1, y1, y1);
    pl[4][i4] = *reinterpret_cast<const float4*>(xs + w);
    pl[5][i4] = *reinterpret_cast<const float4*>(xc + w);
    h += dh; w += dw;
    if (w >= p.W) { w -= p.W; ++h; }
  }
}

extern "C" int petr_sine3d_fwd(const petr_sine3d_args* a, void* stream) {
  PETR_CHECK(a && a->dim_t && a->out, PETR_ERR_INVALID, "sine3d: null pointer");
  PETR_CHECK(a->F > 0 && (a->F & 1) == 0, PETR_ERR_INVALID, "sine3d: num_feats must be even");
  Sine3dParams p{a->mask, a->dim_t, a->out, a->B, a->N, a->H, a->W, a->F, a->normalize, a->scale, a->eps, a->offset};
  const size_t rows_lds = ((size_t)a->F * (a->W + 1) + 2 * (size_t)a->F) * sizeof(float);
  static const bool planes_on = petr_tune("PETR_SINE3D_PLANES", 1) != 0;
  if (!a->mask && planes_on && a->W % 4 == 0 && a->W + a->H + 1 <= 256 && aligned16(a->out)) {
    hipLaunchKernelGGL(sine3d_planes_kernel, dim3(a->F / 2, a->B * a->N), dim3(256), 0, (hipStream_t)stream, p);
    PETR_LAUNCH_CHECK("sine3d_planes");
    return PETR_OK;
  }
  if (!a->mask && rows_lds <= 64 * 1024) {        // no padding anywhere: the per-row kernel
    hipLaunchKernelGGL(sine3d_rows_kernel, dim3(a->H, a->B * a->N, 4), dim3(256), rows_lds, (hipStream_t)stream, p);
    PETR_LAUNCH_CHECK("sine3d_rows");
    return PETR_OK;
  }
  dim3 grid((unsigned)cdiv((long)a->H * a->W, 64), a->B * a->N);
  hipLaunchKernelGGL(sine3d_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  PETR_LAUNCH_CHECK("sine3d");
  return PETR_OK;
}

// ------------------------------------------------------------------------------------------
// K3b.  pos2posemb3d: out[q, blk*F + 2k] = sin(pos[q,ax]*2pi / dim_t[2k]), +1 -> cos; block order
// (y, x, z) = pos axis (1, 0, 2)  (petr_head.py:42).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void posemb3d_kernel(const float* __restrict__ pos, const float* __restrict__ dim_t,
                                                        float* __restrict__ out, int n, int F) {
  const int half = F >> 1;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * 3 * half) return;
  const int q = idx / (3 * half);
  const int rem = idx - q * 3 * half;
  const int blk = rem / half, k = rem - blk * half;
  const int ax = blk == 0 ? 1 : (blk == 1 ? 0 : 2);
  const float scale = 2.f * 3.14159265358979323846f;   // fp32(2*pi), as torch does for a python-float multiplier
  const float v = pos[q * 3 + ax] * scale;
  out[(size_t)q * 3 * F + blk * F + 2 * k] = sinf(v / dim_t[2 * k]);
  out[(size_t)q * 3 * F + blk * F + 2 * k + 1] = cosf(v / dim_t[2 * k + 1]);
}

// dpos[q,ax] = sum_k 2pi/dim_t[2k] * ( cos(a)*dout[sin ch] - sin(a)*dout[cos ch] ); one wave per (q, blk)
__global__ __launch_bounds__(256) void posemb3d_bwd_kernel(const float* __restrict__ pos, const float* __restrict__ dim_t,
                                                            const float* __restrict__ dout, float* __restrict__ dpos,
                                                            int n, int F) {
  const int half = F >> 1;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n * 3) return;
  const int q = wave / 3, blk = wave - q * 3;
  const int ax = blk == 0 ? 1 : (blk == 1 ? 0 : 2);
  const float scale = 2.f * 3.14159265358979323846f;
  const float v = pos[q * 3 + ax] * scale;
  float acc = 0.f;
  for (int k = lane; k < half; k += 64) {
    const float t0 = dim_t[2 * k], t1 = dim_t[2 * k + 1];
    const float g0 = dout[(size_t)q * 3 * F + blk * F + 2 * k];
    const float g1 = dout[(size_t)q * 3 * F + blk * F + 2 * k + 1];
    acc += g0 * cosf(v / t0) * (scale / t0) - g1 * sinf(v / t1) * (scale / t1);
  }
  acc = wave_sum(acc);
  if (lane == 0) dpos[q * 3 + ax] += acc;
}

extern "C" int petr_posemb3d_fwd(const float* pos, const float* dim_t, float* out, int n, int F, void* stream) {
  PETR_CHECK(pos && dim_t && out && n > 0 && F > 0 && (F & 1) == 0, PETR_ERR_INVALID, "posemb3d: bad argument");
  hipLaunchKernelGGL(posemb3d_kernel, dim3((unsigned)cdiv((long)n * 3 * (F / 2), 256)), dim3(256), 0,
                     (hipStream_t)stream, pos, dim_t, out, n, F);
  PETR_LAUNCH_CHECK("posemb3d");
  return PETR_OK;
}

extern "C" int petr_posemb3d_bwd(const float* pos, const float* dim_t, const float* dout, float* dpos, int n, int F,
                                 void* stream) {
  PETR_CHECK(pos && dim_t && dout && dpos && n > 0 && F > 0 && (F & 1) == 0, PETR_ERR_INVALID, "posemb3d_bwd: bad argument");
  hipLaunchKernelGGL(posemb3d_bwd_kernel, dim3((unsigned)cdiv((long)n * 3 * 64, 256)), dim3(256), 0,
                     (hipStream_t)stream, pos, dim_t, dout, dpos, n, F);
  PETR_LAUNCH_CHECK("posemb3d_bwd");
  return PETR_OK;
}

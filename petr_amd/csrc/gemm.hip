// fp32 contraction on the CDNA4 matrix cores:  C[m,n] (+)= act(alpha * sum_k A(m,k) B(n,k) + bias + R)
//
// v_mfma_f32_32x32x2_f32: fp32 operands, fp32 accumulate, exact fp32 products (no xf32/TF32 on
// gfx950), 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak.  At that rate LDS and HBM are never the
// bound for these shapes (a 64x64 wave tile needs 4 ds_read_b32 per 4 MFMAs = 8 LDS cycles per
// 256 MFMA cycles), so the kernel is built for generality and exactness:
//   * both operands may be K-contiguous ([rows][K], nn.Linear weights / token-major activations)
//     or row-contiguous ([K][rows], NCHW feature maps and every transposed use in backward);
//     either way the LDS image is [k][row] so a fragment read is 32 consecutive floats per
//     lane-half (conflict-free ds_read_b32) and the accumulator columns land on lanes, giving
//     128-byte coalesced row-major C stores;
//   * K-contiguous tiles are transposed on the LDS write with a row pitch == 1 (mod 8) so the four
//     ds_write_b32 of a float4 hit 32 distinct banks;
//   * register-staged prefetch of tile t+1 during the MFMAs of tile t (cdna guide T14);
//   * fused prologue (A + A2: query+query_pos / key+key_pos) and epilogue (bias, residual, ReLU,
//     ReLU-mask for backward, sigmoid-gate, accumulate, head-split K/V store, split-K slabs).
#include "common.h"

namespace {

constexpr int BK = 32;

struct OperandView {
  const float* p;
  long ld;
  int rows;    // valid rows (M or N)
  int vec_ok;  // 16-byte vector loads legal (alignment + extents)
};

// ---- global -> register stage -------------------------------------------------------------
template <int ROWS, bool KC>
struct Stage {
  static constexpr int NV = ROWS * BK / 4 / 256;
  static constexpr int LD = KC ? ROWS + 1 : ROWS + 4;
  float4 v[NV];

  __device__ __forceinline__ void load(const OperandView& o, long seg_off, int row0, int k0, int kend, const float* a2,
                                       int a2_rows) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = t + 256 * i;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KC) {
        const int row = row0 + (idx >> 3);
        const int k = k0 + 4 * (idx & 7);
        if (row < o.rows && k < kend) {
          const float* src = o.p + seg_off + (long)row * o.ld + k;
          if (o.vec_ok && k + 3 < kend) {
            r = *reinterpret_cast<const float4*>(src);
          } else {
            r.x = src[0];
            if (k + 1 < kend) r.y = src[1];
            if (k + 2 < kend) r.z = src[2];
            if (k + 3 < kend) r.w = src[3];
          }
          if (a2) {
            const int r2 = a2_rows > 0 ? row % a2_rows : row;
            const float* s2 = a2 + seg_off + (long)r2 * o.ld + k;
            if (o.vec_ok && k + 3 < kend) {
              const float4 q = *reinterpret_cast<const float4*>(s2);
              r.x += q.x; r.y += q.y; r.z += q.z; r.w += q.w;
            } else {
              r.x += s2[0];
              if (k + 1 < kend) r.y += s2[1];
              if (k + 2 < kend) r.z += s2[2];
              if (k + 3 < kend) r.w += s2[3];
            }
          }
        }
      } else {
        constexpr int RPK = ROWS / 4;
        const int k = k0 + idx / RPK;
        const int row = row0 + 4 * (idx % RPK);
        if (k < kend && row < o.rows) {
          const float* src = o.p + seg_off + (long)k * o.ld + row;
          if (o.vec_ok && row + 3 < o.rows) {
            r = *reinterpret_cast<const float4*>(src);
          } else {
            r.x = src[0];
            if (row + 1 < o.rows) r.y = src[1];
            if (row + 2 < o.rows) r.z = src[2];
            if (row + 3 < o.rows) r.w = src[3];
          }
        }
      }
      v[i] = r;
    }
  }

  __device__ __forceinline__ void store(float* lds) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = t + 256 * i;
      if (KC) {
        const int row = idx >> 3, kq = idx & 7;
        float* d = lds + (4 * kq) * LD + row;
        d[0] = v[i].x;
        d[LD] = v[i].y;
        d[2 * LD] = v[i].z;
        d[3 * LD] = v[i].w;
      } else {
        constexpr int RPK = ROWS / 4;
        const int k = idx / RPK, r4 = idx % RPK;
        *reinterpret_cast<float4*>(lds + k * LD + 4 * r4) = v[i];
      }
    }
  }
};

template <int BM, int BN, int WM, int WN, bool AKC, bool BKC>
__global__ __launch_bounds__(256) void gemm_kernel(const petr_gemm_args g, const int a_vec, const int b_vec,
                                                    const int tiles_m, const int tiles_n) {
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
  constexpr int TM = WM / 32, TN = WN / 32;
  using SA = Stage<BM, AKC>;
  using SB = Stage<BN, BKC>;
  __shared__ __attribute__((aligned(16))) float lds[BK * SA::LD + BK * SB::LD];
  float* As = lds;
  float* Bs = lds + BK * SA::LD;

  // ---- which tile / batch / K-slice ----
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm_i = tile / tiles_n, tn_i = tile - tm_i * tiles_n;
  const int m0 = tm_i * BM, n0 = tn_i * BN;
  int z = blockIdx.z;
  const int ks = z % g.split_k;
  z /= g.split_k;
  const int z1 = z % g.nb1, z0 = z / g.nb1;

  OperandView A{g.a + z0 * g.a_bs0 + z1 * g.a_bs1, g.lda, g.M, a_vec};
  OperandView B{g.b + z0 * g.b_bs0 + z1 * g.b_bs1, g.ldb, g.N, b_vec};
  const float* a2 = (g.a2 && (g.a2_ncols <= 0 || n0 < g.a2_ncols)) ? g.a2 : nullptr;

  // K may be cut into segments that live at different base addresses (k = seg*k_seg + kk):
  // sums over (layer, channel), (view, pixel) or (batch, token) in the backward contractions.
  const int kseg = g.k_seg > 0 ? g.k_seg : g.K;
  const int nseg = g.k_seg > 0 ? g.K / g.k_seg : 1;
  const int tps = (kseg + BK - 1) / BK;
  const int ktiles = nseg * tps;
  const int kt_per = (ktiles + g.split_k - 1) / g.split_k;
  const int kt_begin = ks * kt_per;
  const int kt_end = min(ktiles, kt_begin + kt_per);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, c = lane & 31;
  constexpr int WAVES_N = BN / WN;
  const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // bias gradient for free: column sums of the A operand (= dC^T in a weight-gradient
  // contraction) taken from the LDS image by the workgroups of the first N-tile
  const bool do_colsum = g.a_colsum != nullptr && tn_i == 0;
  float colacc = 0.f;

  SA sa;
  SB sb;
  auto stage_load = [&](int kt) {
    const int seg = kt / tps, k0 = (kt - seg * tps) * BK;
    sa.load(A, (long)seg * g.a_seg_stride, m0, k0, kseg, a2, g.a2_rows);
    sb.load(B, (long)seg * g.b_seg_stride, n0, k0, kseg, nullptr, 0);
  };
  if (kt_begin < kt_end) stage_load(kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    __syncthreads();
    sa.store(As);
    sb.store(Bs);
    __syncthreads();
    if (kt + 1 < kt_end) stage_load(kt + 1);
    if (do_colsum && threadIdx.x < BM) {
      float cs = 0.f;
#pragma unroll
      for (int k = 0; k < BK; ++k) cs += As[k * SA::LD + threadIdx.x];
      colacc += cs;
    }
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[(2 * s + h) * SA::LD + wm0 + i * 32 + c];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[(2 * s + h) * SB::LD + wn0 + j * 32 + c];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

  if (do_colsum && threadIdx.x < BM && m0 + (int)threadIdx.x < g.M)
    atomicAdd(g.a_colsum + z0 * g.cs_bs0 + z1 * g.cs_bs1 + m0 + threadIdx.x, colacc);

  // ---- epilogue ----
  const bool atomic = (g.flags & PETR_GEMM_ATOMIC) != 0;
  float* C = g.c + z0 * g.c_bs0 + z1 * g.c_bs1 + (atomic ? 0 : (long)ks * g.c_split_stride);
  const bool plain = g.split_k > 1 && !atomic;   // K-slices store raw partial sums
  const float* bias = (!plain && g.bias) ? g.bias + z0 * g.bias_bs0 + z1 * g.bias_bs1 : nullptr;
  const float* R = (!plain && g.r) ? g.r + z0 * g.r_bs0 + z1 * g.r_bs1 : nullptr;
  const int flags = plain ? 0 : g.flags;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn0 + j * 32 + c;
    if (n >= g.N) continue;
    const float bv = bias ? bias[n] : 0.f;
    const long ccol = g.c_nblk > 0 ? (long)(n / g.c_nblk) * g.c_nblk_stride + (n % g.c_nblk) : (long)n;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + mfma32_row(r, h);
        if (m >= g.M) continue;
        float v = acc[i][j][r] * g.alpha + bv;
        if (flags & PETR_GEMM_SIGMOID_MUL) {
          v = R[(long)m * g.ldr + n] * (1.f / (1.f + expf(-v)));
        } else if (flags & PETR_GEMM_RELU_MASK) {
          v = R[(long)m * g.ldr + n] > 0.f ? v : 0.f;
        } else if (R) {
          v += R[(long)m * g.ldr + n];
        }
        if (flags & PETR_GEMM_RELU) v = fmaxf(v, 0.f);
        float* dst = C + (long)m * g.ldc + ccol;
        if (atomic) {
          atomicAdd(dst, v);
        } else {
          if (flags & PETR_GEMM_ACCUMULATE) v += *dst;
          *dst = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const petr_gemm_args& g, int a_vec, int b_vec, hipStream_t s) {
  const int tiles_m = (int)cdiv(g.M, BM), tiles_n = (int)cdiv(g.N, BN);
  dim3 grid(tiles_m * tiles_n, 1, g.nb0 * g.nb1 * g.split_k);
  dim3 block(256);
#define PETR_GEMM_LAUNCH(AKC, BKC) \
  hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, AKC, BKC>), grid, block, 0, s, g, a_vec, b_vec, tiles_m, tiles_n)
  if (g.a_kcontig && g.b_kcontig) PETR_GEMM_LAUNCH(true, true);
  else if (g.a_kcontig) PETR_GEMM_LAUNCH(true, false);
  else if (g.b_kcontig) PETR_GEMM_LAUNCH(false, true);
  else PETR_GEMM_LAUNCH(false, false);
#undef PETR_GEMM_LAUNCH
  PETR_LAUNCH_CHECK("gemm");
  return PETR_OK;
}

bool operand_vec_ok(const float* p, long ld, long bs0, long bs1, long seg, int kcontig, int rows, int K) {
  if (!aligned16(p) || (ld & 3) || (bs0 & 3) || (bs1 & 3) || (seg & 3)) return false;
  // contiguous extent must be a multiple of 4 so that a float4 never straddles the valid edge
  return kcontig ? (K % 4 == 0) : (rows % 4 == 0);
}

}  // namespace

extern "C" int petr_gemm(const petr_gemm_args* gp, void* stream) {
  PETR_CHECK(gp && gp->a && gp->b && gp->c, PETR_ERR_INVALID, "gemm: null pointer");
  petr_gemm_args g = *gp;
  PETR_CHECK(g.M > 0 && g.N > 0 && g.K > 0, PETR_ERR_INVALID, "gemm: bad shape M=%d N=%d K=%d", g.M, g.N, g.K);
  if (g.nb0 <= 0) g.nb0 = 1;
  if (g.nb1 <= 0) g.nb1 = 1;
  if (g.split_k <= 0) g.split_k = 1;
  if (g.alpha == 0.f) g.alpha = 1.f;
  PETR_CHECK(!(g.a2 && !g.a_kcontig), PETR_ERR_UNSUPPORTED, "gemm: a2 addend needs a K-contiguous A");
  PETR_CHECK(!((g.flags & (PETR_GEMM_RELU_MASK | PETR_GEMM_SIGMOID_MUL)) && !g.r), PETR_ERR_INVALID,
             "gemm: mask/mul flag without r operand");
  PETR_CHECK(!(g.split_k > 1 && g.c_split_stride <= 0 && !(g.flags & PETR_GEMM_ATOMIC)), PETR_ERR_INVALID,
             "gemm: split_k needs c_split_stride (or PETR_GEMM_ATOMIC)");
  PETR_CHECK(!((g.flags & PETR_GEMM_ATOMIC) && (g.bias || g.r || (g.flags & ~PETR_GEMM_ATOMIC))), PETR_ERR_UNSUPPORTED,
             "gemm: PETR_GEMM_ATOMIC excludes bias/residual/other flags");
  PETR_CHECK((long)g.nb0 * g.nb1 * g.split_k <= 65535, PETR_ERR_UNSUPPORTED, "gemm: too many batches");
  PETR_CHECK(g.k_seg <= 0 || g.K % g.k_seg == 0, PETR_ERR_INVALID, "gemm: K=%d is not a multiple of k_seg=%d", g.K, g.k_seg);
  const int kseg = g.k_seg > 0 ? g.k_seg : g.K;
  const int a_vec = operand_vec_ok(g.a, g.lda, g.a_bs0, g.a_bs1, g.a_seg_stride, g.a_kcontig, g.M, kseg) &&
                    (!g.a2 || aligned16(g.a2));
  const int b_vec = operand_vec_ok(g.b, g.ldb, g.b_bs0, g.b_bs1, g.b_seg_stride, g.b_kcontig, g.N, kseg);
  hipStream_t s = (hipStream_t)stream;
  const long nz = (long)g.nb0 * g.nb1 * g.split_k;
  const long b128 = cdiv(g.M, 128) * cdiv(g.N, 128) * nz;
  const long b12864 = cdiv(g.M, 128) * cdiv(g.N, 64) * nz;
  if (b128 >= 384 && g.N > 64) return launch_cfg<128, 128, 64, 64>(g, a_vec, b_vec, s);
  if (b12864 >= 256 && g.N > 32) return launch_cfg<128, 64, 64, 32>(g, a_vec, b_vec, s);
  return launch_cfg<64, 64, 32, 32>(g, a_vec, b_vec, s);
}

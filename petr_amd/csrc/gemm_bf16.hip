// General bf16 contraction (PETR_GEMM_BF16 with any operand layout):
//   C[m,n] (+)= act(alpha * sum_k bf16(A(m,k)) * bf16(B(n,k)) + bias + R), fp32 accumulate on v_mfma_f32_32x32x16_bf16.
//
// The training step of BASELINE configs 3-5 (bf16) needs the token-sized (L = 12 000 .. 24 000 rows) GRADIENT
// contractions on the bf16 matrix cores as well: input gradients (B = a weight read K-major), weight gradients
// (A = dY^T and B = X both K-major, K = B*L cut into segments, split over workgroups, float-atomic accumulation,
// bias gradient as column sums) - the same contractions the reference gets from autograd of its 1x1 convs /
// nn.Linear layers (petr_head.py:220-274, petr_transformer.py:357-362) under autocast.  gemm.hip's two bf16 kernels
// only take a K-contiguous B and a plain epilogue; this one takes every layout combination of petr_gemm_args.
//
//   workgroup = 128 x 128 outputs, 4 waves of 64 x 64 (2 x 2 MFMA tiles), K step 32, 256 threads;
//   LDS image of both operands: [row][32 k] bf16 at an 80-byte pitch (16-byte fragment reads conflict-free);
//     K-contiguous operand: a thread stages 4 float4 (8 threads = one 128-byte row piece), rounds, 8-byte LDS stores;
//     K-major operand (x[k*ld + row]): a thread stages ONE row and 16 consecutive k with dword loads that are
//       coalesced across the lanes (consecutive rows), rounds, two 16-byte LDS stores - no transposition anywhere;
//   register prefetch of the next K step, double-buffered LDS, one barrier per step;
//   loads are unconditional from clamped addresses, the zero-fill select happens at the LDS store and only on ragged
//   tiles (wave-uniform test), so interior tiles carry no per-element VALU besides the rounding;
//   K segments (k_seg), K slices over workgroups (split_k: slabs or float atomics), two batch dims, epilogue: bias,
//   residual, ReLU, ReLU-mask, accumulate, bf16 store; a_colsum (bias gradient) from the fp32 staging registers of a
//   K-major A, i.e. the exact fp32 column sums, not sums of rounded values.
#include <stdlib.h>

#include "common.h"

namespace {

typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x4 __attribute__((ext_vector_type(4)));

constexpr int GB_BM = 128, GB_BN = 128, GB_BK = 32, GB_PITCH = 40;

__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  hbf16x4 o = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
  return __builtin_bit_cast(uint2, o);
}

// one operand tile (128 rows x 32 k) in flight between global memory and LDS
template <bool KC>
struct Stage16 {
  float v[16];            // KC: 4 float4 (row = (t>>3) + 32 i, k = 4 (t&7) ..+3);  KM: row t&127, k = 16 (t>>7) + e
  // loop-invariant addressing
  int off[4];             // KC: element offset of float4 i from the tile origin (row clamped); KM: off[0] = clamped row
  unsigned rowok;         // KC: bit i = row i valid; KM: bit 0
  __device__ __forceinline__ void init(long ld, int row0, int rows) {
    const int t = threadIdx.x & 255;        // the staging role is played by 256 threads (all of them, or waves 4..7)
    rowok = 0;
    if (KC) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + (t >> 3) + 32 * i;
        off[i] = min(row, rows - 1) * (int)ld + 4 * (t & 7);
        rowok |= (row < rows ? 1u : 0u) << i;
      }
    } else {
      const int row = row0 + (t & 127);
      off[0] = min(row, rows - 1);
      off[1] = off[2] = off[3] = 0;
      rowok = row < rows ? 1u : 0u;
    }
  }
  // base = operand + batch + segment; k0 = first k of the tile inside the segment, kend = segment length
  __device__ __forceinline__ void load(const float* base, long ld, int k0, int kend) {
    const int t = threadIdx.x & 255;
    if (KC) {
      const int kc = min(k0 + 4 * (t & 7), kend - 4) - 4 * (t & 7);    // clamp keeps the float4 inside the row
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 x = *reinterpret_cast<const float4*>(base + off[i] + kc);
        v[4 * i] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
      }
    } else {
      const int kb = k0 + 16 * (t >> 7);
      if (kb + 16 <= kend) {                      // wave-uniform: all 16 k inside the segment
        const float* src = base + (long)kb * ld + off[0];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = src[(long)e * ld];
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = base[(long)min(kb + e, kend - 1) * ld + off[0]];
      }
    }
  }
  // round to bf16 and write the [row][k] image; `ragged`: the tile crosses the operand's row or K bound (uniform)
  template <int PITCH = GB_PITCH>
  __device__ __forceinline__ void store(uint16_t* img, bool ragged, int k0, int kend) {
    const int t = threadIdx.x & 255;
    if (KC) {
      if (ragged) {
        const bool kok = k0 + 4 * (t & 7) < kend;       // K % 4 == 0: a float4 is entirely inside or outside
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool ok = kok && ((rowok >> i) & 1u);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * i + e] = ok ? v[4 * i + e] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<uint2*>(img + ((t >> 3) + 32 * i) * PITCH + 4 * (t & 7)) =
            pack4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
    } else {
      if (ragged) {
        const int kb = k0 + 16 * (t >> 7);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = (rowok && kb + e < kend) ? v[e] : 0.f;
      }
      uint4* d = reinterpret_cast<uint4*>(img + (t & 127) * PITCH + 16 * (t >> 7));
      const uint2 p0 = pack4(v[0], v[1], v[2], v[3]), p1 = pack4(v[4], v[5], v[6], v[7]);
      const uint2 p2 = pack4(v[8], v[9], v[10], v[11]), p3 = pack4(v[12], v[13], v[14], v[15]);
      d[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
      d[1] = make_uint4(p2.x, p2.y, p3.x, p3.y);
    }
  }
};

// the same tile when the operand is ALREADY bf16 in memory (PETR_GEMM_A_BF16 / PETR_GEMM_B_BF16: activations that a
// producing epilogue stored as bf16 - dK / dV of the attention backward, the position-embedding hidden layers and their
// gradients): half the bytes, no rounding pass.  Element strides (ld, batch, segment) count bf16 elements.
//   K-contiguous: 2 pieces of 16 bytes (8 k) per thread, piece = t + 256 i -> row piece >> 2, k = 8 (piece & 3);
//   K-major: row t & 127, 16 consecutive k as 2-byte loads coalesced across the lanes.
template <bool KC>
struct Stage16B {
  uint4 x[2];
  int off[2];
  unsigned rowok;
  __device__ __forceinline__ void init(long ld, int row0, int rows) {
    const int t = threadIdx.x & 255;
    rowok = 0;
    if (KC) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pc = t + 256 * i;
        const int row = row0 + (pc >> 2);
        off[i] = min(row, rows - 1) * (int)ld + 8 * (pc & 3);
        rowok |= (row < rows ? 1u : 0u) << i;
      }
    } else {
      const int row = row0 + (t & 127);
      off[0] = min(row, rows - 1);
      off[1] = 0;
      rowok = row < rows ? 1u : 0u;
    }
  }
  __device__ __forceinline__ void load(const float* base_f, long ld, int k0, int kend) {
    const uint16_t* base = reinterpret_cast<const uint16_t*>(base_f);
    const int t = threadIdx.x & 255;
    if (KC) {
      const int kq = 8 * (t & 3);
      const int kc = min(k0 + kq, kend - 8) - kq;                     // K % 8 == 0: the piece stays inside the row
#pragma unroll
      for (int i = 0; i < 2; ++i) x[i] = *reinterpret_cast<const uint4*>(base + off[i] + kc);
    } else {
      const int kb = k0 + 16 * (t >> 7);
      unsigned short e16[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) e16[e] = base[(long)min(kb + e, kend - 1) * ld + off[0]];
      x[0] = make_uint4(e16[0] | (e16[1] << 16), e16[2] | (e16[3] << 16), e16[4] | (e16[5] << 16), e16[6] | (e16[7] << 16));
      x[1] = make_uint4(e16[8] | (e16[9] << 16), e16[10] | (e16[11] << 16), e16[12] | (e16[13] << 16), e16[14] | (e16[15] << 16));
    }
  }
  template <int PITCH = GB_PITCH>
  __device__ __forceinline__ void store(uint16_t* img, bool ragged, int k0, int kend) {
    const int t = threadIdx.x & 255;
    if (KC) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pc = t + 256 * i;
        uint4 v = x[i];
        if (ragged && !(((rowok >> i) & 1u) && k0 + 8 * (pc & 3) < kend)) v = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(img + (pc >> 2) * PITCH + 8 * (pc & 3)) = v;
      }
    } else {
      uint4 a = x[0], b = x[1];
      if (ragged) {
        const int kb = k0 + 16 * (t >> 7);
        uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const uint32_t lo = (rowok && kb + 2 * j < kend) ? (w[j] & 0xFFFFu) : 0u;
          const uint32_t hi = (rowok && kb + 2 * j + 1 < kend) ? (w[j] & 0xFFFF0000u) : 0u;
          w[j] = lo | hi;
        }
        a = make_uint4(w[0], w[1], w[2], w[3]);
        b = make_uint4(w[4], w[5], w[6], w[7]);
      }
      uint4* d = reinterpret_cast<uint4*>(img + (t & 127) * PITCH + 16 * (t >> 7));
      d[0] = a;
      d[1] = b;
    }
  }
  // sum of this thread's 16 k of its row (K-major only): the bias-gradient column sum, from the bf16 values
  __device__ __forceinline__ float ksum(int k0, int kend) const {
    const int kb = k0 + 16 * ((threadIdx.x & 255) >> 7);
    const uint32_t w[8] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w};
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s += (kb + 2 * j < kend) ? __uint_as_float(w[j] << 16) : 0.f;
      s += (kb + 2 * j + 1 < kend) ? __uint_as_float(w[j] & 0xFFFF0000u) : 0.f;
    }
    return s;
  }
};

template <bool KC, bool S16> struct StageSel { typedef Stage16<KC> type; };
template <bool KC> struct StageSel<KC, true> { typedef Stage16B<KC> type; };

template <bool AKC, bool BKC, bool A16 = false, bool B16 = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_gen_kernel(const petr_gemm_args g, const int tiles_m, const int tiles_n) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2][(GB_BM + GB_BN) * GB_PITCH];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int h = lane >> 5, c = lane & 31;
  // 1-D grid, (batch, K slice) index fastest: the workgroups that share an operand panel (the same activation rows for
  // every layer's weights, the same K slice of the input for every layer's weight gradient) run side by side on one XCD
  // and find it in that XCD's L2
  const int zn = g.nb0 * g.nb1 * g.split_k;
  const int lin = xcd_remap(blockIdx.x, tiles_m * tiles_n * zn);
  const int tile = lin / zn;
  const int tm_i = tile / tiles_n, tn_i = tile - tm_i * tiles_n;
  const int m0 = tm_i * GB_BM, n0 = tn_i * GB_BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  int z = lin - tile * zn;
  const int ks = z % g.split_k;
  z /= g.split_k;
  const int z1 = z % g.nb1, z0 = z / g.nb1;
  // operand bases as byte addresses: element strides count 4-byte floats or 2-byte bf16 values
  constexpr int AE = A16 ? 2 : 4, BE = B16 ? 2 : 4;
  const char* Ab = reinterpret_cast<const char*>(g.a) + (z0 * g.a_bs0 + z1 * g.a_bs1) * AE;
  const char* Bb = reinterpret_cast<const char*>(g.b) + (z0 * g.b_bs0 + z1 * g.b_bs1) * BE;

  const int kseg = g.k_seg > 0 ? g.k_seg : g.K;
  const int nseg = g.k_seg > 0 ? g.K / g.k_seg : 1;
  const int tps = (kseg + GB_BK - 1) / GB_BK;
  const int ktiles = nseg * tps;
  const int kt_per = (ktiles + g.split_k - 1) / g.split_k;
  const int kt_begin = ks * kt_per;
  const int kt_end = min(ktiles, kt_begin + kt_per);

  typename StageSel<AKC, A16>::type sa;
  typename StageSel<BKC, B16>::type sb;
  sa.init(g.lda, m0, g.M);
  sb.init(g.ldb, n0, g.N);
  const bool rows_ragged_a = m0 + GB_BM > g.M, rows_ragged_b = n0 + GB_BN > g.N;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool do_colsum = g.a_colsum != nullptr && tn_i == 0;    // host guarantees a K-major A with a_colsum
  float colacc = 0.f;

  int k0_cur = 0, k0_nxt = 0;       // first k (inside its segment) of the tile held in registers / of the one being loaded
  auto gload = [&](int kt, int& k0_out) {
    const int seg = kt / tps;
    const int k0 = (kt - seg * tps) * GB_BK;
    k0_out = k0;
    sa.load(reinterpret_cast<const float*>(Ab + (long)seg * g.a_seg_stride * AE), g.lda, k0, kseg);
    sb.load(reinterpret_cast<const float*>(Bb + (long)seg * g.b_seg_stride * BE), g.ldb, k0, kseg);
  };
  auto lstore = [&](int buf, int k0) {
    const bool kr = k0 + GB_BK > kseg;
    if (!AKC && do_colsum) {        // column sums of A (= dC^T): the bias gradient of a weight-gradient contraction,
      if constexpr (A16) {          // from the fp32 staging values (exact), or from the bf16 values of a bf16 operand
        colacc += sa.ksum(k0, kseg);
      } else {
        const int kb = k0 + 16 * (t >> 7);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) s += (kb + e < kseg) ? sa.v[e] : 0.f;
        colacc += s;
      }
    }
    sa.store(lds[buf], rows_ragged_a || kr, k0, kseg);
    sb.store(lds[buf] + GB_BM * GB_PITCH, rows_ragged_b || kr, k0, kseg);
  };

  const int nk = kt_end - kt_begin;
  if (nk > 0) {
    gload(kt_begin, k0_cur);
    lstore(0, k0_cur);
    if (nk > 1) gload(kt_begin + 1, k0_nxt);
    for (int it = 0; it < nk; ++it) {
      const int buf = it & 1;
      __syncthreads();                       // image `buf` complete; the other one is free again
      if (it + 1 < nk) {
        lstore(buf ^ 1, k0_nxt);
        if (it + 2 < nk) gload(kt_begin + it + 2, k0_nxt);
      }
      const uint16_t* As = lds[buf];
      const uint16_t* Bs = As + GB_BM * GB_PITCH;
#pragma unroll
      for (int j = 0; j < 2; ++j) {          // two 16-deep chunks of the K step
        uint4 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[i] = *reinterpret_cast<const uint4*>(As + (wm + 32 * i + c) * GB_PITCH + 16 * j + 8 * h);
          fb[i] = *reinterpret_cast<const uint4*>(Bs + (wn + 32 * i + c) * GB_PITCH + 16 * j + 8 * h);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jn = 0; jn < 2; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hbf16x8, fa[i]),
                                                                 __builtin_bit_cast(hbf16x8, fb[jn]), acc[i][jn], 0, 0, 0);
      }
    }
  }

  if (!AKC && do_colsum && m0 + (t & 127) < g.M)
    atomicAdd(g.a_colsum + z0 * g.cs_bs0 + z1 * g.cs_bs1 + m0 + (t & 127), colacc);

  // ---- epilogue (as gemm_kernel) ----
  const bool atomic = (g.flags & PETR_GEMM_ATOMIC) != 0;
  const bool plain = g.split_k > 1 && !atomic;       // K slices store raw partial sums
  float* C = g.c + z0 * g.c_bs0 + z1 * g.c_bs1 + (atomic ? 0 : (long)ks * g.c_split_stride);
  const float* bias = (!plain && g.bias) ? g.bias + z0 * g.bias_bs0 + z1 * g.bias_bs1 : nullptr;
  const float* R = (!plain && g.r) ? g.r + z0 * g.r_bs0 + z1 * g.r_bs1 : nullptr;
  const int flags = plain ? 0 : g.flags;
#pragma unroll
  for (int jn = 0; jn < 2; ++jn) {
    const int n = n0 + wn + 32 * jn + c;
    const int nc = min(n, g.N - 1);
    const float bv = bias ? bias[nc] : 0.f;
    const long ccol = g.c_nblk > 0 ? (long)(nc / g.c_nblk) * g.c_nblk_stride + (nc % g.c_nblk) : (long)nc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + 32 * i + mfma32_row(r, h);
        const int mc = min(m, g.M - 1);
        const bool ok = m < g.M && n < g.N;
        float v = acc[i][jn][r] * g.alpha + bv;
        float* dst = C + (long)mc * g.ldc + ccol;
        float rv = 0.f;
        if (R) {
          if (flags & PETR_GEMM_R_BF16)
            rv = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(g.r)[z0 * g.r_bs0 + z1 * g.r_bs1 + (long)mc * g.ldr + nc] << 16);
          else rv = R[(long)mc * g.ldr + nc];
        }
        const float old = (flags & PETR_GEMM_ACCUMULATE) ? *dst : 0.f;
        if (flags & PETR_GEMM_RELU_MASK) v = rv > 0.f ? v : 0.f;
        else v += rv;
        if (flags & PETR_GEMM_RELU) v = fmaxf(v, 0.f);
        v += old;
        if (ok) {
          if (flags & PETR_GEMM_STORE_BF16)
            reinterpret_cast<uint16_t*>(g.c)[z0 * g.c_bs0 + z1 * g.c_bs1 + (long)mc * g.ldc + ccol] =
                __builtin_bit_cast(uint16_t, (__bf16)v);
          else if (atomic) atomicAdd(dst, v);
          else *dst = v;
        }
      }
  }
}

// ---- K-major operand tiles kept in their memory layout [k][row], fragments through ds_read_b64_tr_b16 (see
//      gemm_bf16_km_kernel below for the full story) ----
constexpr int KM_PITCH = 144;
typedef short km_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ km_s16x4 km_tr16(const uint16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((km_s16x4 __attribute__((address_space(3)))*)p);
}
__device__ __forceinline__ hbf16x8 km_cat8(km_s16x4 lo, km_s16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(hbf16x8, v);
}

// 16 consecutive rows of one k, as they come from memory (S16: bf16, else fp32)
template <bool S16>
struct KmStage {
  uint4 x[S16 ? 2 : 4];
  __device__ __forceinline__ void load(const char* base, long ld, int k, int row) {      // base in bytes, ld / row in elements
    constexpr int E = S16 ? 2 : 4;
    const uint4* src = reinterpret_cast<const uint4*>(base + ((long)k * ld + row) * E);
#pragma unroll
    for (int i = 0; i < (S16 ? 2 : 4); ++i) x[i] = src[i];
  }
  __device__ __forceinline__ void store(uint16_t* dst, bool ok) const {
    uint4 a, b;
    if (S16) {
      a = x[0]; b = x[1];
    } else {
      const float4 f0 = __builtin_bit_cast(float4, x[0]), f1 = __builtin_bit_cast(float4, x[1]);
      const float4 f2 = __builtin_bit_cast(float4, x[2 % (S16 ? 2 : 4)]), f3 = __builtin_bit_cast(float4, x[3 % (S16 ? 2 : 4)]);
      const uint2 p0 = pack4(f0.x, f0.y, f0.z, f0.w), p1 = pack4(f1.x, f1.y, f1.z, f1.w);
      const uint2 p2 = pack4(f2.x, f2.y, f2.z, f2.w), p3 = pack4(f3.x, f3.y, f3.z, f3.w);
      a = make_uint4(p0.x, p0.y, p1.x, p1.y);
      b = make_uint4(p2.x, p2.y, p3.x, p3.y);
    }
    if (!ok) a = b = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(dst)[0] = a;
    reinterpret_cast<uint4*>(dst)[1] = b;
  }
  // the 16 values as floats (column sums)
  __device__ __forceinline__ void add_to(float (&s)[16]) const {
    if (S16) {
      const uint32_t w[8] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w};
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[2 * j] += __uint_as_float(w[j] << 16); s[2 * j + 1] += __uint_as_float(w[j] & 0xFFFF0000u); }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 f = __builtin_bit_cast(float4, x[i % (S16 ? 2 : 4)]);
        s[4 * i] += f.x; s[4 * i + 1] += f.y; s[4 * i + 2] += f.z; s[4 * i + 3] += f.w;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------
// Deep-step version of the kernel above: K step 128 instead of 32, the whole step of both operands in flight at once.
//
// Why: the token-sized contractions of the bf16 step have SHORT contractions (K = 192 .. 384 for the 1x1 convolutions and
// the K/V projections, 1024 / 1536 for two of them) over 12 000 .. 24 000 rows.  With K steps of 32 a workgroup spent
// ~2.4 us per step waiting for one memory round trip and 0.2 us multiplying (measured on the K/V projections): eight
// exposed latencies per 128 x 128 tile, 270 TFLOP/s and 1.4 TB/s on a kernel whose HBM floor is 5x lower.  Here one step
// carries 4x the bytes (the same per-thread staging pattern, four 32-k groups issued back to back before anything waits),
// so a K = 256 tile pays two round trips; the LDS image [256 rows][128 k] bf16 (pitch 136: conflict-free 16-byte fragment
// reads) is single-buffered - 68 KB, two workgroups per CU - and the next step's loads are issued before the products of
// the current one.
// Epilogue through LDS: the accumulators go to a [128][132] fp32 image (the operand image is free by then), each wave reads
// its own 64 x 64 block back row-wise and applies bias / residual / mask / activation on float4s: 16-byte loads of the
// residual or mask operand and 16-byte (fp32) or 8-byte (bf16) stores instead of 64 two- or four-byte ones per lane.
// ---------------------------------------------------------------------------------------------------------------
constexpr int GD_BK = 128, GD_PITCH = 136, GD_CP = 132;
constexpr size_t GD_LDS_BYTES = (size_t)(GB_BM + GB_BN) * GD_PITCH * 2;      // 69 632 >= 128 * 132 * 4 = 67 584

// TR: a K-major operand (x[k * ld + row]: the NCHW maps of the forward's first convolutions, the weights of every input gradient)
// keeps its memory layout in LDS - [128 k][rows] at pitch 144, staged with 16-byte loads of 16 consecutive rows - and its
// fragments are transposed reads (ds_read_b64_tr_b16); the K-contiguous operand next to it is then read as two 8-byte pieces
// per chunk, at the k the transposed read delivers (4 h .. and 8 + 4 h ..).  Without TR such a tile is staged with sixteen 2- /
// 4-byte loads per thread and 32-k group.  Needs 16-byte aligned rows in multiples of 16 (host check); no a_colsum.
template <bool AKC, bool BKC, bool A16, bool B16, bool TR = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_deep_kernel(const petr_gemm_args g, const int tiles_m, const int tiles_n,
                                                                const int vec_epi) {
  constexpr bool AT = TR && !AKC, BT = TR && !BKC, PERM = AT || BT;
  constexpr int A_IMG = AT ? GD_BK * KM_PITCH : GB_BM * GD_PITCH;
  extern __shared__ __attribute__((aligned(16))) uint16_t dlds[];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int h = lane >> 5, c = lane & 31;
  // 1-D grid, (batch, K slice) index fastest: the workgroups that share an operand panel (the same activation rows for
  // every layer's weights, the same K slice of the input for every layer's weight gradient) run side by side on one XCD
  // and find it in that XCD's L2
  const int zn = g.nb0 * g.nb1 * g.split_k;
  const int lin = xcd_remap(blockIdx.x, tiles_m * tiles_n * zn);
  const int tile = lin / zn;
  const int tm_i = tile / tiles_n, tn_i = tile - tm_i * tiles_n;
  const int m0 = tm_i * GB_BM, n0 = tn_i * GB_BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  int z = lin - tile * zn;
  const int ks = z % g.split_k;
  z /= g.split_k;
  const int z1 = z % g.nb1, z0 = z / g.nb1;
  constexpr int AE = A16 ? 2 : 4, BE = B16 ? 2 : 4;
  const char* Ab = reinterpret_cast<const char*>(g.a) + (z0 * g.a_bs0 + z1 * g.a_bs1) * AE;
  const char* Bb = reinterpret_cast<const char*>(g.b) + (z0 * g.b_bs0 + z1 * g.b_bs1) * BE;

  const int kseg = g.k_seg > 0 ? g.k_seg : g.K;
  const int nseg = g.k_seg > 0 ? g.K / g.k_seg : 1;
  const int tps = (kseg + GD_BK - 1) / GD_BK;
  const int ktiles = nseg * tps;
  const int kt_per = (ktiles + g.split_k - 1) / g.split_k;
  const int kt_begin = ks * kt_per;
  const int kt_end = min(ktiles, kt_begin + kt_per);

  typename StageSel<AKC, A16>::type sa[4];
  typename StageSel<BKC, B16>::type sb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {         // identical loop-invariant addressing: one copy survives
    sa[q].init(g.lda, m0, g.M);
    sb[q].init(g.ldb, n0, g.N);
  }
  const bool rows_ragged_a = m0 + GB_BM > g.M, rows_ragged_b = n0 + GB_BN > g.N;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool do_colsum = g.a_colsum != nullptr && tn_i == 0;    // host guarantees a K-major A with a_colsum
  float colacc = 0.f;

  uint16_t* As = dlds;
  uint16_t* Bs = dlds + A_IMG;
  // transposed-read staging role: k = t >> 3 of a 32-k group, rows 16 (t & 7) .. + 15 of the tile (clamped to a valid chunk,
  // zero-filled at the store when the chunk lies beyond the operand's rows)
  const int sk = t >> 3, sr = 16 * (t & 7);
  KmStage<A16> sat[AT ? 4 : 1];
  KmStage<B16> sbt[BT ? 4 : 1];
  const bool arow_ok = m0 + sr < g.M, brow_ok = n0 + sr < g.N;
  const int arow = arow_ok ? m0 + sr : 0, brow = brow_ok ? n0 + sr : 0;
  const int tr_q = (lane & 15) >> 2;
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  int k0_cur = 0, k0_nxt = 0;
  auto gload = [&](int kt, int& k0_out) {
    const int seg = kt / tps;
    const int k0 = (kt - seg * tps) * GD_BK;
    k0_out = k0;
    const float* ab = reinterpret_cast<const float*>(Ab + (long)seg * g.a_seg_stride * AE);
    const float* bb = reinterpret_cast<const float*>(Bb + (long)seg * g.b_seg_stride * BE);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (k0 + 32 * q < kseg) {           // uniform: 32-k groups beyond the segment are neither loaded nor multiplied
        if constexpr (AT) sat[q].load(reinterpret_cast<const char*>(ab), g.lda, min(k0 + 32 * q + sk, kseg - 1), arow);
        else sa[q].load(ab, g.lda, k0 + 32 * q, kseg);
        if constexpr (BT) sbt[q].load(reinterpret_cast<const char*>(bb), g.ldb, min(k0 + 32 * q + sk, kseg - 1), brow);
        else sb[q].load(bb, g.ldb, k0 + 32 * q, kseg);
      }
  };
  auto lstore = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kq = k0 + 32 * q;
      if (kq < kseg) {
        const bool kr = kq + 32 > kseg;
        if constexpr (AT) sat[q].store(As + (32 * q + sk) * KM_PITCH + sr, arow_ok && kq + sk < kseg);
        if constexpr (BT) sbt[q].store(Bs + (32 * q + sk) * KM_PITCH + sr, brow_ok && kq + sk < kseg);
        if (!AKC && !AT && do_colsum) {
          if constexpr (A16) {
            colacc += sa[q].ksum(kq, kseg);
          } else {
            const int kb = kq + 16 * (t >> 7);
            float sum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += (kb + e < kseg) ? sa[q].v[e] : 0.f;
            colacc += sum;
          }
        }
        if constexpr (!AT) sa[q].template store<GD_PITCH>(As + 32 * q, rows_ragged_a || kr, kq, kseg);
        if constexpr (!BT) sb[q].template store<GD_PITCH>(Bs + 32 * q, rows_ragged_b || kr, kq, kseg);
      }
    }
  };

  const int nk = kt_end - kt_begin;
  if (nk > 0) {
    gload(kt_begin, k0_cur);
    for (int it = 0; it < nk; ++it) {
      if (it > 0) __syncthreads();           // every wave is done with the previous step's image
      lstore(k0_cur);
      __syncthreads();
      const int chunks = min(GD_BK / 16, (kseg - k0_cur + 15) >> 4);
      if (it + 1 < nk) {
        gload(kt_begin + it + 1, k0_nxt);
        k0_cur = k0_nxt;
      }
      for (int j = 0; j < chunks; ++j) {     // 16-deep chunks of the K step
        hbf16x8 fa[2], fb[2];
        // one operand's fragment for rows r0 .. r0 + 31: transposed read of a [k][row] image, or a [row][k] image read at the
        // plain (8 h) or the permuted (4 h | 8 + 4 h) k positions
        auto frag = [&](const uint16_t* img, int r0, auto tr_c) -> hbf16x8 {
          if constexpr (decltype(tr_c)::value) {
            const uint16_t* p = img + (16 * j + 4 * h + tr_q) * KM_PITCH + tr_col + r0;
            return km_cat8(km_tr16(p), km_tr16(p + 8 * KM_PITCH));
          } else if constexpr (PERM) {
            const uint16_t* p = img + (r0 + c) * GD_PITCH + 16 * j + 4 * h;
            const uint2 lo = *reinterpret_cast<const uint2*>(p), hi = *reinterpret_cast<const uint2*>(p + 8);
            return __builtin_bit_cast(hbf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
          } else {
            return __builtin_bit_cast(hbf16x8, *reinterpret_cast<const uint4*>(img + (r0 + c) * GD_PITCH + 16 * j + 8 * h));
          }
        };
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[i] = frag(As, wm + 32 * i, std::integral_constant<bool, AT>{});
          fb[i] = frag(Bs, wn + 32 * i, std::integral_constant<bool, BT>{});
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jn], acc[i][jn], 0, 0, 0);
      }
    }
  }

  if (!AKC && do_colsum && m0 + (t & 127) < g.M)
    atomicAdd(g.a_colsum + z0 * g.cs_bs0 + z1 * g.cs_bs1 + m0 + (t & 127), colacc);

  // ---- epilogue ----
  const bool atomic = (g.flags & PETR_GEMM_ATOMIC) != 0;
  const bool plain = g.split_k > 1 && !atomic;       // K slices store raw partial sums
  const long cz = z0 * g.c_bs0 + z1 * g.c_bs1 + (atomic ? 0 : (long)ks * g.c_split_stride);
  float* C = g.c + cz;
  const float* bias = (!plain && g.bias) ? g.bias + z0 * g.bias_bs0 + z1 * g.bias_bs1 : nullptr;
  const float* R = (!plain && g.r) ? g.r + z0 * g.r_bs0 + z1 * g.r_bs1 : nullptr;
  const int flags = plain ? 0 : g.flags;
  DropDev dd;                      // dropout of the result (the FFN hidden of the bf16 step); thr == 0: off
  __builtin_memcpy(&dd, &g.drop, sizeof dd);
  if (plain) dd.thr = 0;
  if (vec_epi) {
    float* Cs = reinterpret_cast<float*>(dlds);
    __syncthreads();                         // the operand image is dead
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jn = 0; jn < 2; ++jn)
#pragma unroll
        for (int r = 0; r < 16; ++r) Cs[(wm + 32 * i + mfma32_row(r, h)) * GD_CP + wn + 32 * jn + c] = acc[i][jn][r];
    // each wave reads back only what it wrote (its 64 x 64 block): LDS operations of one wave complete in order
    const int col = wn + 4 * (lane & 15);
    const int n = n0 + col;
    if (n < g.N) {                           // N % 4 == 0: a float4 is entirely inside or outside
      const long ccol = g.c_nblk > 0 ? (long)(n / g.c_nblk) * g.c_nblk_stride + (n % g.c_nblk) : (long)n;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias) bv = *reinterpret_cast<const float4*>(bias + n);
      const uint16_t* R16 = reinterpret_cast<const uint16_t*>(g.r) + z0 * g.r_bs0 + z1 * g.r_bs1;
      uint16_t* C16 = reinterpret_cast<uint16_t*>(g.c) + cz;
#pragma unroll 4
      for (int rr = 0; rr < 16; ++rr) {
        const int row = wm + 4 * rr + (lane >> 4);
        const int m = m0 + row;
        if (m >= g.M) continue;
        const float4 a4 = *reinterpret_cast<const float4*>(Cs + row * GD_CP + col);
        float v[4] = {a4.x * g.alpha + bv.x, a4.y * g.alpha + bv.y, a4.z * g.alpha + bv.z, a4.w * g.alpha + bv.w};
        float rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (R) {
          if (flags & PETR_GEMM_R_BF16) {
            const uint2 u = *reinterpret_cast<const uint2*>(R16 + (long)m * g.ldr + n);
            rv[0] = __uint_as_float(u.x << 16); rv[1] = __uint_as_float(u.x & 0xFFFF0000u);
            rv[2] = __uint_as_float(u.y << 16); rv[3] = __uint_as_float(u.y & 0xFFFF0000u);
          } else {
            const float4 u = *reinterpret_cast<const float4*>(R + (long)m * g.ldr + n);
            rv[0] = u.x; rv[1] = u.y; rv[2] = u.z; rv[3] = u.w;
          }
        }
        float* dst = C + (long)m * g.ldc + ccol;
        float old[4] = {0.f, 0.f, 0.f, 0.f};
        if (flags & PETR_GEMM_ACCUMULATE) {
          const float4 u = *reinterpret_cast<const float4*>(dst);
          old[0] = u.x; old[1] = u.y; old[2] = u.z; old[3] = u.w;
        }
        const uint32_t rk = dd.thr ? drop_row_key(dd, (uint32_t)m) : 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (flags & PETR_GEMM_RELU_MASK) v[e] = rv[e] > 0.f ? v[e] : 0.f;
          else v[e] += rv[e];
          if (flags & PETR_GEMM_RELU) v[e] = fmaxf(v[e], 0.f);
          if (dd.thr) v[e] = drop_keep(rk, (uint32_t)(n + e), dd.thr) ? v[e] * dd.scale : 0.f;
          v[e] += old[e];
        }
        if (flags & PETR_GEMM_STORE_BF16) {
          *reinterpret_cast<uint2*>(C16 + (long)m * g.ldc + ccol) = pack4(v[0], v[1], v[2], v[3]);
        } else if (atomic) {
#pragma unroll
          for (int e = 0; e < 4; ++e) atomicAdd(dst + e, v[e]);
        } else {
          *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int jn = 0; jn < 2; ++jn) {
    const int n = n0 + wn + 32 * jn + c;
    const int nc = min(n, g.N - 1);
    const float bv = bias ? bias[nc] : 0.f;
    const long ccol = g.c_nblk > 0 ? (long)(nc / g.c_nblk) * g.c_nblk_stride + (nc % g.c_nblk) : (long)nc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + 32 * i + mfma32_row(r, h);
        const int mc = min(m, g.M - 1);
        const bool ok = m < g.M && n < g.N;
        float v = acc[i][jn][r] * g.alpha + bv;
        float* dst = C + (long)mc * g.ldc + ccol;
        float rv = 0.f;
        if (R) {
          if (flags & PETR_GEMM_R_BF16)
            rv = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(g.r)[z0 * g.r_bs0 + z1 * g.r_bs1 + (long)mc * g.ldr + nc] << 16);
          else rv = R[(long)mc * g.ldr + nc];
        }
        const float old = (flags & PETR_GEMM_ACCUMULATE) ? *dst : 0.f;
        if (flags & PETR_GEMM_RELU_MASK) v = rv > 0.f ? v : 0.f;
        else v += rv;
        if (flags & PETR_GEMM_RELU) v = fmaxf(v, 0.f);
        if (dd.thr) v = drop_keep(drop_row_key(dd, (uint32_t)mc), (uint32_t)nc, dd.thr) ? v * dd.scale : 0.f;
        v += old;
        if (ok) {
          if (flags & PETR_GEMM_STORE_BF16)
            reinterpret_cast<uint16_t*>(g.c)[cz + (long)mc * g.ldc + ccol] = __builtin_bit_cast(uint16_t, (__bf16)v);
          else if (atomic) atomicAdd(dst, v);
          else *dst = v;
        }
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Both operands K-major (x[k * ld + row]): the weight gradients over all tokens - dW = dY^T X with dY [tokens][out] and
// X [tokens][in] exactly as they lie in memory (dW_k / dW_v of the K / V projections, the second position-embedding
// convolutions).  The kernel above stages such a tile with sixteen 2- or 4-byte loads per thread and operand and K step
// (one element of sixteen consecutive k for one row) - 130-215 TFLOP/s on 24 000-token shapes whose HBM floor is 6-8x lower.
// Here the LDS image KEEPS the memory layout, [k][row] at a pitch of 144 elements: a thread stages 16 consecutive rows of ONE k
// with 16-byte loads (two for a bf16 source, four for fp32) and two 16-byte LDS stores, and the MFMA fragments - 8 consecutive k
// of one row per lane - come out of that image through ds_read_b64_tr_b16 (the transposing LDS read of gfx950; the lane / k
// mapping is the one mha_bwd_bf16.hip uses for its dV / dK products, identical for both operands, so the permuted k order
// inside a 16-deep block cancels).  Pitch 144: the four rows of a 4 x 16 transposed block fall 8 banks apart.
// Needs M, N multiples of 128, 16-byte aligned rows, PETR_GEMM_ATOMIC accumulation (K slices over workgroups); a_colsum as in
// the general kernel.  Everything else (K segments, two batch dims) follows petr_gemm_args.
// ---------------------------------------------------------------------------------------------------------------
// BKC: the B operand is K-CONTIGUOUS instead (x[row * ld + k]: an NCHW map read as [channel][pixel] - the first convolutions'
// and input_proj's weight gradients, dW = dY^T X with X the [view][c][hw] planes).  Its tile is staged like the general
// kernel's ([row][32 k] image, pitch 40) and its fragments are read as TWO 8-byte pieces per 16-deep chunk, at k = 4 h and
// 8 + 4 h - the k order the transposed reads of A produce; N may be ragged (192 = 3 D channels).
template <bool A16, bool B16, bool BKC>
__global__ __launch_bounds__(256, 2) void gemm_bf16_km_kernel(const petr_gemm_args g, const int tiles_m, const int tiles_n) {
  constexpr int B_IMG = BKC ? GB_BN * GB_PITCH : GB_BK * KM_PITCH;                  // elements of a B image
  __shared__ __attribute__((aligned(16))) uint16_t lds[2][GB_BK * KM_PITCH + B_IMG];   // [buffer][A image | B image]
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int h = lane >> 5, c = lane & 31;
  const int zn = g.nb0 * g.nb1 * g.split_k;
  const int lin = xcd_remap(blockIdx.x, tiles_m * tiles_n * zn);
  const int tile = lin / zn;
  const int tm_i = tile / tiles_n, tn_i = tile - tm_i * tiles_n;
  const int m0 = tm_i * GB_BM, n0 = tn_i * GB_BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  int z = lin - tile * zn;
  const int ks = z % g.split_k;
  z /= g.split_k;
  const int z1 = z % g.nb1, z0 = z / g.nb1;
  constexpr int AE = A16 ? 2 : 4, BE = B16 ? 2 : 4;
  const char* Ab = reinterpret_cast<const char*>(g.a) + (z0 * g.a_bs0 + z1 * g.a_bs1) * AE;
  const char* Bb = reinterpret_cast<const char*>(g.b) + (z0 * g.b_bs0 + z1 * g.b_bs1) * BE;

  const int kseg = g.k_seg > 0 ? g.k_seg : g.K;
  const int nseg = g.k_seg > 0 ? g.K / g.k_seg : 1;
  const int tps = (kseg + GB_BK - 1) / GB_BK;
  const int ktiles = nseg * tps;
  const int kt_per = (ktiles + g.split_k - 1) / g.split_k;
  const int kt_begin = ks * kt_per;
  const int kt_end = min(ktiles, kt_begin + kt_per);

  // staging role: k = t >> 3 of the step, rows 16 (t & 7) .. + 15 of the tile
  const int sk = t >> 3, sr = 16 * (t & 7);
  KmStage<A16> sa;
  KmStage<B16> sb;
  typename StageSel<true, B16>::type sbk;                  // BKC
  if (BKC) sbk.init(g.ldb, n0, g.N);
  const bool rows_ragged_b = BKC && n0 + GB_BN > g.N;
  bool ok_nxt = true;
  int k0_nxt = 0;
  auto gload = [&](int kt, bool& ok, int& k0_out) {
    const int seg = kt / tps;
    const int k0 = (kt - seg * tps) * GB_BK;
    const int k = k0 + sk;
    k0_out = k0;
    ok = k < kseg;
    const int kc = min(k, kseg - 1);                       // past the segment: a valid row, zero-filled at the store
    sa.load(Ab + (long)seg * g.a_seg_stride * AE, g.lda, kc, m0 + sr);
    if (BKC) sbk.load(reinterpret_cast<const float*>(Bb + (long)seg * g.b_seg_stride * BE), g.ldb, k0, kseg);
    else sb.load(Bb + (long)seg * g.b_seg_stride * BE, g.ldb, kc, n0 + sr);
  };
  const bool do_colsum = g.a_colsum != nullptr && tn_i == 0;
  float colacc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) colacc[e] = 0.f;
  auto lstore = [&](int buf, bool ok, int k0) {
    if (do_colsum && ok) sa.add_to(colacc);
    sa.store(lds[buf] + sk * KM_PITCH + sr, ok);
    if (BKC) sbk.store(lds[buf] + GB_BK * KM_PITCH, rows_ragged_b || k0 + GB_BK > kseg, k0, kseg);
    else sb.store(lds[buf] + GB_BK * KM_PITCH + sk * KM_PITCH + sr, ok);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read lane constants (mha_bwd_bf16.hip): lane l of a 16-lane group supplies row (l & 15) >> 2, columns 4 (l & 3) ..
  // of a 4 x 16 block and receives column l & 15 of its four rows; lanes 16-31 take the next 16 columns, the upper half wave
  // the next 4 k
  const int tr_q = (lane & 15) >> 2;
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  const int nk = kt_end - kt_begin;
  if (nk > 0) {
    bool ok_cur;
    int k0_cur;
    gload(kt_begin, ok_cur, k0_cur);
    lstore(0, ok_cur, k0_cur);
    if (nk > 1) gload(kt_begin + 1, ok_nxt, k0_nxt);
    for (int it = 0; it < nk; ++it) {
      const int buf = it & 1;
      __syncthreads();                       // image `buf` complete; the other one is free again
      if (it + 1 < nk) {
        lstore(buf ^ 1, ok_nxt, k0_nxt);
        if (it + 2 < nk) gload(kt_begin + it + 2, ok_nxt, k0_nxt);
      }
      const uint16_t* As = lds[buf];
      const uint16_t* Bs = As + GB_BK * KM_PITCH;
#pragma unroll
      for (int j = 0; j < 2; ++j) {          // two 16-deep chunks of the K step
        hbf16x8 fa[2], fb[2];
        const int r0 = (16 * j + 4 * h + tr_q) * KM_PITCH + tr_col, r1 = r0 + 8 * KM_PITCH;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          fa[i] = km_cat8(km_tr16(As + r0 + wm + 32 * i), km_tr16(As + r1 + wm + 32 * i));
          if (BKC) {
            const uint16_t* br = Bs + (wn + 32 * i + c) * GB_PITCH + 16 * j + 4 * h;
            const uint2 lo = *reinterpret_cast<const uint2*>(br), hi = *reinterpret_cast<const uint2*>(br + 8);
            fb[i] = __builtin_bit_cast(hbf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
          } else {
            fb[i] = km_cat8(km_tr16(Bs + r0 + wn + 32 * i), km_tr16(Bs + r1 + wn + 32 * i));
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jn], acc[i][jn], 0, 0, 0);
      }
    }
  }

  if (do_colsum) {        // the 32 k-threads of a 16-row group: sum through the (now free) LDS, one atomic per row
    __syncthreads();
    float* cs = reinterpret_cast<float*>(&lds[0][0]);                 // [32 k][128 rows + 4] floats = 16.9 KB
#pragma unroll
    for (int e = 0; e < 16; ++e) cs[sk * 132 + sr + e] = colacc[e];
    __syncthreads();
    if (t < GB_BM) {
      float v = 0.f;
#pragma unroll 8
      for (int k = 0; k < GB_BK; ++k) v += cs[k * 132 + t];
      atomicAdd(g.a_colsum + z0 * g.cs_bs0 + z1 * g.cs_bs1 + m0 + t, v);
    }
  }

  // ---- epilogue: float atomics (M a multiple of 128; N too unless B is K-contiguous) ----
  float* C = g.c + z0 * g.c_bs0 + z1 * g.c_bs1;
#pragma unroll
  for (int jn = 0; jn < 2; ++jn) {
    const int n = n0 + wn + 32 * jn + c;
    if (BKC && n >= g.N) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + 32 * i + mfma32_row(r, h);
        atomicAdd(C + (long)m * g.ldc + n, acc[i][jn][r] * g.alpha);
      }
  }
}
}  // namespace

// called by petr_gemm() for PETR_GEMM_BF16 requests that gemm.hip's plain-epilogue kernels do not take
int petr_gemm_bf16_general(const petr_gemm_args& g, hipStream_t s) {
  const int kseg = g.k_seg > 0 ? g.k_seg : g.K;
  PETR_CHECK(!g.a2 && !(g.flags & PETR_GEMM_SIGMOID_MUL), PETR_ERR_UNSUPPORTED,
             "gemm: PETR_GEMM_BF16 (general) has no a2 addend / sigmoid-gate epilogue");
  PETR_CHECK(!g.a_colsum || !g.a_kcontig, PETR_ERR_UNSUPPORTED, "gemm: PETR_GEMM_BF16 a_colsum needs a K-major A");
  // K-contiguous operands are read as float4 along K, K-major ones as dwords along the rows
  {
    const int am = (g.flags & PETR_GEMM_A_BF16) ? 7 : 3, bm = (g.flags & PETR_GEMM_B_BF16) ? 7 : 3;   // elements per 16 bytes - 1
    PETR_CHECK((!g.a_kcontig || (!(kseg & am) && kseg > am && !(g.lda & am) && !(g.a_bs0 & am) && !(g.a_bs1 & am) &&
                                 !(g.a_seg_stride & am) && aligned16(g.a))) &&
                   (!g.b_kcontig || (!(kseg & bm) && kseg > bm && !(g.ldb & bm) && !(g.b_bs0 & bm) && !(g.b_bs1 & bm) &&
                                     !(g.b_seg_stride & bm) && aligned16(g.b))),
               PETR_ERR_UNSUPPORTED,
               "gemm: PETR_GEMM_BF16 needs 16-byte aligned K-contiguous operands (K %% 4 == 0 for fp32, K %% 8 == 0 for bf16 sources)");
  }
  // 32-bit element offsets inside one batch / segment
  PETR_CHECK((long)(g.a_kcontig ? g.M : kseg) * g.lda + (g.a_kcontig ? kseg : g.M) < (1L << 31) &&
                 (long)(g.b_kcontig ? g.N : kseg) * g.ldb + (g.b_kcontig ? kseg : g.N) < (1L << 31),
             PETR_ERR_UNSUPPORTED, "gemm: PETR_GEMM_BF16 operand batch/segment must span < 2^31 elements");
  const int tm = (int)cdiv(g.M, GB_BM), tn = (int)cdiv(g.N, GB_BN);
  const bool a16 = (g.flags & PETR_GEMM_A_BF16) != 0, b16 = (g.flags & PETR_GEMM_B_BF16) != 0;
  // both operands K-major, full 128 x 128 tiles, 16-byte rows, atomic accumulation: the transposed-read kernel
  // (PETR_GEMM16_KM=0, diagnostic builds: the general kernel)
  {
    const int am = a16 ? 7 : 3, bm = b16 ? 7 : 3;
    const int other = g.flags & ~(PETR_GEMM_BF16 | PETR_GEMM_A_BF16 | PETR_GEMM_B_BF16 | PETR_GEMM_ATOMIC);
    static const bool km_on = petr_tune("PETR_GEMM16_KM", 1) != 0;
    // (a K-contiguous B has passed the alignment checks above; its N may be ragged)
    if (km_on && !g.a_kcontig && (g.flags & PETR_GEMM_ATOMIC) && !other && !g.bias && !g.r && g.c_nblk <= 0 && !(g.M % GB_BM) &&
        (g.b_kcontig || !(g.N % GB_BN)) && !(g.lda & am) && !(g.ldb & bm) && !(g.a_bs0 & am) && !(g.a_bs1 & am) && !(g.b_bs0 & bm) &&
        !(g.b_bs1 & bm) && !(g.a_seg_stride & am) && !(g.b_seg_stride & bm) && aligned16(g.a) && aligned16(g.b)) {
      DropDev dd;
      memcpy(&dd, &g.drop, sizeof dd);
      if (!dd.thr) {
        dim3 grid(tm * tn * g.nb0 * g.nb1 * g.split_k), block(256);
#define PETR_KM(BKC)                                                                                              \
  do {                                                                                                            \
    if (a16 && b16) hipLaunchKernelGGL((gemm_bf16_km_kernel<true, true, BKC>), grid, block, 0, s, g, tm, tn);     \
    else if (a16) hipLaunchKernelGGL((gemm_bf16_km_kernel<true, false, BKC>), grid, block, 0, s, g, tm, tn);      \
    else if (b16) hipLaunchKernelGGL((gemm_bf16_km_kernel<false, true, BKC>), grid, block, 0, s, g, tm, tn);      \
    else hipLaunchKernelGGL((gemm_bf16_km_kernel<false, false, BKC>), grid, block, 0, s, g, tm, tn);              \
  } while (0)
        if (g.b_kcontig) PETR_KM(true); else PETR_KM(false);
#undef PETR_KM
        PETR_LAUNCH_CHECK("gemm_bf16_km");
        return PETR_OK;
      }
    }
  }
  // the deep-step kernel (K step 128, LDS epilogue) is the default; PETR_GEMM16_DEEP=0 selects the K-step-32 kernel
  static const bool deep_on = petr_tune("PETR_GEMM16_DEEP", 1) != 0;
  // K slices: only short ones (<= 4 deep steps per slice: the split FFN contractions of the 900-row chain); long slices
  // (weight gradients over all tokens) stay with the double-buffered K-step-32 kernel, which measured 2x faster there
  const long deep_steps = (long)(g.k_seg > 0 ? g.K / g.k_seg : 1) * cdiv(kseg, GD_BK);
  if (deep_on && (g.split_k == 1 || cdiv(deep_steps, g.split_k) <= 4)) {
    // vector epilogue: 4 consecutive columns per lane must be contiguous, 16-byte (fp32) / 8-byte (bf16) aligned
    const bool st16 = (g.flags & PETR_GEMM_STORE_BF16) != 0, r16 = (g.flags & PETR_GEMM_R_BF16) != 0;
    const uintptr_t cmask = st16 ? 7 : 15, rmask = r16 ? 7 : 15;
    const bool slabs = g.split_k > 1 && !(g.flags & PETR_GEMM_ATOMIC);
    const int vec_epi =
        !(g.N & 3) && !(g.ldc & 3) && !(g.c_bs0 & 3) && !(g.c_bs1 & 3) && !((uintptr_t)g.c & cmask) &&
        (!slabs || !(g.c_split_stride & 3)) && (g.c_nblk <= 0 || (!(g.c_nblk & 3) && !(g.c_nblk_stride & 3))) &&
        (!g.bias || (aligned16(g.bias) && !(g.bias_bs0 & 3) && !(g.bias_bs1 & 3))) &&
        (!g.r || (!(g.ldr & 3) && !(g.r_bs0 & 3) && !(g.r_bs1 & 3) && !((uintptr_t)g.r & rmask)));
    dim3 grid(tm * tn * g.nb0 * g.nb1 * g.split_k), block(256);
    // transposed-read path for the K-major operand(s): 16-byte rows in multiples of 16, no column sums
    // (PETR_GEMM16_TR=0, diagnostic builds: the scalar-load staging)
    static const bool tr_on = petr_tune("PETR_GEMM16_TR", 1) != 0;
    const int am = a16 ? 7 : 3, bm = b16 ? 7 : 3;
    const bool a_tr_ok = g.a_kcontig || (!(g.lda & am) && !(g.a_bs0 & am) && !(g.a_bs1 & am) && !(g.a_seg_stride & am) && aligned16(g.a) &&
                                         !(g.M & 15));
    const bool b_tr_ok = g.b_kcontig || (!(g.ldb & bm) && !(g.b_bs0 & bm) && !(g.b_bs1 & bm) && !(g.b_seg_stride & bm) && aligned16(g.b) &&
                                         !(g.N & 15));
    const bool tr = tr_on && !(g.a_kcontig && g.b_kcontig) && a_tr_ok && b_tr_ok && !g.a_colsum;
#define PETR_GD(AKC, BKC, A16, B16, TR)                                                                                  \
  do {                                                                                                                   \
    auto kern = gemm_bf16_deep_kernel<AKC, BKC, A16, B16, TR>;                                                           \
    constexpr size_t img = ((TR && !(AKC) ? GD_BK * KM_PITCH : GB_BM * GD_PITCH) + (TR && !(BKC) ? GD_BK * KM_PITCH : GB_BN * GD_PITCH)) * 2; \
    constexpr size_t bytes = img > GD_LDS_BYTES ? img : GD_LDS_BYTES;                                                    \
    static PetrLdsLimit lds_limit;                                                                                       \
    petr_raise_lds_limit(lds_limit, (const void*)kern, (int)bytes);                                                      \
    hipLaunchKernelGGL(kern, grid, block, bytes, s, g, tm, tn, vec_epi);                                                 \
  } while (0)
#define PETR_GD2(AKC, BKC, TR)                              \
  do {                                                      \
    if (a16 && b16) PETR_GD(AKC, BKC, true, true, TR);      \
    else if (a16) PETR_GD(AKC, BKC, true, false, TR);       \
    else if (b16) PETR_GD(AKC, BKC, false, true, TR);       \
    else PETR_GD(AKC, BKC, false, false, TR);               \
  } while (0)
    if (g.a_kcontig && g.b_kcontig) PETR_GD2(true, true, false);
    else if (g.a_kcontig) { if (tr) PETR_GD2(true, false, true); else PETR_GD2(true, false, false); }
    else if (g.b_kcontig) { if (tr) PETR_GD2(false, true, true); else PETR_GD2(false, true, false); }
    else { if (tr) PETR_GD2(false, false, true); else PETR_GD2(false, false, false); }
#undef PETR_GD2
#undef PETR_GD
    PETR_LAUNCH_CHECK("gemm_bf16_deep");
    return PETR_OK;
  }
  {
    DropDev dd;
    memcpy(&dd, &g.drop, sizeof dd);      // petr_gemm() has already derived the device keys
    PETR_CHECK(!dd.thr, PETR_ERR_UNSUPPORTED, "gemm: PETR_GEMM_BF16 dropout epilogue needs the deep-step kernel (short K slices)");
  }
  dim3 grid(tm * tn * g.nb0 * g.nb1 * g.split_k), block(256);
#define PETR_G16(AKC, BKC)                                                                                              \
  do {                                                                                                                  \
    if (a16 && b16) hipLaunchKernelGGL((gemm_bf16_gen_kernel<AKC, BKC, true, true>), grid, block, 0, s, g, tm, tn);     \
    else if (a16) hipLaunchKernelGGL((gemm_bf16_gen_kernel<AKC, BKC, true, false>), grid, block, 0, s, g, tm, tn);      \
    else if (b16) hipLaunchKernelGGL((gemm_bf16_gen_kernel<AKC, BKC, false, true>), grid, block, 0, s, g, tm, tn);      \
    else hipLaunchKernelGGL((gemm_bf16_gen_kernel<AKC, BKC, false, false>), grid, block, 0, s, g, tm, tn);              \
  } while (0)
  if (g.a_kcontig && g.b_kcontig) PETR_G16(true, true);
  else if (g.a_kcontig) PETR_G16(true, false);
  else if (g.b_kcontig) PETR_G16(false, true);
  else PETR_G16(false, false);
#undef PETR_G16
  PETR_LAUNCH_CHECK("gemm_bf16_gen");
  return PETR_OK;
}

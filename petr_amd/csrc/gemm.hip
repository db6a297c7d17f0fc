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
#include <stdlib.h>

#include "common.h"

int petr_gemm_bf16_general(const petr_gemm_args& g, hipStream_t s);   // gemm_bf16.hip

namespace {


struct OperandView {
  const float* p;
  long ld;
  int rows;    // valid rows (M or N)
};

// ---- global -> register stage -------------------------------------------------------------
// Every load is UNCONDITIONAL (addresses clamped into the operand, the value zeroed afterwards by a
// select): a load inside an exec-masked branch makes hipcc wait vmcnt(0) at the end of the branch, which
// serialises the whole prefetch ring into one exposed memory latency per float4 (cdna guide §5 trap (c)).
template <int ROWS, int BK, bool KC, bool VEC, bool A2>
struct Stage {
  static constexpr int NV = ROWS * BK / 4 / 256;
  static constexpr int KPR = BK / 4;        // float4 per row of a K-contiguous tile
  static constexpr int LD = KC ? ROWS + 1 : ROWS + 4;
  float4 v[NV];            // RAW loaded values: nothing may read them before store(), or hipcc waits at the load
  float4 w[A2 ? NV : 1];
  unsigned okm[NV];        // 4 validity bits per float4 (+16: addend active), applied in store()

  // Loop-invariant part of the addressing (per thread, set once per tile row/column block): the PMC profile of
  // the first version showed 7-11 VALU instructions per MFMA, almost all of it per-tile index/clamp/64-bit address
  // arithmetic.  Interior tiles now cost one SGPR-base + 32-bit-VGPR-offset load per float4 and no VALU.
  struct Inv {
    unsigned offb[NV];     // byte offset of this thread's float4 from the tile origin (rows clamped)
    unsigned off2b[NV];    // same for the addend operand (row % a2_rows)
    unsigned rmask[NV];    // row validity bits
  };
  __device__ __forceinline__ static void init(Inv& inv, const OperandView& o, int row0, int a2_rows) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = t + 256 * i;
      if (KC) {
        const int row = row0 + idx / KPR, kl = 4 * (idx % KPR);
        const int rowc = min(row, o.rows - 1);
        inv.offb[i] = (unsigned)(((long)rowc * o.ld + kl) * 4);
        const int r2 = a2_rows > 0 ? rowc % a2_rows : rowc;
        inv.off2b[i] = (unsigned)(((long)r2 * o.ld + kl) * 4);
        inv.rmask[i] = row < o.rows ? 15u : 0u;
      } else {
        constexpr int RPK = ROWS / 4;
        const int kl = idx / RPK, row = row0 + 4 * (idx % RPK);
        inv.offb[i] = (unsigned)(((long)kl * o.ld + min(row, max(o.rows - 4, 0))) * 4);
        inv.off2b[i] = 0;
        inv.rmask[i] = (row < o.rows ? 1u : 0u) | (row + 1 < o.rows ? 2u : 0u) | (row + 2 < o.rows ? 4u : 0u) |
                       (row + 3 < o.rows ? 8u : 0u);
      }
    }
  }
  // interior tile: tile_base = operand + segment + k0 (* ld) is wave-uniform (SGPRs), every k of the tile is valid
  __device__ __forceinline__ void load_fast(const Inv& inv, const float* tile_base, const float* a2_tile_base, bool a2_on) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(tile_base) + inv.offb[i]);
      if (A2) w[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a2_tile_base) + inv.off2b[i]);
      okm[i] = inv.rmask[i] | (a2_on ? 16u : 0u);
    }
  }

  __device__ __forceinline__ static float4 ld4(const float* base, long off0, long off1, long off2, long off3) {
    return make_float4(base[off0], base[off1], base[off2], base[off3]);
  }

  __device__ __forceinline__ void load(const OperandView& o, long seg_off, int row0, int k0, int kend, const float* a2,
                                       int a2_rows, bool a2_on) {
    const int t = threadIdx.x;
    const float* base = o.p + seg_off;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = t + 256 * i;
      float4 r, r2 = make_float4(0.f, 0.f, 0.f, 0.f);
      bool ok0, ok1, ok2, ok3;
      if (KC) {
        const int row = row0 + idx / KPR;
        const int k = k0 + 4 * (idx % KPR);
        const int rowc = min(row, o.rows - 1);
        const bool rok = row < o.rows;
        ok0 = rok && k < kend; ok1 = rok && k + 1 < kend; ok2 = rok && k + 2 < kend; ok3 = rok && k + 3 < kend;
        if (VEC) {
          const int kc = min(k, kend - 4);
          r = *reinterpret_cast<const float4*>(base + (long)rowc * o.ld + kc);
          ok1 = ok2 = ok3 = ok0;       // K % 4 == 0: a float4 is entirely inside or outside
        } else {
          const long ro = (long)rowc * o.ld;
          r = ld4(base, ro + min(k, kend - 1), ro + min(k + 1, kend - 1), ro + min(k + 2, kend - 1), ro + min(k + 3, kend - 1));
        }
        if (A2) {
          const int r2row = a2_rows > 0 ? rowc % a2_rows : rowc;
          const float* b2 = a2 + seg_off;
          if (VEC) {
            r2 = *reinterpret_cast<const float4*>(b2 + (long)r2row * o.ld + min(k, kend - 4));
          } else {
            const long ro = (long)r2row * o.ld;
            r2 = ld4(b2, ro + min(k, kend - 1), ro + min(k + 1, kend - 1), ro + min(k + 2, kend - 1), ro + min(k + 3, kend - 1));
          }
        }
      } else {
        constexpr int RPK = ROWS / 4;
        const int k = k0 + idx / RPK;
        const int row = row0 + 4 * (idx % RPK);
        const int kc = min(k, kend - 1);
        const bool kok = k < kend;
        ok0 = kok && row < o.rows; ok1 = kok && row + 1 < o.rows; ok2 = kok && row + 2 < o.rows; ok3 = kok && row + 3 < o.rows;
        if (VEC) {
          r = *reinterpret_cast<const float4*>(base + (long)kc * o.ld + min(row, o.rows - 4));
          ok1 = ok2 = ok3 = ok0;       // rows % 4 == 0
        } else {
          const long ko = (long)kc * o.ld;
          r = ld4(base, ko + min(row, o.rows - 1), ko + min(row + 1, o.rows - 1), ko + min(row + 2, o.rows - 1),
                  ko + min(row + 3, o.rows - 1));
        }
      }
      v[i] = r;
      if (A2) w[i] = r2;
      okm[i] = (ok0 ? 1u : 0u) | (ok1 ? 2u : 0u) | (ok2 ? 4u : 0u) | (ok3 ? 8u : 0u) | (a2_on ? 16u : 0u);
    }
  }

  __device__ __forceinline__ void store(float* lds) const {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = t + 256 * i;
      float4 x = v[i];
      const unsigned m = okm[i];
      if (A2 && (m & 16u)) { x.x += w[i].x; x.y += w[i].y; x.z += w[i].z; x.w += w[i].w; }
      x.x = (m & 1u) ? x.x : 0.f; x.y = (m & 2u) ? x.y : 0.f; x.z = (m & 4u) ? x.z : 0.f; x.w = (m & 8u) ? x.w : 0.f;
      if (KC) {
        const int row = idx / KPR, kq = idx % KPR;
        float* d = lds + (4 * kq) * LD + row;
        d[0] = x.x;
        d[LD] = x.y;
        d[2 * LD] = x.z;
        d[3 * LD] = x.w;
      } else {
        constexpr int RPK = ROWS / 4;
        const int k = idx / RPK, r4 = idx % RPK;
        *reinterpret_cast<float4*>(lds + k * LD + 4 * r4) = x;
      }
    }
  }
};

template <int BM, int BN, int WM, int WN, int BK, bool AKC, bool BKC, bool VEC, bool A2, int PD = 2>
__global__ __launch_bounds__(256) void gemm_kernel(const petr_gemm_args g, const int tiles_m, const int tiles_n) {
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
  constexpr int TM = WM / 32, TN = WN / 32;
  using SA = Stage<BM, BK, AKC, VEC, A2>;
  using SB = Stage<BN, BK, BKC, VEC, false>;
  // two LDS images of (A tile, B tile): tile t+1 is written while the MFMAs of tile t read the other image,
  // ONE barrier per k-tile, and the ds_writes/global loads sit in the same basic block as the MFMAs so the
  // compiler interleaves them (a single wave per SIMD otherwise serialises store -> barrier -> read -> MFMA)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int BUF = BK * SA::LD + BK * SB::LD;

  // ---- which tile / batch / K-slice ----
  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm_i = tile / tiles_n, tn_i = tile - tm_i * tiles_n;
  const int m0 = tm_i * BM, n0 = tn_i * BN;
  int z = blockIdx.z;
  const int ks = z % g.split_k;
  z /= g.split_k;
  const int z1 = z % g.nb1, z0 = z / g.nb1;

  OperandView A{g.a + z0 * g.a_bs0 + z1 * g.a_bs1, g.lda, g.M};
  OperandView B{g.b + z0 * g.b_bs0 + z1 * g.b_bs1, g.ldb, g.N};
  // the addend applies to the first a2_ncols output columns only (block-uniform)
  const bool use_a2 = A2 && (g.a2_ncols <= 0 || n0 < g.a2_ncols);
  const float* a2 = g.a2;

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

  // Register prefetch ring, PD tiles deep: these contractions are short chains of k-tiles whose
  // per-tile MFMA time (16 MFMAs) is far below one global-load latency, so ONE tile of look-ahead
  // leaves every iteration waiting ~1-2 us on its loads; PD tiles in flight hide it.  Two, not three: the third
  // stage costs ~25 VGPRs, i.e. a wave of occupancy, and lost 4 % of the whole step in a same-box A/B (c5 5.32 ->
  // 5.10 ms, p4-1600 12.0 -> 11.45 ms); PD = 1 ties with 2.
  // PD = 4 (launch_cfg picks it for grids of at most one workgroup per CU: the 900-row contractions of the decoder): there
  // occupancy is irrelevant, the workgroup is alone on its CU and its 8 serial k-tiles each wait for an exposed share of the
  // memory latency, so four tiles in flight instead of two halve the chain.
  SA sa[PD];
  SB sb[PD];
  // Tiles past kt_end are loaded from clamped addresses with an all-zero validity mask and multiplied as zeros:
  // the loop body stays branch-free straight-line code ([store t+1][load t+1+PD][MFMA t][barrier] x PD), which is
  // what lets hipcc keep COUNTED vmcnt waits (two stages in flight) instead of draining the ring at every store.
  typename SA::Inv ia;
  typename SB::Inv ib;
  SA::init(ia, A, m0, g.a2_rows);
  SB::init(ib, B, n0, 0);
  auto stage_load = [&](SA& ra, SB& rb, int kt) {
    const bool live = kt < kt_end;
    const int ktc = live ? kt : kt_end - 1;
    const int seg = ktc / tps;
    const int k0 = live ? (ktc - seg * tps) * BK : kseg;      // k0 >= kseg: every element masked out
    if (VEC && k0 + BK <= kseg) {     // wave-uniform: interior tile
      const float* ta = A.p + (long)seg * g.a_seg_stride + (AKC ? (long)k0 : (long)k0 * A.ld);
      const float* tb = B.p + (long)seg * g.b_seg_stride + (BKC ? (long)k0 : (long)k0 * B.ld);
      const float* t2 = A2 ? a2 + (long)seg * g.a_seg_stride + k0 : nullptr;
      ra.load_fast(ia, ta, t2, use_a2);
      rb.load_fast(ib, tb, nullptr, false);
    } else {
      ra.load(A, (long)seg * g.a_seg_stride, m0, k0, kseg, a2, g.a2_rows, use_a2);
      rb.load(B, (long)seg * g.b_seg_stride, n0, k0, kseg, nullptr, 0, false);
    }
  };
  const int nkt = kt_end - kt_begin;
  const int kt_stop = kt_begin + ((nkt + PD - 1) / PD) * PD;   // padded trip count
  if (nkt > 0) {
#pragma unroll
    for (int i = 0; i < PD; ++i) stage_load(sa[i], sb[i], kt_begin + i);
    sa[0].store(lds);
    sb[0].store(lds + BK * SA::LD);
    stage_load(sa[0], sb[0], kt_begin + PD);
    __syncthreads();
    for (int kt0 = kt_begin; kt0 < kt_stop; kt0 += PD) {
#pragma unroll
      for (int j = 0; j < PD; ++j) {
        const int kt = kt0 + j;
        const int cur = (kt - kt_begin) & 1;
        const float* As = lds + cur * BUF;
        const float* Bs = As + BK * SA::LD;
        const int jn = (j + 1) % PD;   // compile-time after unrolling
        float* An = lds + (cur ^ 1) * BUF;
        sa[jn].store(An);
        sb[jn].store(An + BK * SA::LD);
        stage_load(sa[jn], sb[jn], kt + 1 + PD);
        if (do_colsum && threadIdx.x < BM) {
          float cs = 0.f;
#pragma unroll
          for (int k = 0; k < BK; ++k) cs += As[k * SA::LD + threadIdx.x];
          colacc += cs;
        }
        // Fragment reads run AHEAD of the MFMAs that consume them (left alone, hipcc emits read -> lgkmcnt wait ->
        // MFMA per step and the ~100-cycle LDS latency sits in front of every 64-cycle MFMA of the dependent chain):
        // LA steps of fragments are requested before the chain starts and each step refills the slot it frees.
        constexpr int STEPS = BK / 2;
        constexpr int LA = (TM * TN == 1) ? 4 : (TM * TN == 2 ? 4 : 2);   // look-ahead in k-steps (registers: LA*(TM+TN))
        float af[LA][TM], bf[LA][TN];
#pragma unroll
        for (int s = 0; s < LA; ++s) {
#pragma unroll
          for (int i = 0; i < TM; ++i) af[s][i] = As[(2 * s + h) * SA::LD + wm0 + i * 32 + c];
#pragma unroll
          for (int jj = 0; jj < TN; ++jj) bf[s][jj] = Bs[(2 * s + h) * SB::LD + wn0 + jj * 32 + c];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jj = 0; jj < TN; ++jj)
              acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s % LA][i], bf[s % LA][jj], acc[i][jj], 0, 0, 0);
          if (s + LA < STEPS) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[s % LA][i] = As[(2 * (s + LA) + h) * SA::LD + wm0 + i * 32 + c];
#pragma unroll
            for (int jj = 0; jj < TN; ++jj) bf[s % LA][jj] = Bs[(2 * (s + LA) + h) * SB::LD + wn0 + jj * 32 + c];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
      }
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
  DropDev dd;
  __builtin_memcpy(&dd, &g.drop, sizeof dd);   // petr_gemm() stored the derived keys here
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn0 + j * 32 + c;
    const int nc = min(n, g.N - 1);                 // clamped for loads; stores are predicated
    const float bv = (bias && !(flags & PETR_GEMM_BIAS_M)) ? bias[nc] : 0.f;   // BIAS_M: the vector has M entries, not N
    const long ccol = g.c_nblk > 0 ? (long)(nc / g.c_nblk) * g.c_nblk_stride + (nc % g.c_nblk) : (long)nc;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + i * 32 + mfma32_row(r, h);
        const int mc = min(m, g.M - 1);
        const bool ok = m < g.M && n < g.N;
        float v = acc[i][j][r] * g.alpha + ((flags & PETR_GEMM_BIAS_M) ? (bias ? bias[mc] : 0.f) : bv);
        float* dst = C + (long)mc * g.ldc + ccol;
        const float rv = R ? R[(long)mc * g.ldr + nc] : 0.f;      // R is kernel-uniform: no exec-masked load
        const float old = (flags & PETR_GEMM_ACCUMULATE) ? *dst : 0.f;
        if (flags & PETR_GEMM_SIGMOID_MUL) v = rv * (1.f / (1.f + expf(-v)));
        else if (flags & PETR_GEMM_RELU_MASK) v = rv > 0.f ? v : 0.f;
        else v += rv;
        if (flags & PETR_GEMM_RELU) v = fmaxf(v, 0.f);
        if (dd.thr) v = drop_keep(drop_row_key(dd, (uint32_t)mc), (uint32_t)nc, dd.thr) ? v * dd.scale : 0.f;
        v += old;
        if (ok) {
          if (flags & PETR_GEMM_STORE_BF16)     // bf16 output: same index arithmetic on 2-byte elements
            reinterpret_cast<uint16_t*>(g.c)[z0 * g.c_bs0 + z1 * g.c_bs1 + (long)mc * g.ldc + ccol] =
                __builtin_bit_cast(uint16_t, (__bf16)v);
          else if (atomic) atomicAdd(dst, v);
          else *dst = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// "Skinny" contraction for the query side of the decoder and for weight gradients: outputs with few
// 64x64 tiles (900 x 256, 256 x 256, ...) where the tiled kernel above leaves 3/4 of the CUs idle and
// spends its time in a serial chain of k-tiles (load -> LDS -> barrier -> 16 MFMAs).
//   workgroup = one 32x32 output tile, its NW waves split K between them (K-split INSIDE the
//   workgroup: the partial tiles are summed through LDS, so there are no float atomics and no slab
//   pass even for weight gradients); operands go global -> registers directly in MFMA fragment order
//   (lane (c,h) holds row c, k = 16h..16h+15 of a 32-deep chunk), no LDS staging, no barrier in the
//   main loop, the next chunk is in flight while the 16 MFMAs of the current one run.
// 900x256x256 becomes 232 workgroups x 8 waves with ONE chunk (16 MFMAs) per wave instead of
// 60 workgroups x 8 serial k-tiles.
// ---------------------------------------------------------------------------------------------
template <bool KC, bool VEC>
__device__ __forceinline__ void skinny_load(const float* base, long ld, int row, int rows, int k0, int kend, float f[16]) {
  // fragment order: element s of lane-half h is k = k0 + 16h + s (h folded into k0 by the caller)
  const int rowc = min(row, rows - 1);
  const bool rok = row < rows;
  if (KC) {
    const float* p = base + (long)rowc * ld;
    if (VEC) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + 4 * j;
        const float4 v = *reinterpret_cast<const float4*>(p + min(k, kend - 4));
        const bool ok = rok && k < kend;
        f[4 * j] = ok ? v.x : 0.f; f[4 * j + 1] = ok ? v.y : 0.f; f[4 * j + 2] = ok ? v.z : 0.f; f[4 * j + 3] = ok ? v.w : 0.f;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float v = p[min(k0 + s, kend - 1)];
        f[s] = (rok && k0 + s < kend) ? v : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float v = base[(long)min(k0 + s, kend - 1) * ld + rowc];
      f[s] = (rok && k0 + s < kend) ? v : 0.f;
    }
  }
}

template <bool AKC, bool BKC, bool VEC, bool A2>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(const petr_gemm_args g, const int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [NW][16][64] partial tiles (+ [NW][64] column sums)
  const int NW = blockDim.x >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int h = lane >> 5, c = lane & 31;
  const int tiles = gridDim.x;
  const int tile = xcd_remap(blockIdx.x, tiles);
  const int tm_i = tile / tiles_n, tn_i = tile - tm_i * tiles_n;
  const int m0 = tm_i * 32, n0 = tn_i * 32;
  const int z1 = blockIdx.z % g.nb1, z0 = blockIdx.z / g.nb1;
  const float* Ab = g.a + z0 * g.a_bs0 + z1 * g.a_bs1;
  const float* Bb = g.b + z0 * g.b_bs0 + z1 * g.b_bs1;
  const bool use_a2 = A2 && (g.a2_ncols <= 0 || n0 < g.a2_ncols);

  const int kseg = g.k_seg > 0 ? g.k_seg : g.K;
  const int nseg = g.k_seg > 0 ? g.K / g.k_seg : 1;
  const int cps = (kseg + 31) >> 5;                    // 32-deep chunks per segment
  const int chunks = nseg * cps;
  const int per = (chunks + NW - 1) / NW;
  const int ch_begin = wave * per, ch_end = min(chunks, ch_begin + per);

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float colacc = 0.f;
  const bool do_colsum = g.a_colsum != nullptr && tn_i == 0;

  float fa[2][16], fb[2][16], fa2[A2 ? 16 : 1];
  auto load_chunk = [&](int ch, float (&da)[16], float (&db)[16]) {
    const int seg = ch / cps, k0 = (ch - seg * cps) * 32 + 16 * h;
    skinny_load<AKC, VEC>(Ab + (long)seg * g.a_seg_stride, g.lda, m0 + c, g.M, k0, kseg, da);
    skinny_load<BKC, VEC>(Bb + (long)seg * g.b_seg_stride, g.ldb, n0 + c, g.N, k0, kseg, db);
    if (A2) {
      const int row = m0 + c;
      const int r2 = g.a2_rows > 0 ? min(row, g.M - 1) % g.a2_rows : row;
      skinny_load<true, VEC>(g.a2 + (long)seg * g.a_seg_stride, g.lda, r2, g.a2_rows > 0 ? g.a2_rows : g.M, k0, kseg, fa2);
      const bool on = use_a2 && row < g.M;
#pragma unroll
      for (int s = 0; s < 16; ++s) da[s] += on ? fa2[s] : 0.f;
    }
  };
  if (ch_begin < ch_end) load_chunk(ch_begin, fa[0], fb[0]);
  for (int ch = ch_begin; ch < ch_end; ch += 2) {
    if (ch + 1 < ch_end) load_chunk(ch + 1, fa[1], fb[1]);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][s], fb[0][s], acc, 0, 0, 0);
      colacc += fa[0][s];
    }
    if (ch + 1 < ch_end) {
      if (ch + 2 < ch_end) load_chunk(ch + 2, fa[0], fb[0]);
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][s], fb[1][s], acc, 0, 0, 0);
        colacc += fa[1][s];
      }
    }
  }

  // ---- sum the NW partial tiles through LDS (deterministic order) ----
  float* csum = red + NW * 1024;
  if (NW > 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    if (do_colsum) csum[wave * 64 + lane] = colacc;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r];
      for (int w = 1; w < NW; ++w) v += red[(w * 16 + r) * 64 + lane];
      acc[r] = v;
    }
    if (do_colsum)
      for (int w = 1; w < NW; ++w) colacc += csum[w * 64 + lane];
  }
  if (do_colsum) {
    const float tot = xhalf_sum(colacc);              // the two lane halves hold the two k halves of row c
    if (h == 0 && m0 + c < g.M) {
      float* dst = g.a_colsum + z0 * g.cs_bs0 + z1 * g.cs_bs1 + m0 + c;
      *dst += tot;                                     // single owner per row: plain read-modify-write
    }
  }

  // ---- epilogue (same semantics as the tiled kernel; PETR_GEMM_ATOMIC degrades to a plain += here) ----
  float* C = g.c + z0 * g.c_bs0 + z1 * g.c_bs1;
  const float* bias = g.bias ? g.bias + z0 * g.bias_bs0 + z1 * g.bias_bs1 : nullptr;
  const float* R = g.r ? g.r + z0 * g.r_bs0 + z1 * g.r_bs1 : nullptr;
  const int flags = g.flags;
  const bool accumulate = (flags & (PETR_GEMM_ACCUMULATE | PETR_GEMM_ATOMIC)) != 0;
  DropDev dd;
  __builtin_memcpy(&dd, &g.drop, sizeof dd);
  const int n = n0 + c;
  const int nc = min(n, g.N - 1);
  const float bv = bias ? bias[nc] : 0.f;
  const long ccol = g.c_nblk > 0 ? (long)(nc / g.c_nblk) * g.c_nblk_stride + (nc % g.c_nblk) : (long)nc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + mfma32_row(r, h);
    const int mc = min(m, g.M - 1);
    float v = acc[r] * g.alpha + bv;
    float* dst = C + (long)mc * g.ldc + ccol;
    const float rv = R ? R[(long)mc * g.ldr + nc] : 0.f;
    const float old = accumulate ? *dst : 0.f;
    if (flags & PETR_GEMM_SIGMOID_MUL) v = rv * (1.f / (1.f + expf(-v)));
    else if (flags & PETR_GEMM_RELU_MASK) v = rv > 0.f ? v : 0.f;
    else v += rv;
    if (flags & PETR_GEMM_RELU) v = fmaxf(v, 0.f);
    if (dd.thr) v = drop_keep(drop_row_key(dd, (uint32_t)mc), (uint32_t)nc, dd.thr) ? v * dd.scale : 0.f;
    v += old;
    if (m < g.M && n < g.N) *dst = v;
  }
}

template <bool VEC>
int launch_skinny(const petr_gemm_args& g, hipStream_t s) {
  const int tiles_m = (int)cdiv(g.M, 32), tiles_n = (int)cdiv(g.N, 32);
  const long tiles = (long)tiles_m * tiles_n * g.nb0 * g.nb1;
  const int kseg = g.k_seg > 0 ? g.k_seg : g.K;
  const long chunks = (long)(g.k_seg > 0 ? g.K / g.k_seg : 1) * cdiv(kseg, 32);
  // enough waves to cover the 1024 SIMDs ~1.5x, but never less than one chunk per wave
  int nw = 1;
  while (nw < 8 && tiles * nw < 1536 && chunks >= 2L * nw) nw *= 2;
  while (nw < 8 && chunks > 8L * nw) nw *= 2;           // keep each wave's MFMA chain short
  dim3 grid(tiles_m * tiles_n, 1, g.nb0 * g.nb1), block(64 * nw);
  const size_t lds = (size_t)nw * (1024 + 64) * sizeof(float);
#define PETR_SKINNY_LAUNCH(AKC, BKC, A2) \
  hipLaunchKernelGGL((gemm_skinny_kernel<AKC, BKC, VEC, A2>), grid, block, lds, s, g, tiles_n)
  if (g.a_kcontig && g.b_kcontig) {
    if (g.a2) PETR_SKINNY_LAUNCH(true, true, true);
    else PETR_SKINNY_LAUNCH(true, true, false);
  } else if (g.a_kcontig) {
    if (g.a2) PETR_SKINNY_LAUNCH(true, false, true);
    else PETR_SKINNY_LAUNCH(true, false, false);
  } else if (g.b_kcontig) {
    PETR_SKINNY_LAUNCH(false, true, false);
  } else {
    PETR_SKINNY_LAUNCH(false, false, false);
  }
#undef PETR_SKINNY_LAUNCH
  PETR_LAUNCH_CHECK("gemm_skinny");
  return PETR_OK;
}

template <int BM, int BN, int WM, int WN, int BK, bool VEC>
int launch_cfg(const petr_gemm_args& g, hipStream_t s) {
  const int tiles_m = (int)cdiv(g.M, BM), tiles_n = (int)cdiv(g.N, BN);
  dim3 grid(tiles_m * tiles_n, 1, g.nb0 * g.nb1 * g.split_k);
  dim3 block(256);
  // small grids with a chain of k-tiles: a deep-prefetch instantiation (four tiles in flight), opt-in with PETR_GEMM_DEEP=1.
  // Measured and rejected as the default (same-box A/B, scripts/ab_env.sh, two rounds): c5 fp32 5.12 -> 5.20 ms/step,
  // p4-1600 fp32 11.18 -> 11.24, bf16 6.43 -> 6.51: the 900-row contractions are not waiting for memory (one 32 x 32
  // accumulator per wave = one dependent MFMA chain of 8 x 16 x 64 cycles), the extra registers only cost.
  static const bool deep_on = petr_tune("PETR_GEMM_DEEP", 0) != 0;
  const int kseg_ = g.k_seg > 0 ? g.k_seg : g.K;
  const long ktiles_ = (long)(g.k_seg > 0 ? g.K / g.k_seg : 1) * cdiv(kseg_, BK);
  const bool deep = deep_on && VEC && (long)grid.x * grid.z <= 256 && ktiles_ / g.split_k >= 4;
  // dynamic LDS: two images; beyond 64 KB the per-kernel limit is raised once (host-side attribute, not a launch)
#define PETR_GEMM_LAUNCH(AKC, BKC, A2)                                                                          \
  do {                                                                                                          \
    constexpr size_t lds_bytes =                                                                                \
        2 * (size_t)BK * (Stage<BM, BK, AKC, VEC, A2>::LD + Stage<BN, BK, BKC, VEC, false>::LD) * 4;              \
    auto kern = gemm_kernel<BM, BN, WM, WN, BK, AKC, BKC, VEC, A2>;                                                 \
    static PetrLdsLimit lds_limit;                                                                              \
    if (lds_bytes > 65536) petr_raise_lds_limit(lds_limit, (const void*)kern, (int)lds_bytes);                  \
    if (deep) hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, BK, AKC, BKC, VEC, A2, 4>), grid, block, lds_bytes, s, g,   \
                                 tiles_m, tiles_n);                                                              \
    else hipLaunchKernelGGL(kern, grid, block, lds_bytes, s, g, tiles_m, tiles_n);                              \
  } while (0)
  if (g.a_kcontig && g.b_kcontig) {
    if (g.a2) PETR_GEMM_LAUNCH(true, true, true);
    else PETR_GEMM_LAUNCH(true, true, false);
  } else if (g.a_kcontig) {
    if (g.a2) PETR_GEMM_LAUNCH(true, false, true);
    else PETR_GEMM_LAUNCH(true, false, false);
  } else if (g.b_kcontig) {
    PETR_GEMM_LAUNCH(false, true, false);
  } else {
    PETR_GEMM_LAUNCH(false, false, false);
  }
#undef PETR_GEMM_LAUNCH
  PETR_LAUNCH_CHECK("gemm");
  return PETR_OK;
}

bool operand_vec_ok(const float* p, long ld, long bs0, long bs1, long seg, int kcontig, int rows, int K) {
  if (!aligned16(p) || (ld & 3) || (bs0 & 3) || (bs1 & 3) || (seg & 3)) return false;
  // contiguous extent must be a multiple of 4 (>= 4) so that a float4 never straddles the valid edge
  return kcontig ? (K % 4 == 0) : (rows % 4 == 0);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// bf16 contraction (PETR_GEMM_BF16): fp32 operands in memory, rounded to bf16 on load, v_mfma_f32_32x32x16_bf16, fp32
// accumulation, epilogue as above (bias, residual, ReLU, fp32 or bf16 store).  For the L-sized projections of the
// bf16 inference path (BASELINE configs 3-5).  Both operands K-contiguous, K a multiple of 16.
//   wave = 32 rows x 64 columns (two accumulators share the A fragment), workgroup = 4 waves = 64 x 128;
//   fragments go global -> registers directly: lane (c, h) holds row c, k = 16 j + 8 h .. + 7 of chunk j as two
//   float4 loads (the lane pair h = 0, 1 covers 64 contiguous bytes of the row), converted with four
//   v_cvt_pk_bf16_f32; the k-slot order inside an MFMA only has to agree between the operands.  No LDS, no barrier:
//   the next chunk's loads are issued before the current chunk's products.
// ---------------------------------------------------------------------------------------------
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));

template <bool A2>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const petr_gemm_args g, const int tiles_n) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int h = lane >> 5, c = lane & 31;
  const int tiles = gridDim.x;
  const int tile = xcd_remap(blockIdx.x, tiles);
  const int tm_i = tile / tiles_n, tn_i = tile - tm_i * tiles_n;
  const int m0 = tm_i * 64 + (wave >> 1) * 32, n0 = tn_i * 128 + (wave & 1) * 64;
  const int z1 = blockIdx.z % g.nb1, z0 = blockIdx.z / g.nb1;
  if (m0 >= g.M || n0 >= g.N) return;                      // wave-uniform
  const float* Ab = g.a + z0 * g.a_bs0 + z1 * g.a_bs1 + (long)min(m0 + c, g.M - 1) * g.lda + 8 * h;
  const float* Bb0 = g.b + z0 * g.b_bs0 + z1 * g.b_bs1 + (long)min(n0 + c, g.N - 1) * g.ldb + 8 * h;
  const float* Bb1 = g.b + z0 * g.b_bs0 + z1 * g.b_bs1 + (long)min(n0 + 32 + c, g.N - 1) * g.ldb + 8 * h;
  const bool use_a2 = A2 && (g.a2_ncols <= 0 || n0 < g.a2_ncols);
  const float* A2b = nullptr;
  if (A2) {
    const int row = min(m0 + c, g.M - 1);
    A2b = g.a2 + (long)(g.a2_rows > 0 ? row % g.a2_rows : row) * g.lda + 8 * h;
  }
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  struct Frag { float4 a[2], b0[2], b1[2], a2[2]; };
  auto load = [&](int k, Frag& f) {
    f.a[0] = *reinterpret_cast<const float4*>(Ab + k); f.a[1] = *reinterpret_cast<const float4*>(Ab + k + 4);
    f.b0[0] = *reinterpret_cast<const float4*>(Bb0 + k); f.b0[1] = *reinterpret_cast<const float4*>(Bb0 + k + 4);
    f.b1[0] = *reinterpret_cast<const float4*>(Bb1 + k); f.b1[1] = *reinterpret_cast<const float4*>(Bb1 + k + 4);
    if (A2) { f.a2[0] = *reinterpret_cast<const float4*>(A2b + k); f.a2[1] = *reinterpret_cast<const float4*>(A2b + k + 4); }
  };
  auto cvt = [](const float4& u, const float4& v) {
    gbf16x8 o = {(__bf16)u.x, (__bf16)u.y, (__bf16)u.z, (__bf16)u.w, (__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    return o;
  };
  auto mac = [&](Frag& f) {
    if (A2 && use_a2) {
      f.a[0].x += f.a2[0].x; f.a[0].y += f.a2[0].y; f.a[0].z += f.a2[0].z; f.a[0].w += f.a2[0].w;
      f.a[1].x += f.a2[1].x; f.a[1].y += f.a2[1].y; f.a[1].z += f.a2[1].z; f.a[1].w += f.a2[1].w;
    }
    const gbf16x8 fa = cvt(f.a[0], f.a[1]);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, cvt(f.b0[0], f.b0[1]), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, cvt(f.b1[0], f.b1[1]), acc1, 0, 0, 0);
  };
  Frag f0, f1;
  load(0, f0);
  for (int k = 0; k < g.K; k += 32) {
    if (k + 16 < g.K) load(k + 16, f1);
    mac(f0);
    if (k + 16 < g.K) {
      if (k + 32 < g.K) load(k + 32, f0);
      mac(f1);
    }
  }
  // ---- epilogue: D[m][n]: lane holds column n = c, rows mfma32_row(r, h) ----
  const float* bias = g.bias ? g.bias + z0 * g.bias_bs0 + z1 * g.bias_bs1 : nullptr;
  const float* R = g.r ? g.r + z0 * g.r_bs0 + z1 * g.r_bs1 : nullptr;
  const long cbase = z0 * g.c_bs0 + z1 * g.c_bs1;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + 32 * j + c;
    const int nc = min(n, g.N - 1);
    const float bv = bias ? bias[nc] : 0.f;
    const long ccol = g.c_nblk > 0 ? (long)(nc / g.c_nblk) * g.c_nblk_stride + (nc % g.c_nblk) : (long)nc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mfma32_row(r, h);
      const int mc = min(m, g.M - 1);
      float v = (j == 0 ? acc0[r] : acc1[r]) * g.alpha + bv;
      if (R) v += R[(long)mc * g.ldr + nc];
      if (g.flags & PETR_GEMM_RELU) v = fmaxf(v, 0.f);
      if (m < g.M && n < g.N) {
        const long off = cbase + (long)mc * g.ldc + ccol;
        if (g.flags & PETR_GEMM_STORE_BF16) reinterpret_cast<uint16_t*>(g.c)[off] = __builtin_bit_cast(uint16_t, (__bf16)v);
        else g.c[off] = v;
      }
    }
  }
}

// LDS-staged version of the bf16 contraction (K % 32 == 0): workgroup = 128 x 128 outputs, 4 waves of 64 x 64 (2 x 2 MFMA
// tiles), K step 32.  Operand tiles are read with coalesced float4 loads (8 consecutive threads = one 128-byte row
// piece), rounded to bf16 and written to LDS rows of 32 + 8 elements (80-byte pitch: the 16-byte fragment reads of 16
// lanes fall into 64 distinct banks); register prefetch of the next K step, double-buffered LDS, one barrier per step.
// AKM: A is K-major (A(m,k) = a[k*lda + m], the NCHW feature / coordinate maps): a thread stages one row m and 16
// consecutive k with dword loads that are coalesced ACROSS the lanes (consecutive m), so the LDS image needs no transposition.
template <bool A2, bool AKM>
__global__ __launch_bounds__(256, 2) void gemm_bf16_lds_kernel(const petr_gemm_args g, const int tiles_n) {
  constexpr int BM = 128, BN = 128, BK = 32, PITCH = 40;
  __shared__ __attribute__((aligned(16))) uint16_t lds[2][(BM + BN) * PITCH];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int h = lane >> 5, c = lane & 31;
  const int tiles = gridDim.x;
  const int tile = xcd_remap(blockIdx.x, tiles);
  const int tm_i = tile / tiles_n, tn_i = tile - tm_i * tiles_n;
  const int m0 = tm_i * BM, n0 = tn_i * BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int z1 = blockIdx.z % g.nb1, z0 = blockIdx.z / g.nb1;
  const float* Ab = g.a + z0 * g.a_bs0 + z1 * g.a_bs1;
  const float* Bb = g.b + z0 * g.b_bs0 + z1 * g.b_bs1;
  const bool use_a2 = A2 && (g.a2_ncols <= 0 || n0 < g.a2_ncols);

  // staging assignment: 4 float4 per thread and operand; float4 index = t + 256 i -> row = idx >> 3, piece = idx & 7
  const int s_row = t >> 3, s_c4 = t & 7;                   // rows s_row + 32 i
  int a_off[4], b_off[4], a2_off[4];      // element offsets inside one batch (checked < 2^31 on the host)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = min(m0 + s_row + 32 * i, g.M - 1), rb = min(n0 + s_row + 32 * i, g.N - 1);
    a_off[i] = ra * (int)g.lda + 4 * s_c4;
    b_off[i] = rb * (int)g.ldb + 4 * s_c4;
    a2_off[i] = A2 ? (g.a2_rows > 0 ? ra % g.a2_rows : ra) * (int)g.lda + 4 * s_c4 : 0;
  }
  float4 ra_[4], rb_[4], ra2_[4];
  const int km_row = min(m0 + (t & 127), g.M - 1), km_k = (t >> 7) * 16;      // AKM: row and first k of this thread
  auto gload = [&](int k0) {
    if (AKM) {
      const float* src = Ab + (long)(k0 + km_k) * g.lda + km_row;
      float* dst = reinterpret_cast<float*>(ra_);
#pragma unroll
      for (int e = 0; e < 16; ++e) dst[e] = src[(long)e * g.lda];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!AKM) ra_[i] = *reinterpret_cast<const float4*>(Ab + a_off[i] + k0);
      rb_[i] = *reinterpret_cast<const float4*>(Bb + b_off[i] + k0);
      if (A2) ra2_[i] = *reinterpret_cast<const float4*>(g.a2 + a2_off[i] + k0);
    }
  };
  auto pack = [](const float4& v) {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    b4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    return __builtin_bit_cast(uint2, o);
  };
  auto lstore = [&](int buf) {
    uint16_t* As = lds[buf];
    uint16_t* Bs = As + BM * PITCH;
    if (AKM) {
      const uint2 p0 = pack(ra_[0]), p1 = pack(ra_[1]), p2 = pack(ra_[2]), p3 = pack(ra_[3]);
      uint4* d = reinterpret_cast<uint4*>(As + (t & 127) * PITCH + km_k);
      d[0] = make_uint4(p0.x, p0.y, p1.x, p1.y);
      d[1] = make_uint4(p2.x, p2.y, p3.x, p3.y);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!AKM) {
        float4 av = ra_[i];
        if (A2 && use_a2) { av.x += ra2_[i].x; av.y += ra2_[i].y; av.z += ra2_[i].z; av.w += ra2_[i].w; }
        *reinterpret_cast<uint2*>(As + (s_row + 32 * i) * PITCH + 4 * s_c4) = pack(av);
      }
      *reinterpret_cast<uint2*>(Bs + (s_row + 32 * i) * PITCH + 4 * s_c4) = pack(rb_[i]);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / BK;
  gload(0);
  lstore(0);
  if (nk > 1) gload(BK);
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    __syncthreads();                       // image `buf` complete; the other one is free again
    if (ks + 1 < nk) {
      lstore(buf ^ 1);
      if (ks + 2 < nk) gload((ks + 2) * BK);
    }
    const uint16_t* As = lds[buf];
    const uint16_t* Bs = As + BM * PITCH;
#pragma unroll
    for (int j = 0; j < 2; ++j) {          // two 16-deep chunks of the K step
      uint4 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const uint4*>(As + (wm + 32 * i + c) * PITCH + 16 * j + 8 * h);
        fb[i] = *reinterpret_cast<const uint4*>(Bs + (wn + 32 * i + c) * PITCH + 16 * j + 8 * h);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
          acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gbf16x8, fa[i]), __builtin_bit_cast(gbf16x8, fb[jn]),
                                                               acc[i][jn], 0, 0, 0);
    }
  }

  const float* bias = g.bias ? g.bias + z0 * g.bias_bs0 + z1 * g.bias_bs1 : nullptr;
  const float* R = g.r ? g.r + z0 * g.r_bs0 + z1 * g.r_bs1 : nullptr;
  const long cbase = z0 * g.c_bs0 + z1 * g.c_bs1;
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
        float v = acc[i][jn][r] * g.alpha + bv;
        if (R) v += R[(long)mc * g.ldr + nc];
        if (g.flags & PETR_GEMM_RELU) v = fmaxf(v, 0.f);
        if (m < g.M && n < g.N) {
          const long off = cbase + (long)mc * g.ldc + ccol;
          if (g.flags & PETR_GEMM_STORE_BF16) reinterpret_cast<uint16_t*>(g.c)[off] = __builtin_bit_cast(uint16_t, (__bf16)v);
          else g.c[off] = v;
        }
      }
  }
}

static int launch_bf16(const petr_gemm_args& g, hipStream_t s) {
  const bool staged = g.K % 32 == 0 && (long)g.M * g.N * g.nb0 * g.nb1 >= 128L * 128 * 64;
  PETR_CHECK(g.a_kcontig || staged, PETR_ERR_UNSUPPORTED, "gemm: PETR_GEMM_BF16 with a K-major A needs K %% 32 == 0 and >= 64 output tiles");
  if (staged) {     // enough 128 x 128 tiles: the LDS-staged kernel
    const int tm = (int)cdiv(g.M, 128), tn = (int)cdiv(g.N, 128);
    dim3 grid2(tm * tn, 1, g.nb0 * g.nb1), block2(256);
    if (!g.a_kcontig) hipLaunchKernelGGL((gemm_bf16_lds_kernel<false, true>), grid2, block2, 0, s, g, tn);
    else if (g.a2) hipLaunchKernelGGL((gemm_bf16_lds_kernel<true, false>), grid2, block2, 0, s, g, tn);
    else hipLaunchKernelGGL((gemm_bf16_lds_kernel<false, false>), grid2, block2, 0, s, g, tn);
    PETR_LAUNCH_CHECK("gemm_bf16_lds");
    return PETR_OK;
  }
  const int tiles_m = (int)cdiv(g.M, 64), tiles_n = (int)cdiv(g.N, 128);
  dim3 grid(tiles_m * tiles_n, 1, g.nb0 * g.nb1), block(256);
  if (g.a2) hipLaunchKernelGGL((gemm_bf16_kernel<true>), grid, block, 0, s, g, tiles_n);
  else hipLaunchKernelGGL((gemm_bf16_kernel<false>), grid, block, 0, s, g, tiles_n);
  PETR_LAUNCH_CHECK("gemm_bf16");
  return PETR_OK;
}

extern "C" int petr_gemm(const petr_gemm_args* gp, void* stream) {
  PETR_CHECK(gp && gp->a && gp->b && gp->c, PETR_ERR_INVALID, "gemm: null pointer");
  petr_gemm_args g = *gp;
  PETR_CHECK(g.M > 0 && g.N > 0 && g.K > 0, PETR_ERR_INVALID, "gemm: bad shape M=%d N=%d K=%d", g.M, g.N, g.K);
  if (g.nb0 <= 0) g.nb0 = 1;
  if (g.nb1 <= 0) g.nb1 = 1;
  if (g.split_k <= 0) g.split_k = 1;
  if (g.alpha == 0.f) g.alpha = 1.f;
  {
    // the kernels read g.drop as a DropDev (same 16 bytes): derived keys instead of (seed, site, p)
    static_assert(sizeof(petr_dropout) == sizeof(DropDev), "petr_dropout / DropDev must have the same size");
    PETR_CHECK(gp->drop.p >= 0.f && gp->drop.p < 1.f, PETR_ERR_INVALID, "gemm: dropout p=%g outside [0,1)", (double)gp->drop.p);
    PETR_CHECK(!(gp->drop.p > 0.f) || (g.split_k == 1 && g.nb0 == 1 && g.nb1 == 1 &&
                                       !(g.flags & (PETR_GEMM_ATOMIC | PETR_GEMM_ACCUMULATE))),
               PETR_ERR_UNSUPPORTED, "gemm: dropout needs a plain single-batch store");
    const DropDev dd = make_drop(gp->drop);
    memcpy(&g.drop, &dd, sizeof dd);
  }
  PETR_CHECK(!(g.a2 && !g.a_kcontig), PETR_ERR_UNSUPPORTED, "gemm: a2 addend needs a K-contiguous A");
  PETR_CHECK(!((g.flags & (PETR_GEMM_RELU_MASK | PETR_GEMM_SIGMOID_MUL)) && !g.r), PETR_ERR_INVALID,
             "gemm: mask/mul flag without r operand");
  PETR_CHECK(!(g.split_k > 1 && g.c_split_stride <= 0 && !(g.flags & PETR_GEMM_ATOMIC)), PETR_ERR_INVALID,
             "gemm: split_k needs c_split_stride (or PETR_GEMM_ATOMIC)");
  PETR_CHECK(!((g.flags & (PETR_GEMM_A_BF16 | PETR_GEMM_B_BF16 | PETR_GEMM_R_BF16)) && !(g.flags & PETR_GEMM_BF16)), PETR_ERR_INVALID,
             "gemm: bf16 source operands need PETR_GEMM_BF16");
  PETR_CHECK(!((g.flags & PETR_GEMM_ATOMIC) && (g.bias || g.r || (g.flags & ~(PETR_GEMM_ATOMIC | PETR_GEMM_BF16 | PETR_GEMM_A_BF16 |
                                                                              PETR_GEMM_B_BF16)))), PETR_ERR_UNSUPPORTED,
             "gemm: PETR_GEMM_ATOMIC excludes bias/residual/other flags");
  PETR_CHECK((long)g.nb0 * g.nb1 * g.split_k <= 65535, PETR_ERR_UNSUPPORTED, "gemm: too many batches");
  PETR_CHECK(g.k_seg <= 0 || g.K % g.k_seg == 0, PETR_ERR_INVALID, "gemm: K=%d is not a multiple of k_seg=%d", g.K, g.k_seg);
  const int kseg = g.k_seg > 0 ? g.k_seg : g.K;
  const bool vec = operand_vec_ok(g.a, g.lda, g.a_bs0, g.a_bs1, g.a_seg_stride, g.a_kcontig, g.M, kseg) &&
                   (!g.a2 || aligned16(g.a2)) &&
                   operand_vec_ok(g.b, g.ldb, g.b_bs0, g.b_bs1, g.b_seg_stride, g.b_kcontig, g.N, kseg);
  // the tiled kernel addresses a tile with 32-bit byte offsets from a 64-bit uniform base
  PETR_CHECK(((long)(g.a_kcontig ? g.M : kseg) * g.lda + (g.a_kcontig ? kseg : g.M)) * 4 < (1L << 32) &&
                 ((long)(g.b_kcontig ? g.N : kseg) * g.ldb + (g.b_kcontig ? kseg : g.N)) * 4 < (1L << 32),
             PETR_ERR_UNSUPPORTED, "gemm: one operand batch/segment must span < 4 GiB");
  hipStream_t s = (hipStream_t)stream;
  // Few output tiles AND a long contraction: the latency-optimised kernel (K split inside the workgroup, no
  // LDS staging, no atomics).  It re-reads operands once per tile row/column, so it only pays while the
  // output is small (<= 256 tiles of 32x32: 900x256, 256x256, 256x10 ...); measured on MI355X (same-box A/B of the whole
  // step) it is ~1 % of the step ahead of the tiled kernel at K = 256 and 2x faster at K >= 1024 for K-contiguous operands (float4 fragment loads);
  // with row-contiguous operands (gradients) its 4-byte loads lose to the tiled kernel.  Slabs stay tiled.
  PETR_CHECK(!(g.flags & PETR_GEMM_STORE_BF16) || (g.split_k == 1 && !(g.flags & (PETR_GEMM_ACCUMULATE | PETR_GEMM_ATOMIC))),
             PETR_ERR_UNSUPPORTED, "gemm: PETR_GEMM_STORE_BF16 needs a plain store (no accumulate / atomic / split_k)");
  if (g.flags & PETR_GEMM_BF16) {
    const bool has_drop = gp->drop.p > 0.f;      // only the deep-step kernel of gemm_bf16.hip has the dropout epilogue
    // forward-shaped requests (K-contiguous B, plain epilogue) keep gemm.hip's two kernels; everything else - K-major B,
    // K segments, K slices, accumulate / atomic / ReLU-mask epilogues, bias-gradient column sums: the gradient
    // contractions of the bf16 training step - goes to the general kernel of gemm_bf16.hip
    const bool staged = g.K % 32 == 0 && (long)g.M * g.N * g.nb0 * g.nb1 >= 128L * 128 * 64;
    const bool simple = !has_drop && g.b_kcontig && g.K % 16 == 0 && g.split_k == 1 && g.k_seg <= 0 && !g.a_colsum &&
                        !(g.flags & (PETR_GEMM_ACCUMULATE | PETR_GEMM_ATOMIC | PETR_GEMM_RELU_MASK | PETR_GEMM_SIGMOID_MUL |
                                     PETR_GEMM_A_BF16 | PETR_GEMM_B_BF16 | PETR_GEMM_R_BF16)) && vec &&
                        (!g.a_kcontig || !(g.lda & 3)) && !(g.ldb & 3) && (g.a_kcontig || staged);
    // PETR_GEMM16_SIMPLE=0: everything through gemm_bf16.hip (same-box A/B of the two families)
    static const bool simple_on = petr_tune("PETR_GEMM16_SIMPLE", 1) != 0;
    if (simple && (simple_on || g.a2)) return launch_bf16(g, s);
    return petr_gemm_bf16_general(g, s);
  }
  PETR_CHECK(!((g.flags & PETR_GEMM_BIAS_M) && (g.flags & PETR_GEMM_BF16)), PETR_ERR_UNSUPPORTED,
             "gemm: PETR_GEMM_BIAS_M (bias per output row) is implemented by the fp32 tiled kernel only");
  const long tiles32 = cdiv(g.M, 32) * cdiv(g.N, 32) * (long)g.nb0 * g.nb1;
  const bool slabs = g.split_k > 1 && !(g.flags & PETR_GEMM_ATOMIC);
  // (round 2, same-box A/B: routing input gradients (B read K-major) or weight gradients (both K-major) through this kernel as
  // well lost again - c5 fp32 5.16 -> 5.20 / 5.48 ms per step - so it stays with K-contiguous operands)
  if (!slabs && tiles32 <= 256 && g.K >= 256 && g.a_kcontig && g.b_kcontig &&
      !(g.flags & (PETR_GEMM_ATOMIC | PETR_GEMM_STORE_BF16 | PETR_GEMM_BIAS_M)) && g.k_seg <= 0) {
    petr_gemm_args q = g;
    q.split_k = 1;
    return vec ? launch_skinny<true>(q, s) : launch_skinny<false>(q, s);
  }
  if (!vec) return launch_cfg<64, 64, 32, 32, 32, false>(g, s);   // generic scalar-load path (odd shapes)
  // One tile shape for everything: 64 x 64 per workgroup, 32 x 32 per wave.  128 x 64 and 128 x 128 configurations
  // of this kernel lost to it in every same-box A/B (c5 step 5.56 -> 5.33 ms, p4-1600 step 12.67 -> 11.98 ms,
  // 16384 x 4096 x 1024: 81 -> 92 TFLOP/s): with the f32 MFMA at the vector rate neither LDS bandwidth nor operand
  // reuse bounds the loop, so the bigger tiles only cost occupancy (more accumulator registers, fewer waves to hide
  // LDS / global latency and VALU work behind).
  return launch_cfg<64, 64, 32, 32, 32, true>(g, s);
}

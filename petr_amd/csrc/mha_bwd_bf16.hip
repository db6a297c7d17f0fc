// Multi-head attention backward with bf16 operands (head_dim 32): dQ, dK, dV from Q, O, dO (fp32), bf16 K / V and the
// forward's log-sum-exp.  The gradient of petr_mha_fwd_bf16, i.e. of the cross-attention reached from reference
// petr_transformer.py:357-362 when the head runs in bf16 (BASELINE configs 3-5; the reference gets it from autograd
// through bmm / softmax / dropout / bmm under its fp16 hook).
//
// All five products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; LSE, delta = rowsum(dO*O), the softmax
// recomputation and every accumulator stay fp32.  Q is pre-multiplied by scale*log2(e) and rounded exactly like the
// forward does, so the recomputed probabilities are the forward's.
//
// Structure (cdna guide, Appendix B "Attention backward"):
//   workgroup = 8 waves = 256*KT keys of one (batch, head) x a slice of the query tiles; a wave owns KT tiles of 32
//   keys: their K / V rows (three register fragments: K and V as B operands of the score products, K^T as B operand of
//   the dQ product) and their dK^T / dV^T accumulators stay in registers for the whole sweep, so dK / dV need no
//   cross-wave sum.  Per 32-query tile, KEY ON THE LANE:
//     S'  = Q2 K^T  (accumulator initialised to -LSE*log2e + key bias)   -> p  = exp2(S')
//     dP' = dO V^T  (accumulator initialised to -delta)                  -> ds = p * dP'
//     dV^T += dO^T p , dK^T += Q2^T ds : the p / ds accumulators, rounded to bf16, ARE the B operands (k order of the
//                   accumulator rows; the A operands are fetched in that order with ds_read_b64_tr_b16 from the
//                   natural [q][d] images of Q2 and dO - no transposed image, no transposing store);
//     dQ  += ds K  : ds crosses LDS once, per wave and without a barrier (8-byte stores of the packed accumulator,
//                   transposed reads); the eight waves' partial tiles are summed through LDS and added to global dQ
//                   with 128-byte-segment float atomics.
//   ONE barrier per query tile: the partial tiles of tile t are summed right after the barrier that opens tile t+1
//   (double-buffered), the images of tile t+1 are written behind it from registers loaded one tile earlier.
//   256*KT keys per workgroup (fp32 kernel: 128) cut the dQ atomic traffic (0.92 MB per adder and layer) 2-4x.
// Outputs are ACCUMULATED (+=): the caller zero-fills them.  dQ (and dK / dV when the query range is split over
// workgroups) use float atomics, so their low-order bits depend on arrival order.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int QP = 40;        // bf16 elements per row of a [32 q][32 d] tile image (80 bytes)
constexpr int DSP = 40;       // bf16 elements per row of a wave's [32 keys][32 q] dS image
constexpr int RP = 33;        // fp32 pitch of a [32][32] partial tile
constexpr int RED_SLAB = 32 * RP;
constexpr int OFF_IMG = 0;                                   // [2 buffers][Q2, dO][32 * QP] bf16
constexpr int OFF_ROWS = OFF_IMG + 2 * 2 * 32 * QP * 2;      // [2 buffers][-lse2, -delta, row key][32] 4-byte words
constexpr int OFF_DS = OFF_ROWS + 2 * 3 * 32 * 4;            // [8 waves][32 * DSP] bf16
constexpr int OFF_RED = OFF_DS + 8 * 32 * DSP * 2;           // [2 buffers][8 waves][RED_SLAB] fp32
constexpr int SMEM_BYTES = OFF_RED + 2 * 8 * RED_SLAB * 4;
static_assert(OFF_ROWS % 16 == 0 && OFF_DS % 16 == 0 && OFF_RED % 16 == 0, "LDS regions must stay 16-byte aligned");
static_assert(SMEM_BYTES <= 160 * 1024, "LDS budget");

struct MhaBwd16Params {
  petr_mha_bwd_bf16_args a;
  int nkb, q_splits, qtiles_per_split;
  DropDev drop;        // the forward's probability dropout (thr == 0: off)
};

__device__ __forceinline__ s16x4 tr16(const uint16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}
__device__ __forceinline__ bf16x8 cat8(s16x4 lo, s16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ uint2 pack4f(float a, float b, float c, float d) {
  bf16x4 o = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
  return __builtin_bit_cast(uint2, o);
}

// DROP: the forward multiplied the probabilities by keep/(1-p) after the softmax, so with m = keep/(1-p)
//   dV += dO^T (p*m) ,  dP = m * (dO V^T) ,  ds = p * (dP - delta)   (delta = rowsum(dO*O) of the DROPPED output)
// and the masks are regenerated from (seed, site, row, key), row = (b*H+h)*Q + q.
template <int KT, bool HAS_MASK, bool DROP>
__global__ __launch_bounds__(512) void mha_bwd_bf16_kernel(const MhaBwd16Params p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  uint16_t* img = reinterpret_cast<uint16_t*>(smem + OFF_IMG);
  float* rows_s = reinterpret_cast<float*>(smem + OFF_ROWS);
  uint16_t* ds_all = reinterpret_cast<uint16_t*>(smem + OFF_DS);
  float* red = reinterpret_cast<float*>(smem + OFF_RED);

  const petr_mha_bwd_bf16_args& a = p.a;
  const int total = p.nkb * a.B * a.H * p.q_splits;
  const int w = xcd_remap(blockIdx.x, total);
  const int qs = w % p.q_splits;
  const int rest = w / p.q_splits;
  const int kb = rest % p.nkb;
  const int bh = rest / p.nkb;
  const int b = bh / a.H, hd = bh - b * a.H;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int key_base = kb * (256 * KT) + wave * (32 * KT);

  const float* qp = a.q + (long)b * a.q_bs + (long)hd * a.q_hs;
  const float* gp = a.d_o + (long)b * a.do_bs + (long)hd * a.do_hs;
  const float* op = a.o + (long)b * a.o_bs + (long)hd * a.o_hs;
  const uint16_t* kp = a.k + (long)b * a.k_bs + (long)hd * a.k_hs;
  const uint16_t* vp = a.v + (long)b * a.v_bs + (long)hd * a.v_hs;

  // ---- this wave's keys: three register fragments per 32-key tile ----
  //   kB / vB [i][s]: B operand of S' / dP' (lane (c,h): row key c, d = 16 s + 8 h .. + 7, one 16-byte load)
  //   kT [i][s]     : B operand of dQ     (lane (c,h): column d = c, keys 16 s + 8 h .. + 7 of the tile)
  bf16x8 kB[KT][2], vB[KT][2], kT[KT][2];
  float kbias[KT];
  bool tile_dead[KT];
#pragma unroll
  for (int i = 0; i < KT; ++i) {
    const int key = key_base + 32 * i + c;
    const bool ok = key < a.L;
    const int key_ld = min(key, a.L - 1);
    const uint16_t* kr = kp + (long)key_ld * a.k_rs + 8 * h;
    const uint16_t* vr = vp + (long)key_ld * a.v_rs + 8 * h;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 kx = *reinterpret_cast<const uint4*>(kr + 16 * s);
      uint4 vx = *reinterpret_cast<const uint4*>(vr + 16 * s);
      if (!ok) kx = vx = make_uint4(0, 0, 0, 0);
      kB[i][s] = __builtin_bit_cast(bf16x8, kx);
      vB[i][s] = __builtin_bit_cast(bf16x8, vx);
      typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
      u16x8 tt;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kk = key_base + 32 * i + 16 * s + 8 * h + j;
        const uint16_t x = kp[(long)min(kk, a.L - 1) * a.k_rs + c];
        tt[j] = kk < a.L ? x : (uint16_t)0;
      }
      kT[i][s] = __builtin_bit_cast(bf16x8, tt);
    }
    float kbv = ok ? 0.f : -INFINITY;
    if (HAS_MASK && ok && a.kpm[(long)b * a.L + key]) kbv = -INFINITY;
    kbias[i] = kbv;
    tile_dead[i] = __builtin_amdgcn_ballot_w64(kbv != 0.f) != 0;     // wave-uniform
  }

  f32x16 dKt[KT], dVt[KT];
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dKt[i][r] = dVt[i][r] = 0.f;

  const int qtiles = (a.Q + 31) / 32;
  const int qt_begin = qs * p.qtiles_per_split;
  const int qt_end = min(qtiles, qt_begin + p.qtiles_per_split);
  const float sc2 = a.scale * LOG2E;

  // ---- roles: waves 0..3 stage the query tiles (global loads -> LDS images), waves 4..7 sum the partial dQ tiles and
  //      issue the float atomics.  A wave that did both would wait for its loads with s_waitcnt vmcnt(0) (hipcc cannot
  //      count a wait across the loop's back edge) and so drain the atomics it has just issued - ~3 000 cycles under
  //      load, once per query tile (first version: 228 us at L = 24 000; the ISA showed vmcnt(0) right behind the atomics).
  const bool stager = __builtin_amdgcn_readfirstlane(t >> 6) < 4;          // wave-uniform, provably
  const int srow = (t & 255) >> 3, sc4 = t & 7;
  float4 rq = make_float4(0.f, 0.f, 0.f, 0.f), rg = rq, ro = rq;
  float lreg = 0.f;
  auto gload = [&](int qt) {
    const int row = min(qt * 32 + srow, a.Q - 1);        // rows beyond Q re-read row Q-1 (zeroed at the LDS store)
    rq = *reinterpret_cast<const float4*>(qp + (long)row * a.q_rs + 4 * sc4);
    rg = *reinterpret_cast<const float4*>(gp + (long)row * a.do_rs + 4 * sc4);
    ro = *reinterpret_cast<const float4*>(op + (long)row * a.o_rs + 4 * sc4);
    lreg = a.lse[(long)bh * a.Q + min(qt * 32 + (t & 31), a.Q - 1)];
  };
  auto stage = [&](int qt, int buf) {
    const bool ok = qt * 32 + srow < a.Q;
    float* rw = rows_s + buf * 96;
    const uint2 vq = pack4f(ok ? rq.x * sc2 : 0.f, ok ? rq.y * sc2 : 0.f, ok ? rq.z * sc2 : 0.f, ok ? rq.w * sc2 : 0.f);
    *reinterpret_cast<uint2*>(img + (buf * 2 + 0) * 32 * QP + srow * QP + 4 * sc4) = vq;
    const uint2 vg = pack4f(ok ? rg.x : 0.f, ok ? rg.y : 0.f, ok ? rg.z : 0.f, ok ? rg.w : 0.f);
    *reinterpret_cast<uint2*>(img + (buf * 2 + 1) * 32 * QP + srow * QP + 4 * sc4) = vg;
    // delta from the ROUNDED dO, the one the dP' product multiplies: sum_k ds[q][k] = sum_k p (dP' - delta) must vanish
    // (softmax shift invariance), and with near-uniform attention over thousands of keys dQ = sum_k ds K is the small
    // residual of that cancellation - a delta formed from the unrounded dO would leave a coherent 2^-9 |dO||O| offset on
    // every ds of the row
    const bf16x4 rb = __builtin_bit_cast(bf16x4, vg);
    float dl = ((float)rb[0] * ro.x + (float)rb[1] * ro.y) + ((float)rb[2] * ro.z + (float)rb[3] * ro.w);   // 8 lanes share a row
    dl += __shfl_xor(dl, 1, 64);
    dl += __shfl_xor(dl, 2, 64);
    dl += __shfl_xor(dl, 4, 64);
    if (sc4 == 0) rw[32 + srow] = ok ? -dl : 0.f;
    if (t < 32) {
      const bool rok = qt * 32 + t < a.Q;
      rw[t] = rok ? -lreg * LOG2E : -INFINITY;         // rows beyond Q: p = exp2(-inf) = 0
      if (DROP) reinterpret_cast<uint32_t*>(rw)[64 + t] = drop_row_key(p.drop, (uint32_t)(bh * a.Q + min(qt * 32 + t, a.Q - 1)));
    }
  };
  // sum the eight waves' dQ partial tiles of query tile qt and add them to global dQ (two 128-byte rows per instruction);
  // called by waves 4..7: thread u = t - 256 owns elements u + 256 j
  auto flush = [&](const float* rb, int qt) {
    float* dq = a.dq + (long)b * a.dq_bs + (long)hd * a.dq_hs + (long)qt * 32 * a.dq_rs;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = (t - 256) + 256 * j;
      const int q = idx >> 5, d = idx & 31;
      const float* s = rb + q * RP + d;
      const float v = ((s[0] + s[RED_SLAB]) + (s[2 * RED_SLAB] + s[3 * RED_SLAB])) +
                      ((s[4 * RED_SLAB] + s[5 * RED_SLAB]) + (s[6 * RED_SLAB] + s[7 * RED_SLAB]));
      if (qt * 32 + q < a.Q) atomicAdd(dq + (long)q * a.dq_rs + d, v * a.scale);
    }
  };

  // transposed-read lane constants: lane l of a 16-lane group supplies row (l&15)>>2, columns 4*(l&3) of a 4 x 16
  // block and receives column (l&15) of its four rows
  const int tr_q = (lane & 15) >> 2;
  const int tr_col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  uint16_t* dsw = ds_all + wave * 32 * DSP;

  if (stager && qt_begin < qt_end) {
    gload(qt_begin);
    stage(qt_begin, 0);
    if (qt_begin + 1 < qt_end) gload(qt_begin + 1);
  }
  for (int qt = qt_begin; qt < qt_end; ++qt) {
    const int buf = (qt - qt_begin) & 1;
    __syncthreads();     // images / row constants of tile qt complete; every dQ partial of tile qt-1 written
    if (stager) {
      if (qt + 1 < qt_end) {
        stage(qt + 1, buf ^ 1);
        if (qt + 2 < qt_end) gload(qt + 2);
      }
    } else if (qt > qt_begin) {
      flush(red + (buf ^ 1) * 8 * RED_SLAB, qt - 1);
    }
    const uint16_t* Qi = img + (buf * 2 + 0) * 32 * QP;
    const uint16_t* Gi = img + (buf * 2 + 1) * 32 * QP;
    const float* rw = rows_s + buf * 96;
    // A operands of the score products: lane (c,h) = query row c, d = 16 s + 8 h .. + 7
    bf16x8 qa[2], ga[2], qT[2], gT[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      qa[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Qi + c * QP + 16 * s + 8 * h));
      ga[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Gi + c * QP + 16 * s + 8 * h));
      // A operands of dK^T / dV^T: lane (c,h) = d column c, queries 16 s + 4 h + {0..3} and 16 s + 8 + 4 h + {0..3}:
      // the k order in which the p / ds accumulator rows arrive
      qT[s] = cat8(tr16(Qi + (16 * s + 4 * h + tr_q) * QP + tr_col), tr16(Qi + (16 * s + 8 + 4 * h + tr_q) * QP + tr_col));
      gT[s] = cat8(tr16(Gi + (16 * s + 4 * h + tr_q) * QP + tr_col), tr16(Gi + (16 * s + 8 + 4 * h + tr_q) * QP + tr_col));
    }
    f32x16 dQp;
#pragma unroll
    for (int r = 0; r < 16; ++r) dQp[r] = 0.f;

#pragma unroll
    for (int i = 0; i < KT; ++i) {
      if (key_base + 32 * i >= a.L) continue;          // wave-uniform: this key tile lies beyond L
      f32x16 S, dP, nd;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 x = *reinterpret_cast<const float4*>(rw + 8 * g + 4 * h);        // -lse2 of rows 8g + 4h .. + 3
        const float4 y = *reinterpret_cast<const float4*>(rw + 32 + 8 * g + 4 * h);   // -delta
        S[4 * g] = x.x; S[4 * g + 1] = x.y; S[4 * g + 2] = x.z; S[4 * g + 3] = x.w;
        nd[4 * g] = y.x; nd[4 * g + 1] = y.y; nd[4 * g + 2] = y.z; nd[4 * g + 3] = y.w;
      }
      if (tile_dead[i]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] += kbias[i];
      }
      if (DROP) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dP[r] = 0.f;
      } else {
        dP = nd;
      }
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[0], kB[i][0], S, 0, 0, 0);
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[1], kB[i][1], S, 0, 0, 0);
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[0], vB[i][0], dP, 0, 0, 0);
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[1], vB[i][1], dP, 0, 0, 0);
      if (DROP) {
        const uint32_t key = (uint32_t)(key_base + 32 * i + c);
        const uint32_t* rk = reinterpret_cast<const uint32_t*>(rw) + 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint4 kx = *reinterpret_cast<const uint4*>(rk + 8 * g + 4 * h);
          const uint32_t kk[4] = {kx.x, kx.y, kx.z, kx.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const float pr = __builtin_amdgcn_exp2f(S[r]);
            const float m = drop_keep(kk[e], key, p.drop.thr) ? p.drop.scale : 0.f;
            dP[r] = pr * (dP[r] * m + nd[r]);      // ds = p * (m * dO.V - delta)
            S[r] = pr * m;                         // dropped probability: B operand of dV
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          S[r] = __builtin_amdgcn_exp2f(S[r]);     // p
          dP[r] = S[r] * dP[r];                    // ds (without the softmax scale)
        }
      }
      bf16x8 pb[2], sb[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pb[s][j] = (__bf16)S[8 * s + j];
          sb[s][j] = (__bf16)dP[8 * s + j];
        }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        dVt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gT[s], pb[s], dVt[i], 0, 0, 0);
        dKt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT[s], sb[s], dKt[i], 0, 0, 0);
      }
      // ds -> this wave's [key][q] image: registers 4g .. 4g+3 are queries 8g + 4h .. + 3 of key c (one 8-byte store)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint4 bits = __builtin_bit_cast(uint4, sb[s]);
        *reinterpret_cast<uint2*>(dsw + c * DSP + 16 * s + 4 * h) = make_uint2(bits.x, bits.y);
        *reinterpret_cast<uint2*>(dsw + c * DSP + 16 * s + 8 + 4 * h) = make_uint2(bits.z, bits.w);
      }
      // dQ += ds K: A = ds^T read back transposed (lane (c,h) = query c, keys 16 s + 8 h .. + 7 of the tile);
      // one wave executes its LDS operations in order, so no barrier separates the stores from these reads
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 da = cat8(tr16(dsw + (16 * s + 8 * h + tr_q) * DSP + tr_col),
                               tr16(dsw + (16 * s + 8 * h + 4 + tr_q) * DSP + tr_col));
        dQp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, kT[i][s], dQp, 0, 0, 0);
      }
    }
    float* rb = red + (buf * 8 + wave) * RED_SLAB;
#pragma unroll
    for (int r = 0; r < 16; ++r) rb[mfma32_row(r, h) * RP + c] = dQp[r];
  }
  __syncthreads();
  if (!stager && qt_begin < qt_end) flush(red + ((qt_end - 1 - qt_begin) & 1) * 8 * RED_SLAB, qt_end - 1);
  __syncthreads();     // every wave is done reading the partial tiles: the slabs become transposition scratch

  // ---- dK / dV: transpose each wave's 32 x 32 accumulators through LDS, then row-major adds ----
  {
    float* tk = red + wave * RED_SLAB;
    float* tv = red + (8 + wave) * RED_SLAB;
    float* dk = a.dk + (long)b * a.dk_bs + (long)hd * a.dk_hs;
    float* dv = a.dv + (long)b * a.dv_bs + (long)hd * a.dv_hs;
    const bool use_atomic = p.q_splits > 1;
#pragma unroll
    for (int i = 0; i < KT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        tk[c * RP + mfma32_row(r, h)] = dKt[i][r] * LN2;      // Q2 carried scale*log2e: dK = scale * ds^T Q = ln2 * ds^T Q2
        tv[c * RP + mfma32_row(r, h)] = dVt[i][r];
      }
      // += with ALL loads of the old values issued before the first store (a load-add-store per element serialises
      // sixteen dependent memory round trips at the end of every workgroup)
      float oldk[16], oldv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int idx = j * 64 + lane;
        const int kg = min(key_base + 32 * i + (idx >> 5), a.L - 1);
        oldk[j] = use_atomic ? 0.f : dk[(long)kg * a.dk_rs + (idx & 31)];
        oldv[j] = use_atomic ? 0.f : dv[(long)kg * a.dv_rs + (idx & 31)];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int idx = j * 64 + lane;
        const int kr = idx >> 5, d = idx & 31;
        const int kg = key_base + 32 * i + kr;
        const float gk = tk[kr * RP + d], gv = tv[kr * RP + d];
        if (kg < a.L) {
          float* pk = dk + (long)kg * a.dk_rs + d;
          float* pv = dv + (long)kg * a.dv_rs + d;
          if (use_atomic) {
            atomicAdd(pk, gk);
            atomicAdd(pv, gv);
          } else {
            *pk = oldk[j] + gk;
            *pv = oldv[j] + gv;
          }
        }
      }
    }
  }
}

// KT (32-key tiles per wave) and the number of query-range splits: workgroups come in rounds of one per CU (99 KB of
// LDS), a round costs per-tile work ~ KT + a fixed part (barrier, partial-tile sum), and every adder of dQ (one per key
// block) / of dK, dV (one per query split beyond the first) pays float-atomic bytes at ~1.3 TB/s chip-wide
void choose_cfg(int B, int H, int Q, int L, int kt_max, int& kt_out, int& qs_out) {
  const int qtiles = (int)cdiv(Q, 32);
  double best = 1e30;
  kt_out = 1; qs_out = 1;
  for (int kt = 1; kt <= kt_max; ++kt)
    for (int s = 1; s <= qtiles && s <= 8; ++s) {
      const int per = (int)cdiv(qtiles, s);
      if ((long)(s - 1) * per >= qtiles) continue;
      const long nkb = cdiv(L, 256 * kt);
      const long wgs = nkb * B * H * s;
      const double compute_us = (double)cdiv(wgs, 256) * per * (0.7 * kt + 0.4);
      const double atomic_mb = (double)B * H * Q * 128e-6 * nkb + (s > 1 ? 2.0 * B * H * (double)L * 128e-6 * s : 0.0);
      const double cost = compute_us + 0.5 * atomic_mb / 1.3;
      if (cost < best - 1e-9) { best = cost; kt_out = kt; qs_out = s; }
    }
}

}  // namespace

extern "C" size_t petr_mha_bwd_bf16_workspace_bytes(int B, int H, int Q, int L) {
  (void)B; (void)H; (void)Q; (void)L;
  return 0;      // delta is formed per query tile inside the kernel; no scratch
}

extern "C" int petr_mha_bwd_bf16(const petr_mha_bwd_bf16_args* ap, void* stream) {
  PETR_CHECK(ap && ap->q && ap->k && ap->v && ap->o && ap->d_o && ap->lse && ap->dq && ap->dk && ap->dv, PETR_ERR_INVALID,
             "mha_bwd_bf16: null pointer");
  PETR_CHECK(ap->B > 0 && ap->H > 0 && ap->Q > 0 && ap->L > 0, PETR_ERR_INVALID, "mha_bwd_bf16: bad shape");
  MhaBwd16Params p;
  p.a = *ap;
  const petr_mha_bwd_bf16_args& a = p.a;
  PETR_CHECK(aligned16(a.k) && aligned16(a.v) && !(a.k_bs & 7) && !(a.k_hs & 7) && !(a.k_rs & 7) && !(a.v_bs & 7) &&
                 !(a.v_hs & 7) && !(a.v_rs & 7),
             PETR_ERR_UNSUPPORTED, "mha_bwd_bf16: K/V rows must be 16-byte aligned (strides multiples of 8 elements)");
  PETR_CHECK(aligned16(a.q) && aligned16(a.d_o) && aligned16(a.o) && !(a.q_bs & 3) && !(a.q_hs & 3) && !(a.q_rs & 3) &&
                 !(a.do_bs & 3) && !(a.do_hs & 3) && !(a.do_rs & 3) && !(a.o_bs & 3) && !(a.o_hs & 3) && !(a.o_rs & 3),
             PETR_ERR_UNSUPPORTED, "mha_bwd_bf16: Q / O / dO rows must be 16-byte aligned (strides multiples of 4 elements)");
  PETR_CHECK(a.drop.p >= 0.f && a.drop.p < 1.f, PETR_ERR_INVALID, "mha_bwd_bf16: dropout p=%g outside [0,1)", (double)a.drop.p);
  PETR_CHECK((long)a.B * a.H * a.Q < (1L << 32), PETR_ERR_UNSUPPORTED, "mha_bwd_bf16: dropout row index needs B*H*Q < 2^32");
  p.drop = make_drop(a.drop);
  // KT = 2 needs 260+ registers as written (hipcc spills 20-42 of them into the tile loop); until its K fragments
  // move to an LDS image it is an opt-in (PETR_MHA_BWD16_KT=2) and the default is one key tile per wave
  int kt = 1, qsp = 1, kt_max = 1;
  if (const char* e = getenv("PETR_MHA_BWD16_KT")) kt_max = atoi(e) == 2 ? 2 : 1;
  choose_cfg(a.B, a.H, a.Q, a.L, kt_max, kt, qsp);
  if (const char* e = getenv("PETR_MHA_BWD16_KT")) {        // tuning overrides
    const int v = atoi(e);
    if (v == 1 || v == 2) kt = v;
  }
  if (const char* e = getenv("PETR_MHA_BWD16_QSPLITS")) {
    const int v = atoi(e);
    if (v >= 1 && v <= (int)cdiv(a.Q, 32)) qsp = v;
  }
  p.nkb = (int)cdiv(a.L, 256 * kt);
  p.q_splits = qsp;
  p.qtiles_per_split = (int)cdiv(cdiv(a.Q, 32), qsp);
  if ((long)(qsp - 1) * p.qtiles_per_split >= cdiv(a.Q, 32)) {     // an override that would leave an empty split
    p.q_splits = (int)cdiv(cdiv(a.Q, 32), p.qtiles_per_split);
  }
  const long total = (long)p.nkb * a.B * a.H * p.q_splits;
  PETR_CHECK(total < (1L << 31), PETR_ERR_UNSUPPORTED, "mha_bwd_bf16: grid too large");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t ev0, ev1;   // null unless bench.py's profiler is on
  petr_prof_claim(PETR_PROF_MHA_BWD + 16 * (a.L > a.Q ? 1 : 0), &ev0, &ev1);
  auto launch = [&](auto kern) { hipExtLaunchKernelGGL(kern, dim3((unsigned)total), dim3(512), 0, s, ev0, ev1, 0, p); };
  const int variant = (kt == 2 ? 4 : 0) | (p.drop.thr ? 2 : 0) | (a.kpm ? 1 : 0);
  switch (variant) {
    case 0: launch(mha_bwd_bf16_kernel<1, false, false>); break;
    case 1: launch(mha_bwd_bf16_kernel<1, true, false>); break;
    case 2: launch(mha_bwd_bf16_kernel<1, false, true>); break;
    case 3: launch(mha_bwd_bf16_kernel<1, true, true>); break;
    case 4: launch(mha_bwd_bf16_kernel<2, false, false>); break;
    case 5: launch(mha_bwd_bf16_kernel<2, true, false>); break;
    case 6: launch(mha_bwd_bf16_kernel<2, false, true>); break;
    default: launch(mha_bwd_bf16_kernel<2, true, true>); break;
  }
  PETR_LAUNCH_CHECK("mha_bwd_bf16");
  return PETR_OK;
}

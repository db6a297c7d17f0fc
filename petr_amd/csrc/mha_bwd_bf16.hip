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
//   workgroup = NW waves = 256 keys of one (batch, head) x a slice of the query tiles; a wave owns KT = 256/(32 NW) tiles of
//   32 keys: their dK^T / dV^T accumulators and their V fragments stay in registers for the whole sweep (so dK / dV need
//   no cross-wave sum), the workgroup's K rows sit in one LDS image that serves both K fragments (row reads for S',
//   transposed reads for dQ).  Per 32-query tile, KEY ON THE LANE:
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
//   256 keys per workgroup (fp32 kernel: 128) halve the dQ atomic traffic (0.92 MB per adder and layer).
// Outputs are ACCUMULATED (+=): the caller zero-fills them.  dQ (and dK / dV when the query range is split over
// workgroups) use float atomics, so their low-order bits depend on arrival order.
#include <stdlib.h>

#include "common.h"

namespace {

#ifndef PETR_BWD16_DIAG
#define PETR_BWD16_DIAG 0      // timing-only ablations (results wrong): 1 no atomics, 2 no flush, 4 no dQ path, 8 no exp,
#endif                         // 16 no dV/dK products, 32 no staging, 64 no S/dP products
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int QP = 40;        // bf16 elements per row of a [32 q][32 d] tile image (80 bytes)
constexpr int DSP = 40;       // bf16 elements per row of a wave's [32 keys][32 q] dS image
constexpr int RP = 33;        // fp32 pitch of a [32][32] partial tile
constexpr int RED_SLAB = 32 * RP;
// LDS layout for NW waves per workgroup (byte offsets):
//   [2 buffers][Q2, dO][32 * QP] bf16 | [2 buffers][-lse2, -delta, row key][32] words | [NW waves][32 * DSP] bf16 (ds images)
//   | [2 buffers][NW waves][RED_SLAB] fp32 (dQ partial tiles; dK / dV transposition scratch at the end)
//   | [NW * KT * 32 keys][QP] bf16 (the workgroup's K rows)
constexpr int OFF_IMG = 0;
constexpr int OFF_ROWS = OFF_IMG + 2 * 2 * 32 * QP * 2;
constexpr int OFF_DS = OFF_ROWS + 2 * 3 * 32 * 4;
constexpr int off_red(int nw) { return OFF_DS + nw * 32 * DSP * 2; }
constexpr int off_kimg(int nw) { return off_red(nw) + 2 * nw * RED_SLAB * 4; }
constexpr int smem_bytes(int nw, int kt) { return off_kimg(nw) + nw * kt * 32 * QP * 2; }
static_assert(OFF_ROWS % 16 == 0 && OFF_DS % 16 == 0 && off_red(8) % 16 == 0 && off_red(4) % 16 == 0 && off_kimg(8) % 16 == 0 &&
                  off_kimg(4) % 16 == 0, "LDS regions must stay 16-byte aligned");
static_assert(smem_bytes(8, 2) <= 160 * 1024 && 2 * smem_bytes(4, 2) <= 160 * 1024, "LDS budget");

struct MhaBwd16Params {
  petr_mha_bwd_bf16_args a;
  int nkb, q_splits, qtiles_per_split;
  DropDev drop;        // the forward's probability dropout (thr == 0: off)
  const uint32_t* drop_bits;   // key-major packed mask (petr_dropout_bits) or null: re-hash
  int nqt32, lpad;             // its dimensions: ceil(Q/32) query tiles, 32 * ceil(L/32) keys per tile
  int pair;                    // heads 2j / 2j+1 of one key block are neighbours in the workgroup order (see mha_fwd_bf16_kernel)
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
// NW waves per workgroup, KT key tiles per wave: (8, 1) = eight waves in lock step on one CU, (4, 2) = four waves with
// twice the keys each, two workgroups per CU (75 KB of LDS, <= 256 registers).  The loop is bound by the LDS pipe, not
// by MFMA (ablations: removing any single stage changes the time by < 8 %; ~70 LDS instructions per wave and tile):
// with (4, 2) the Q2 / dO fragments are fetched once per 64 keys instead of once per 32 and four partial dQ tiles are
// summed instead of eight, i.e. a third less LDS traffic for the same products, and the two co-resident workgroups are
// not tied to each other's barrier.
template <int NW, int KT, bool HAS_MASK, bool DROP>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void mha_bwd_bf16_kernel(const MhaBwd16Params p) {
  constexpr int NT = NW * 64;            // threads
  constexpr int NS = NW * 32;            // stager threads (waves 0 .. NW/2-1) = flusher threads (the other half)
  constexpr int NP = 256 / NS;           // (row, 4 columns) pieces of a 32 x 32 tile per stager thread
  __shared__ __attribute__((aligned(16))) unsigned char smem[smem_bytes(NW, KT)];
  uint16_t* img = reinterpret_cast<uint16_t*>(smem + OFF_IMG);
  float* rows_s = reinterpret_cast<float*>(smem + OFF_ROWS);
  uint16_t* ds_all = reinterpret_cast<uint16_t*>(smem + OFF_DS);
  float* red = reinterpret_cast<float*>(smem + off_red(NW));
  __shared__ uint32_t bits_s[2][NW * 32 * KT];       // packed dropout mask words of the workgroup's keys, per query-tile buffer

  const petr_mha_bwd_bf16_args& a = p.a;
  const int total = p.nkb * a.B * a.H * p.q_splits;
  const int w = xcd_remap(blockIdx.x, total);
  const int qs = w % p.q_splits;
  int rest = w / p.q_splits;
  const int h0 = p.pair ? (rest & 1) : 0;
  if (p.pair) rest >>= 1;
  const int kb = rest % p.nkb;
  const int bh = p.pair ? 2 * (rest / p.nkb) + h0 : rest / p.nkb;
  const int b = bh / a.H, hd = bh - b * a.H;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int key_base = kb * (NW * 32 * KT) + wave * (32 * KT);

  const float* qp = a.q + (long)b * a.q_bs + (long)hd * a.q_hs;
  const float* gp = a.d_o + (long)b * a.do_bs + (long)hd * a.do_hs;
  const float* op = a.o + (long)b * a.o_bs + (long)hd * a.o_hs;
  const uint16_t* kp = a.k + (long)b * a.k_bs + (long)hd * a.k_hs;
  const uint16_t* vp = a.v + (long)b * a.v_bs + (long)hd * a.v_hs;

  // ---- this workgroup's keys ----
  //   K: one LDS image [256*KT keys][32 d] (80-byte pitch), filled once; per query tile a wave reads from it
  //        kB[s]: B operand of S'  (lane (c,h): row key c, d = 16 s + 8 h .. + 7: one 16-byte read)
  //        kT[s]: B operand of dQ  (lane (c,h): column d = c, keys 16 s + 8 h .. + 7: two transposed 8-byte reads)
  //      (in registers these two fragments cost 16 per key tile; with them there KT = 2 spilled 25-42 registers)
  //   V: vB[i][s], B operand of dP', in registers (8 per key tile)
  uint16_t* kimg = reinterpret_cast<uint16_t*>(smem + off_kimg(NW));
  {
    const int wg_key0 = kb * (NW * 32 * KT);
#pragma unroll
    for (int j = 0; j < 2 * KT; ++j) {
      const int idx = t + NT * j;                   // 16-byte piece: key idx >> 2, d = 8 * (idx & 3)
      const int kl = idx >> 2, pc = idx & 3;
      const int kg = wg_key0 + kl;
      uint4 x = *reinterpret_cast<const uint4*>(kp + (long)min(kg, a.L - 1) * a.k_rs + 8 * pc);
      if (kg >= a.L) x = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(kimg + kl * QP + 8 * pc) = x;
    }
  }
  const uint16_t* kw = kimg + wave * (32 * KT) * QP;      // this wave's 32*KT key rows
  // KT == 1 (the default 8-wave shape): the two K fragments stay in registers, filled once from the image (180 vs 192 us
  // at L = 24 000 against re-reading them per query tile); KT == 2 reads them from the image (register budget)
  constexpr bool KREG = KT == 1;
  bf16x8 kBr[2], kTr[2];
  bf16x8 vB[KT][2];
  float kbias[KT];
  bool tile_dead[KT];
#pragma unroll
  for (int i = 0; i < KT; ++i) {
    const int key = key_base + 32 * i + c;
    const bool ok = key < a.L;
    const int key_ld = min(key, a.L - 1);
    const uint16_t* vr = vp + (long)key_ld * a.v_rs + 8 * h;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 vx = *reinterpret_cast<const uint4*>(vr + 16 * s);
      if (!ok) vx = make_uint4(0, 0, 0, 0);
      vB[i][s] = __builtin_bit_cast(bf16x8, vx);
    }
    float kbv = ok ? 0.f : -INFINITY;
    if (HAS_MASK && ok && a.kpm[(long)b * a.L + key]) kbv = -INFINITY;
    kbias[i] = kbv;
    tile_dead[i] = __builtin_amdgcn_ballot_w64(kbv != 0.f) != 0;     // wave-uniform
  }

  if (KREG) {
    __syncthreads();                 // the image is complete
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      kBr[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(kw + c * QP + 16 * s + 8 * h));
      kTr[s] = cat8(tr16(kw + (16 * s + 8 * h + ((lane & 15) >> 2)) * QP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)),
                    tr16(kw + (16 * s + 8 * h + 4 + ((lane & 15) >> 2)) * QP + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)));
    }
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
  const bool stager = __builtin_amdgcn_readfirstlane(t >> 6) < NW / 2;     // wave-uniform, provably
  float4 rq[NP], rg[NP], ro[NP];
  float lreg = 0.f;
  uint32_t breg[KT];
#pragma unroll
  for (int i = 0; i < KT; ++i) breg[i] = 0u;
  const bool use_bits = DROP && p.drop_bits != nullptr;
#pragma unroll
  for (int j = 0; j < NP; ++j) rq[j] = rg[j] = ro[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto gload = [&](int qt) {
    if (use_bits) {       // the mask words of the workgroup's keys for this query tile travel with the tile (stager waves:
#pragma unroll            // the flusher waves' vector-memory queue holds atomics only)
      for (int i = 0; i < KT; ++i) {
        const int kidx = kb * (NW * 32 * KT) + (t & (NS - 1)) + NS * i;          // this thread's key: block kidx >> 5, slot of kidx & 31
        breg[i] = p.drop_bits[((long)bh * p.nqt32 + qt) * p.lpad + min(kidx & ~31, p.lpad - 32) + petr_bits_slot(kidx & 31)];
      }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int pc = (t & (NS - 1)) + NS * j;              // piece: row pc >> 3, columns 4 (pc & 7) ..
      const int row = min(qt * 32 + (pc >> 3), a.Q - 1);   // rows beyond Q re-read row Q-1 (zeroed at the LDS store)
      const int c4 = pc & 7;
      rq[j] = *reinterpret_cast<const float4*>(qp + (long)row * a.q_rs + 4 * c4);
      rg[j] = *reinterpret_cast<const float4*>(gp + (long)row * a.do_rs + 4 * c4);
      ro[j] = *reinterpret_cast<const float4*>(op + (long)row * a.o_rs + 4 * c4);
    }
    lreg = a.lse[(long)bh * a.Q + min(qt * 32 + (t & 31), a.Q - 1)];
  };
  auto stage = [&](int qt, int buf) {
    float* rw = rows_s + buf * 96;
    if (use_bits) {
#pragma unroll
      for (int i = 0; i < KT; ++i) bits_s[buf][(t & (NS - 1)) + NS * i] = breg[i];
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int pc = (t & (NS - 1)) + NS * j;
      const int srow = pc >> 3, sc4 = pc & 7;
      const bool ok = qt * 32 + srow < a.Q;
      const uint2 vq = pack4f(ok ? rq[j].x * sc2 : 0.f, ok ? rq[j].y * sc2 : 0.f, ok ? rq[j].z * sc2 : 0.f, ok ? rq[j].w * sc2 : 0.f);
      *reinterpret_cast<uint2*>(img + (buf * 2 + 0) * 32 * QP + srow * QP + 4 * sc4) = vq;
      const uint2 vg = pack4f(ok ? rg[j].x : 0.f, ok ? rg[j].y : 0.f, ok ? rg[j].z : 0.f, ok ? rg[j].w : 0.f);
      *reinterpret_cast<uint2*>(img + (buf * 2 + 1) * 32 * QP + srow * QP + 4 * sc4) = vg;
      // delta from the ROUNDED dO, the one the dP' product multiplies: sum_k ds[q][k] = sum_k p (dP' - delta) must vanish
      // (softmax shift invariance), and with near-uniform attention over thousands of keys dQ = sum_k ds K is the small
      // residual of that cancellation - a delta formed from the unrounded dO would leave a coherent 2^-9 |dO||O| offset
      // on every ds of the row
      const bf16x4 rb = __builtin_bit_cast(bf16x4, vg);
      float dl = ((float)rb[0] * ro[j].x + (float)rb[1] * ro[j].y) + ((float)rb[2] * ro[j].z + (float)rb[3] * ro[j].w);
      dl += __shfl_xor(dl, 1, 64);       // 8 lanes share a row
      dl += __shfl_xor(dl, 2, 64);
      dl += __shfl_xor(dl, 4, 64);
      if (sc4 == 0) rw[32 + srow] = ok ? -dl : 0.f;
    }
    if (t < 32) {
      const bool rok = qt * 32 + t < a.Q;
      rw[t] = rok ? -lreg * LOG2E : -INFINITY;         // rows beyond Q: p = exp2(-inf) = 0
      if (DROP) reinterpret_cast<uint32_t*>(rw)[64 + t] = drop_row_key(p.drop, (uint32_t)(bh * a.Q + min(qt * 32 + t, a.Q - 1)));
    }
  };
  // sum the NW waves' dQ partial tiles of query tile qt and add them to global dQ (two 128-byte rows per instruction);
  // called by the flusher waves: thread u = t - NS owns elements u + NS j
  auto flush = [&](const float* rb, int qt) {
    float* dq = a.dq + (long)b * a.dq_bs + (long)hd * a.dq_hs + (long)qt * 32 * a.dq_rs;
#pragma unroll
    for (int j = 0; j < 1024 / NS; ++j) {
      const int idx = (t - NS) + NS * j;
      const int q = idx >> 5, d = idx & 31;
      const float* s = rb + q * RP + d;
      float v = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NW; w2 += 2) v += s[w2 * RED_SLAB] + s[(w2 + 1) * RED_SLAB];
      if (PETR_BWD16_DIAG & 1) { if (v == 123.456f) dq[q] = v; }
      else if (qt * 32 + q < a.Q) atomicAdd(dq + (long)q * a.dq_rs + d, v * a.scale);
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
      if (qt + 1 < qt_end && !(PETR_BWD16_DIAG & 32)) {
        stage(qt + 1, buf ^ 1);
        if (qt + 2 < qt_end) gload(qt + 2);
      }
    } else if (qt > qt_begin && !(PETR_BWD16_DIAG & 2)) {
      flush(red + (buf ^ 1) * NW * RED_SLAB, qt - 1);
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
      f32x16 S, dP;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 x = *reinterpret_cast<const float4*>(rw + 8 * g + 4 * h);        // -lse2 of rows 8g + 4h .. + 3
        S[4 * g] = x.x; S[4 * g + 1] = x.y; S[4 * g + 2] = x.z; S[4 * g + 3] = x.w;
        if (!DROP) {                                                                  // -delta: the dP' accumulator's start
          const float4 y = *reinterpret_cast<const float4*>(rw + 32 + 8 * g + 4 * h);
          dP[4 * g] = y.x; dP[4 * g + 1] = y.y; dP[4 * g + 2] = y.z; dP[4 * g + 3] = y.w;
        }
      }
      if (tile_dead[i]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] += kbias[i];
      }
      if (DROP) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dP[r] = 0.f;
      }
      {
        const uint16_t* kr = kw + (32 * i + c) * QP + 8 * h;
        const bf16x8 kb0 = KREG ? kBr[0] : __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(kr));
        const bf16x8 kb1 = KREG ? kBr[1] : __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(kr + 16));
        if (!(PETR_BWD16_DIAG & 64)) {
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[0], kb0, S, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[1], kb1, S, 0, 0, 0);
        } else { S[0] += (float)kb0[0] + (float)qa[0][0] + (float)kb1[1] + (float)qa[1][1]; }
      }
      if (!(PETR_BWD16_DIAG & 64)) {
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[0], vB[i][0], dP, 0, 0, 0);
      dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[1], vB[i][1], dP, 0, 0, 0);
      } else { dP[0] += (float)ga[0][0] + (float)vB[i][0][0] + (float)ga[1][1] + (float)vB[i][1][1]; }
      if (DROP && use_bits) {
        // one bit test per probability: row mfma32_row(r, h) of this key's word (shifted by 4 h once)
        const uint32_t w = bits_s[buf][wave * 32 * KT + 32 * i + c] >> (4 * h);
        const uint32_t sbits = __float_as_uint(p.drop.scale);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 y = *reinterpret_cast<const float4*>(rw + 32 + 8 * g + 4 * h);   // -delta, read where it is used
          const float nd4[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const float pr = __builtin_amdgcn_exp2f(S[r]);
            const float m = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe(w, e + 8 * g, 1) & sbits);     // scale or 0
            dP[r] = pr * (dP[r] * m + nd4[e]);     // ds = p * (m * dO.V - delta)
            S[r] = pr * m;                         // dropped probability: B operand of dV
          }
        }
      } else if (DROP) {
        const uint32_t key = (uint32_t)(key_base + 32 * i + c);
        const uint32_t* rk = reinterpret_cast<const uint32_t*>(rw) + 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const uint4 kx = *reinterpret_cast<const uint4*>(rk + 8 * g + 4 * h);
          const uint32_t kk[4] = {kx.x, kx.y, kx.z, kx.w};
          const float4 y = *reinterpret_cast<const float4*>(rw + 32 + 8 * g + 4 * h);   // -delta, read where it is used
          const float nd4[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            const float pr = __builtin_amdgcn_exp2f(S[r]);
            const float m = drop_keep(kk[e], key, p.drop.thr) ? p.drop.scale : 0.f;
            dP[r] = pr * (dP[r] * m + nd4[e]);     // ds = p * (m * dO.V - delta)
            S[r] = pr * m;                         // dropped probability: B operand of dV
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          S[r] = (PETR_BWD16_DIAG & 8) ? S[r] * 0.5f : __builtin_amdgcn_exp2f(S[r]);     // p
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
        if (PETR_BWD16_DIAG & 16) { dVt[i][s] += (float)gT[s][0] + (float)pb[s][0]; dKt[i][s] += (float)qT[s][0] + (float)sb[s][1]; continue; }
        dVt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gT[s], pb[s], dVt[i], 0, 0, 0);
        dKt[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT[s], sb[s], dKt[i], 0, 0, 0);
      }
      // ds -> this wave's [key][q] image: registers 4g .. 4g+3 are queries 8g + 4h .. + 3 of key c (one 8-byte store)
      if (PETR_BWD16_DIAG & 4) { dQp[0] += (float)sb[0][0] + (float)sb[1][1]; continue; }
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
        const bf16x8 kt = KREG ? kTr[s]
                               : cat8(tr16(kw + (32 * i + 16 * s + 8 * h + tr_q) * QP + tr_col),
                                      tr16(kw + (32 * i + 16 * s + 8 * h + 4 + tr_q) * QP + tr_col));
        dQp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, kt, dQp, 0, 0, 0);
      }
    }
    float* rb = red + (buf * NW + wave) * RED_SLAB;
#pragma unroll
    for (int r = 0; r < 16; ++r) rb[mfma32_row(r, h) * RP + c] = dQp[r];
  }
  __syncthreads();
  if (!stager && qt_begin < qt_end) flush(red + ((qt_end - 1 - qt_begin) & 1) * NW * RED_SLAB, qt_end - 1);
  __syncthreads();     // every wave is done reading the partial tiles: the slabs become transposition scratch

  // ---- dK / dV: transpose each wave's 32 x 32 accumulators through LDS, then row-major adds ----
  {
    float* tk = red + wave * RED_SLAB;
    float* tv = red + (NW + wave) * RED_SLAB;
    // element strides count floats, or bf16 values with dkv_bf16: the base offset is applied in the matching unit
    float* dk = a.dkv_bf16 ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(a.dk) + (long)b * a.dk_bs + (long)hd * a.dk_hs)
                           : a.dk + (long)b * a.dk_bs + (long)hd * a.dk_hs;
    float* dv = a.dkv_bf16 ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(a.dv) + (long)b * a.dv_bs + (long)hd * a.dv_hs)
                           : a.dv + (long)b * a.dv_bs + (long)hd * a.dv_hs;
    const bool use_atomic = p.q_splits > 1;
    const bool overwrite = a.dkv_overwrite != 0;          // host guarantees q_splits == 1
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
        oldk[j] = (use_atomic || overwrite) ? 0.f : dk[(long)kg * a.dk_rs + (idx & 31)];
        oldv[j] = (use_atomic || overwrite) ? 0.f : dv[(long)kg * a.dv_rs + (idx & 31)];
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
          if (a.dkv_bf16) {          // bf16 stores: same element offsets on 2-byte elements
            reinterpret_cast<uint16_t*>(dk)[(long)kg * a.dk_rs + d] = __builtin_bit_cast(uint16_t, (__bf16)gk);
            reinterpret_cast<uint16_t*>(dv)[(long)kg * a.dv_rs + d] = __builtin_bit_cast(uint16_t, (__bf16)gv);
          } else if (use_atomic) {
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

// Number of query-range splits: every workgroup covers 256 keys; workgroups come in rounds of `per_cu` per CU, a round
// costs a fixed part (~10 us: K image, first tile, dK / dV epilogue) plus ~1.4 us per query tile, and every adder of dQ
// (one per key block) / of dK, dV (one per query split beyond the first) pays float-atomic bytes at ~1.3 TB/s chip-wide
int choose_qsplits(int B, int H, int Q, int L, int per_cu) {
  const int qtiles = (int)cdiv(Q, 32);
  double best = 1e30;
  int qs_out = 1;
  for (int s = 1; s <= qtiles && s <= 8; ++s) {
    const int per = (int)cdiv(qtiles, s);
    if ((long)(s - 1) * per >= qtiles) continue;
    const long nkb = cdiv(L, 256);
    const long wgs = nkb * B * H * s;
    const double compute_us = (double)cdiv(wgs, 256L * per_cu) * (10.0 + 1.4 * per * per_cu);
    const double atomic_mb = (double)B * H * Q * 128e-6 * nkb + (s > 1 ? 2.0 * B * H * (double)L * 128e-6 * s : 0.0);
    const double cost = compute_us + 0.5 * atomic_mb / 1.3;
    if (cost < best - 1e-9) { best = cost; qs_out = s; }
  }
  return qs_out;
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
  p.drop_bits = a.drop.p > 0.f ? a.drop_bits : nullptr;
  p.nqt32 = (int)cdiv(a.Q, 32);
  p.lpad = 32 * (int)cdiv(a.L, 32);
  // (NW, KT) = (8, 1): eight waves with one 32-key tile each, one workgroup per CU.  The (4, 2) instantiation of the template -
  // four waves with two key tiles each, two workgroups per CU - was built to cut the LDS traffic per product by a third, but
  // hipcc spills 27-40 registers of it into the tile loop and it measured 1.5-2x slower (L = 24 000: 356 vs 192 us): not compiled
  // into the library any more
  int qsp = choose_qsplits(a.B, a.H, a.Q, a.L, 1);
  PETR_CHECK(!a.dkv_bf16 || a.dkv_overwrite, PETR_ERR_INVALID, "mha_bwd_bf16: dkv_bf16 needs dkv_overwrite");
  if (a.dkv_overwrite) qsp = 1;        // a stored dK / dV needs the whole query range in one workgroup per key block
  p.nkb = (int)cdiv(a.L, 256);
  p.q_splits = qsp;
  p.qtiles_per_split = (int)cdiv(cdiv(a.Q, 32), qsp);
  if ((long)(qsp - 1) * p.qtiles_per_split >= cdiv(a.Q, 32)) {     // an override that would leave an empty split
    p.q_splits = (int)cdiv(cdiv(a.Q, 32), p.qtiles_per_split);
  }
  static const int pair_on = petr_tune("PETR_MHA16_PAIR", 1) != 0;
  p.pair = pair_on && !(a.H & 1) && a.k_hs == 32 && a.v_hs == 32;
  const long total = (long)p.nkb * a.B * a.H * p.q_splits;
  PETR_CHECK(total < (1L << 31), PETR_ERR_UNSUPPORTED, "mha_bwd_bf16: grid too large");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t ev0, ev1;   // null unless bench.py's profiler is on
  petr_prof_claim(PETR_PROF_MHA_BWD + 16 * (a.L > a.Q ? 1 : 0), &ev0, &ev1);
  auto launch = [&](auto kern) { hipExtLaunchKernelGGL(kern, dim3((unsigned)total), dim3(512), 0, s, ev0, ev1, 0, p); };
  switch ((p.drop.thr ? 2 : 0) | (a.kpm ? 1 : 0)) {
    case 0: launch(mha_bwd_bf16_kernel<8, 1, false, false>); break;
    case 1: launch(mha_bwd_bf16_kernel<8, 1, true, false>); break;
    case 2: launch(mha_bwd_bf16_kernel<8, 1, false, true>); break;
    default: launch(mha_bwd_bf16_kernel<8, 1, true, true>); break;
  }
  PETR_LAUNCH_CHECK("mha_bwd_bf16");
  return PETR_OK;
}

// Multi-head attention backward (head_dim 32, fp32): dQ, dK, dV from Q, K, V, O, dO and the
// forward's log-sum-exp; P is recomputed, the [Q x L] score matrix is never stored.
// Gradient of the attention core reached from reference petr_transformer.py:357-362 (the reference
// gets it from autograd through bmm/softmax/bmm and, because of cp.checkpoint :201-212, re-runs the
// whole layer forward first).
//
// Structure (cdna guide, Appendix B "Attention backward", adapted to fp32 MFMA):
//   workgroup = 4 waves = 128 keys of one (batch, head) pair; each wave owns 32 keys: its K and V rows live in registers
//   for the whole sweep and its dK^T / dV^T accumulators (32x32 each) stay in registers, so dK/dV need no cross-wave sum.
//   Per 32-query tile, with the KEY ON THE LANE:
//     S'  = Q K^T  (acc initialised to -LSE/scale + key bias)      -> p  = exp2(S' * scale*log2e)
//     dP' = dO V^T (acc initialised to -delta)                     -> ds = p * dP'
//     dV^T += dO^T p ,  dK^T += Q^T ds   : p / ds accumulators are used directly as B operands
//     dQ   += ds K  : ds crosses LDS once (key-on-lane -> [query][key] image); each wave then owns a 16 x 16 quadrant of the
//                     dQ tile over ALL 128 keys (v_mfma_f32_16x16x4_f32) and adds it to global dQ with float atomics.
//   LDS tiles of Q and dO use pitch 33 so that both the row-fragment read (lane = query) and the
//   column-fragment read (lane = d) are conflict-free ds_read_b32.
//   delta = rowsum(dO*O) is formed from the O tile that is loaded with Q and dO (no separate pass).
// Outputs are ACCUMULATED (+=): the caller zero-fills them (the executor clears its gradient
// workspace once).  dQ (and dK / dV of a pair shared by several workgroups) use float atomics, so their low-order bits
// depend on arrival order.
#include <stdlib.h>

#include "common.h"
#include <type_traits>

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int P33 = 33;

// Diagnostic build only (-DPETR_DIAG_BWD_STAMPS, scripts/bwd32_stamps.sh): s_memtime stamps at the phase boundaries of
// the tile loop, summed over the tiles of a wave and added (wave 0, lane 0 of every workgroup) to 16 counters at the
// front of the otherwise unused workspace.  No stamp exists in the product build.
#ifdef PETR_DIAG_BWD_STAMPS
#define STAMP(i)                                                  \
  do {                                                            \
    __builtin_amdgcn_sched_barrier(0);                            \
    const uint64_t now__ = __builtin_amdgcn_s_memtime();          \
    __builtin_amdgcn_sched_barrier(0);                            \
    st_acc[i] += now__ - st_last;                                 \
    st_last = now__;                                              \
  } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// =====================================================================================================================
// Persistent schedule (round 3).  What in-kernel stamps of round 2's form - one workgroup per (key block, head, query split),
// dQ partial tiles of the four waves summed through LDS behind a second barrier - showed (scripts/bwd32_stamps.py): the tile loop
// kept the matrix pipe ~80 % busy, but (i) at c5 the uneven query splits still left a third of the workgroup slots' time
// unused and every split workgroup ended with 41 000 cycles of dK / dV float atomics, and (ii) the dQ step took 5 400 of a
// tile's 12 500 cycles for 1 024 cycles of matrix work.  Here:
//   * ONE launch of G = (resident workgroups per CU) x CUs persistent workgroups.  The P (key block, head) pairs are dealt as
//     floor(P / G) WHOLE pairs per workgroup (dK / dV of a whole pair: plain read-add-store, no atomics) and the remaining
//     rem = P mod G pairs as one sequence of rem x nqt query tiles cut into G equal contiguous ranges (a range may cross
//     a pair boundary: the workgroup flushes dK / dV with float atomics and reloads K / V there).  Every workgroup gets the
//     same number of query tiles (+-1) whatever P, Q and L are: no shape-fitted split planner, no dependence on the
//     hardware's dispatch order.  The remainder ranges run FIRST so that their atomics drain under the whole pairs.
//   * One barrier per query tile.  Q / dO / ds images are double-buffered; the tile for iteration t + 2 is staged behind
//     iteration t's barrier.  dQ: wave w owns the (16 query x 16 d) quadrant (w >> 1, w & 1) of the tile and sums it over ALL
//     128 keys of the block with v_mfma_f32_16x16x4_f32 (same matrix cycles as one 32x32x2 chain over 32 keys), reading the
//     four waves' ds tiles [query][key] (pitch 34) and a transposed K image [d][key] (pitch 132) with conflict-free
//     ds_read_b64; the quadrant goes to global dQ with float atomics straight from the accumulator (4 row segments of 64 B
//     per instruction): no partial tiles, no second barrier.
// =====================================================================================================================
constexpr int P34 = 34;
constexpr int KTP = 132;

struct MhaBwdSkParams {
  petr_mha_bwd_args a;
  int nkb, nqt;
  int G;             // persistent workgroups
  int full_rounds;   // whole pairs per workgroup
  int rem;           // pairs whose query tiles are dealt as rem * nqt / G tiles per workgroup
  DropDev drop;
  const uint32_t* drop_bits;
  int nqt32, lpad;
};

struct TileRegs {
  float4 q, g, o;
  float lse;
  uint32_t bits;
};

// DROP: 0 = no dropout, 1 = masks re-hashed, 2 = masks read from the packed bits the forward left
template <bool HAS_MASK, bool VEC, int DROP>
__global__ __launch_bounds__(256, 2) void mha_bwd_sk_kernel(const MhaBwdSkParams p) {
  __shared__ __attribute__((aligned(16))) float Qs[2][32 * P33];
  __shared__ __attribute__((aligned(16))) float dOs[2][32 * P33];
  __shared__ __attribute__((aligned(16))) float dSs[2][4][32 * P34];
  __shared__ __attribute__((aligned(16))) float KsT[32 * KTP];
  __shared__ float lse_s[2][32], dl_s[2][32];
  __shared__ uint32_t rk_s[2][32];

  const petr_mha_bwd_args& a = p.a;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int li = lane & 15, kq = lane >> 4;           // 16x16x4 operand / result coordinates
  const int qh = wave >> 1, dh = wave & 1;            // this wave's dQ quadrant
  const float inv_scale = 1.f / a.scale;
  const float sc2 = a.scale * LOG2E;
  // every global address of the tile loop = a wave-uniform 64-bit base + an unsigned 32-bit element offset (scalar tile part
  // + a per-lane part computed once): the f32 MFMA shares the vector issue port, 64-bit vector arithmetic is paid in full
  const int ld_row = t >> 3, ld_c4 = t & 7;
  const unsigned q_rs = (unsigned)a.q_rs, g_rs = (unsigned)a.do_rs, o_rs = (unsigned)a.o_rs, dq_rs = (unsigned)a.dq_rs;
  const unsigned q_off = ld_row * q_rs + 4 * ld_c4, g_off = ld_row * g_rs + 4 * ld_c4, o_off = ld_row * o_rs + 4 * ld_c4;
  const unsigned dq_off = (16 * qh + 4 * kq) * dq_rs + 16 * dh + li;

#ifdef PETR_DIAG_BWD_STAMPS
  uint64_t st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t st_last = __builtin_amdgcn_s_memtime();
  int st_tiles = 0, st_visits = 0;
#endif
  const int lw = xcd_remap(blockIdx.x, p.G);          // consecutive logical ids share an XCD (and with them Q / dO / O in L2)
  const long R = (long)p.rem * p.nqt;
  int u0 = (int)((long)lw * R / p.G);
  const int u1 = (int)((long)(lw + 1) * R / p.G);
  int round = 0;

  for (;;) {
    int pair, qb, qe;
    if (u0 < u1) {
      const int pr = u0 / p.nqt;
      qb = u0 - pr * p.nqt;
      qe = min(p.nqt, qb + (u1 - u0));
      pair = p.full_rounds * p.G + pr;
      u0 += qe - qb;
    } else if (round < p.full_rounds) {
      pair = round * p.G + lw;
      qb = 0;
      qe = p.nqt;
      ++round;
    } else {
      break;
    }
    const int kb = pair % p.nkb;
    const int bh = pair / p.nkb;
    const int b = bh / a.H, hd = bh - b * a.H;
    const int key0 = kb * 128 + wave * 32;
    const int key = key0 + c;
    const bool key_ok = key < a.L;
    const int key_ld = key_ok ? key : a.L - 1;

    const float* qp = a.q + (long)b * a.q_bs + (long)hd * a.q_hs;
    const float* gp = a.d_o + (long)b * a.do_bs + (long)hd * a.do_hs;
    const float* op = a.o + (long)b * a.o_bs + (long)hd * a.o_hs;
    const float* lp = a.lse + (long)bh * a.Q;
    float* dqp = a.dq + (long)b * a.dq_bs + (long)hd * a.dq_hs;
    const uint32_t* bits_p = DROP == 2 ? p.drop_bits + (long)bh * p.nqt32 * p.lpad + min(key0, p.lpad - 32) : nullptr;
    const unsigned bits_off = (unsigned)petr_bits_slot(c);

    auto gload = [&](TileRegs& r, int qt) {
      unsigned qo = q_off, go = g_off, oo = o_off, lo = (unsigned)(t & 31);
      if (qt * 32 + 32 > a.Q) {   // wave-uniform: ragged last tile, rows beyond Q re-read row Q-1
        const unsigned rr = (unsigned)min(ld_row, a.Q - 1 - qt * 32);
        qo = rr * q_rs + 4 * ld_c4; go = rr * g_rs + 4 * ld_c4; oo = rr * o_rs + 4 * ld_c4;
        lo = (unsigned)min(t & 31, a.Q - 1 - qt * 32);
      }
      const unsigned tq = (unsigned)(qt * 32);
      const float* s = qp + (size_t)(tq * q_rs + qo);
      const float* g = gp + (size_t)(tq * g_rs + go);
      const float* o = op + (size_t)(tq * o_rs + oo);
      if (VEC) {
        r.q = *reinterpret_cast<const float4*>(s);
        r.g = *reinterpret_cast<const float4*>(g);
        r.o = *reinterpret_cast<const float4*>(o);
      } else {
        r.q = make_float4(s[0], s[1], s[2], s[3]);
        r.g = make_float4(g[0], g[1], g[2], g[3]);
        r.o = make_float4(o[0], o[1], o[2], o[3]);
      }
      r.lse = lp[(size_t)(tq + lo)];
      r.bits = 0u;
      if (DROP == 2) r.bits = bits_p[(size_t)((unsigned)qt * (unsigned)p.lpad + bits_off)];
    };
    uint32_t dbits0 = 0u, dbits1 = 0u;     // this lane's mask word of the tile in image 0 / 1
    auto stage = [&](const TileRegs& r, int qt, auto buf_c) {
      constexpr int buf = decltype(buf_c)::value;
      if (buf) dbits1 = r.bits; else dbits0 = r.bits;
      // rows beyond Q hold a copy of row Q-1 (gload clamps): their -LSE is -inf, so p = ds = 0 whatever the images hold
      const int row = t >> 3, c4 = t & 7;
      float* d = Qs[buf] + row * P33 + 4 * c4;
      d[0] = r.q.x; d[1] = r.q.y; d[2] = r.q.z; d[3] = r.q.w;
      float* e = dOs[buf] + row * P33 + 4 * c4;
      e[0] = r.g.x; e[1] = r.g.y; e[2] = r.g.z; e[3] = r.g.w;
      // delta of the row = sum over its 8 lanes: two quad permutes and a half-row mirror (DPP, no LDS round trips)
      float dl = (r.g.x * r.o.x + r.g.y * r.o.y) + (r.g.z * r.o.z + r.g.w * r.o.w);
      dl += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(dl), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
      dl += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(dl), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
      dl += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(dl), 0x141, 0xF, 0xF, true));   // row_half_mirror
      if (c4 == 0) dl_s[buf][row] = -dl;
      if (t < 32) {
        const bool rok = qt * 32 + t < a.Q;
        lse_s[buf][t] = rok ? -r.lse * inv_scale : -INFINITY;   // rows beyond Q: p = 0
        if (DROP == 1) rk_s[buf][t] = drop_row_key(p.drop, (uint32_t)(bh * a.Q + min(qt * 32 + t, a.Q - 1)));
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // the first two query tiles are requested before anything else of the visit
    TileRegs ra, rb;
    gload(ra, qb);
    if (qb + 1 < qe) gload(rb, qb + 1);

    // ---- this wave's K / V rows: lane (c,h) holds row key, columns 16h..16h+15 ----
    const float* kp = a.k + (long)b * a.k_bs + (long)hd * a.k_hs;
    float kf[16], vf[16];
    {
      const float* vp = a.v + (long)b * a.v_bs + (long)hd * a.v_hs;
      const float* ks = kp + (long)key_ld * a.k_rs + 16 * h;
      const float* vs = vp + (long)key_ld * a.v_rs + 16 * h;
      if (VEC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 kx = *reinterpret_cast<const float4*>(ks + 4 * i), vx = *reinterpret_cast<const float4*>(vs + 4 * i);
          kf[4 * i] = key_ok ? kx.x : 0.f; kf[4 * i + 1] = key_ok ? kx.y : 0.f; kf[4 * i + 2] = key_ok ? kx.z : 0.f; kf[4 * i + 3] = key_ok ? kx.w : 0.f;
          vf[4 * i] = key_ok ? vx.x : 0.f; vf[4 * i + 1] = key_ok ? vx.y : 0.f; vf[4 * i + 2] = key_ok ? vx.z : 0.f; vf[4 * i + 3] = key_ok ? vx.w : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float kx = ks[i], vx = vs[i];
          kf[i] = key_ok ? kx : 0.f;
          vf[i] = key_ok ? vx : 0.f;
        }
      }
    }
    // K transposed in LDS, [d][key of the block], as the B operand of the dQ quadrants
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + 256 * i;
      const int kr = idx >> 3, c4 = idx & 7;
      const int kg = kb * 128 + kr;
      const float* s = kp + (long)min(kg, a.L - 1) * a.k_rs + 4 * c4;
      const bool ok = kg < a.L;
      float4 v;
      if (VEC) v = *reinterpret_cast<const float4*>(s);
      else v = make_float4(s[0], s[1], s[2], s[3]);
      KsT[(4 * c4 + 0) * KTP + kr] = ok ? v.x : 0.f;
      KsT[(4 * c4 + 1) * KTP + kr] = ok ? v.y : 0.f;
      KsT[(4 * c4 + 2) * KTP + kr] = ok ? v.z : 0.f;
      KsT[(4 * c4 + 3) * KTP + kr] = ok ? v.w : 0.f;
    }
    float key_bias = key_ok ? 0.f : -INFINITY;
    if (HAS_MASK && key_ok && a.kpm[(long)b * a.L + key]) key_bias = -INFINITY;

    f32x16 dKt, dVt;
#pragma unroll
    for (int r = 0; r < 16; ++r) dKt[r] = dVt[r] = 0.f;

    stage(ra, qb, I0{});
    if (qb + 1 < qe) stage(rb, qb + 1, I1{});
    if (qb + 2 < qe) gload(ra, qb + 2);
    __syncthreads();
    STAMP(0);      // visit prologue
#ifdef PETR_DIAG_BWD_STAMPS
    ++st_visits;
    st_tiles += qe - qb;
#endif

    // one query tile on image CUR (compile-time: every LDS address of the tile is a base + an immediate offset)
    auto tile = [&](auto cur_c, const int qt) {
      constexpr int cur = decltype(cur_c)::value;
      const float* Qc = Qs[cur];
      const float* Gc = dOs[cur];
      f32x16 S, dP;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qr = mfma32_row(r, h);
        S[r] = lse_s[cur][qr] + key_bias;
        dP[r] = DROP ? 0.f : dl_s[cur][qr];
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(Qc[c * P33 + 16 * h + s], kf[s], S, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_32x32x2f32(Gc[c * P33 + 16 * h + s], vf[s], dP, 0, 0, 0);
      }
      STAMP(1);
      const uint32_t dbits = cur ? dbits1 : dbits0;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        S[r] = __builtin_amdgcn_exp2f(S[r] * sc2);   // p
        if (DROP) {
          const int qr = mfma32_row(r, h);
          float m;
          if (DROP == 2)      // bit qr of the key's mask word
            m = __uint_as_float((uint32_t)__builtin_amdgcn_sbfe(dbits >> (4 * h), (r & 3) + 8 * (r >> 2), 1) & __float_as_uint(p.drop.scale));
          else
            m = drop_keep(rk_s[cur][qr], (uint32_t)key, p.drop.thr) ? p.drop.scale : 0.f;
          dP[r] = S[r] * (dP[r] * m + dl_s[cur][qr]);     // ds = p * (m * dO.V - delta)
          S[r] *= m;                                      // dropped probability: B operand of dV
        } else {
          dP[r] = S[r] * dP[r];
        }
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int qr = mfma32_row(s, h);
        dVt = __builtin_amdgcn_mfma_f32_32x32x2f32(Gc[qr * P33 + c], S[s], dVt, 0, 0, 0);
        dKt = __builtin_amdgcn_mfma_f32_32x32x2f32(Qc[qr * P33 + c], dP[s], dKt, 0, 0, 0);
      }
      {
        float* dsw = dSs[cur][wave];
#pragma unroll
        for (int r = 0; r < 16; ++r) dsw[mfma32_row(r, h) * P34 + c] = dP[r];
      }
      STAMP(2);
      __syncthreads();   // every wave is done with image `cur` of Q / dO; the four ds tiles are complete
      STAMP(3);
      // dQ quadrant (16 queries x 16 d) over the block's 128 keys: operands of key group kw + 1 are requested before the
      // products of group kw
      const float* ap = &dSs[cur][0][0] + (16 * qh + li) * P34 + 2 * kq;
      const float* bp = KsT + (16 * dh + li) * KTP + 2 * kq;
      float2 av[2][4], bv[2][4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        av[0][m] = *reinterpret_cast<const float2*>(ap + 8 * m);
        bv[0][m] = *reinterpret_cast<const float2*>(bp + 8 * m);
      }
      __builtin_amdgcn_sched_barrier(0);     // (hipcc otherwise sinks every operand read to just in front of its product)
      if (qt + 2 < qe) {
        stage(ra, qt + 2, cur_c);
        if (qt + 3 < qe) gload(ra, qt + 3);
      }
      STAMP(4);    // first dQ operands requested, tile + 2 staged, tile + 3 requested
      f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        if (kw < 3) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            av[(kw + 1) & 1][m] = *reinterpret_cast<const float2*>(ap + (kw + 1) * (32 * P34) + 8 * m);
            bv[(kw + 1) & 1][m] = *reinterpret_cast<const float2*>(bp + 32 * (kw + 1) + 8 * m);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          dq0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kw & 1][m].x, bv[kw & 1][m].x, dq0, 0, 0, 0);
          dq1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kw & 1][m].y, bv[kw & 1][m].y, dq1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        const unsigned tq = (unsigned)(qt * 32) * dq_rs + dq_off;
        if (qt * 32 + 32 <= a.Q) {     // wave-uniform
#pragma unroll
          for (int r = 0; r < 4; ++r) atomicAdd(dqp + (size_t)(tq + r * dq_rs), (dq0[r] + dq1[r]) * a.scale);
        } else {
          const int row0 = qt * 32 + 16 * qh + 4 * kq;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (row0 + r < a.Q) atomicAdd(dqp + (size_t)(tq + r * dq_rs), (dq0[r] + dq1[r]) * a.scale);
        }
      }
      STAMP(5);    // dQ quadrant products + atomics issued
    };
    for (int qt = qb; qt < qe; qt += 2) {
      tile(I0{}, qt);
      if (qt + 1 < qe) tile(I1{}, qt + 1);
    }

    // ---- dK / dV of this visit: transpose the accumulators through LDS (the ds images are free), then row-major adds ----
    __syncthreads();
    {
      float* tk = dSs[0][wave];
      float* tv = dSs[1][wave];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        tk[c * P33 + mfma32_row(r, h)] = dKt[r] * a.scale;
        tv[c * P33 + mfma32_row(r, h)] = dVt[r];
      }
      __syncthreads();
      float* dk = a.dk + (long)b * a.dk_bs + (long)hd * a.dk_hs;
      float* dv = a.dv + (long)b * a.dv_bs + (long)hd * a.dv_hs;
      const bool use_atomic = qe - qb < p.nqt;       // a piece of a pair: other workgroups add to the same rows
      float oldk[16], oldv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int idx = j * 64 + lane;
        const int kg = min(key0 + (idx >> 5), a.L - 1);
        oldk[j] = use_atomic ? 0.f : dk[(long)kg * a.dk_rs + (idx & 31)];
        oldv[j] = use_atomic ? 0.f : dv[(long)kg * a.dv_rs + (idx & 31)];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int idx = j * 64 + lane;
        const int kr = idx >> 5, d = idx & 31;
        const int kg = key0 + kr;
        if (kg < a.L) {
          const float gk = tk[kr * P33 + d], gv = tv[kr * P33 + d];
          float* pk = dk + (long)kg * a.dk_rs + d;
          float* pv = dv + (long)kg * a.dv_rs + d;
          if (use_atomic) {
#ifndef PETR_DIAG_BWD_NOFLUSH      // (timing-only diagnostic build: what the partial pieces' dK / dV atomics cost)
            atomicAdd(pk, gk);
            atomicAdd(pv, gv);
#endif
          } else {
            *pk = oldk[j] + gk;
            *pv = oldv[j] + gv;
          }
        }
      }
    }
    __syncthreads();     // the images are rewritten by the next visit
    STAMP(7);      // dK / dV flush
  }
#ifdef PETR_DIAG_BWD_STAMPS
  if (t == 0) {
    unsigned long long* out = reinterpret_cast<unsigned long long*>(a.ws);
    for (int i = 0; i < 8; ++i) atomicAdd(out + i, (unsigned long long)st_acc[i]);
    atomicAdd(out + 8, (unsigned long long)st_tiles);
    atomicAdd(out + 9, (unsigned long long)st_visits);
  }
#endif
}

}  // namespace

extern "C" size_t petr_mha_bwd_workspace_bytes(int B, int H, int Q, int L) {
  (void)L;
  return (size_t)B * H * Q * sizeof(float);
}

extern "C" int petr_mha_bwd(const petr_mha_bwd_args* ap, void* stream) {
  PETR_CHECK(ap && ap->q && ap->k && ap->v && ap->o && ap->d_o && ap->lse && ap->dq && ap->dk && ap->dv, PETR_ERR_INVALID,
             "mha_bwd: null pointer");
  PETR_CHECK(ap->B > 0 && ap->H > 0 && ap->Q > 0 && ap->L > 0, PETR_ERR_INVALID, "mha_bwd: bad shape");
  const size_t need = petr_mha_bwd_workspace_bytes(ap->B, ap->H, ap->Q, ap->L);
  PETR_CHECK(ap->ws && ap->ws_bytes >= need, PETR_ERR_WORKSPACE, "mha_bwd: workspace %zu < %zu bytes", ap->ws_bytes, need);
  MhaBwdSkParams k;
  k.a = *ap;
  const petr_mha_bwd_args& a = k.a;
  k.nkb = (int)cdiv(a.L, 128);
  k.nqt = (int)cdiv(a.Q, 32);
  const long P = (long)k.nkb * a.B * a.H;
  PETR_CHECK(P * k.nqt < (1L << 31), PETR_ERR_UNSUPPORTED, "mha_bwd: too many (key block, head, query tile) units");
  const int cus = petr_num_cus();
  PETR_CHECK(cus > 0, PETR_ERR_LAUNCH, "mha_bwd: cannot read the device's CU count");
  // Resident workgroups per CU: two (the kernel's LDS / register budget) while every workgroup still gets >= 8 query tiles;
  // below that (the 900 x 900 self-attention: 1 856 tiles) a visit's fixed cost - K / V rows in, dK / dV out through float
  // atomics, ~5 tiles' worth - outweighs the second wave per SIMD and one workgroup per CU is faster (measured, same box:
  // 38.7 against 43.3 us at 900 x 900, 124.8 against 115.9 us at c5).  Everything else about the schedule follows from
  // (P, nqt, CU count): see the kernel.
  const int slots = P * k.nqt >= 16L * cus ? 2 : 1;
  long G = (long)cus * slots;
  if (G > P * k.nqt) G = P * k.nqt;
  k.G = (int)G;
  k.full_rounds = (int)(P / G);
  k.rem = (int)(P % G);
  auto rows_ok = [](const float* p, long bs, long hs, long rs) { return aligned16(p) && !(bs & 3) && !(hs & 3) && !(rs & 3); };
  const bool vec = rows_ok(a.q, a.q_bs, a.q_hs, a.q_rs) && rows_ok(a.d_o, a.do_bs, a.do_hs, a.do_rs) && rows_ok(a.o, a.o_bs, a.o_hs, a.o_rs) &&
                   rows_ok(a.k, a.k_bs, a.k_hs, a.k_rs) && rows_ok(a.v, a.v_bs, a.v_hs, a.v_rs);
  PETR_CHECK(a.drop.p >= 0.f && a.drop.p < 1.f, PETR_ERR_INVALID, "mha_bwd: dropout p=%g outside [0,1)", (double)a.drop.p);
  PETR_CHECK((long)a.B * a.H * a.Q < (1L << 32), PETR_ERR_UNSUPPORTED, "mha_bwd: dropout row index needs B*H*Q < 2^32");
  k.drop = make_drop(a.drop);
  k.drop_bits = a.drop.p > 0.f ? a.drop_bits : nullptr;
  k.nqt32 = k.nqt;
  k.lpad = 32 * (int)cdiv(a.L, 32);
  PETR_CHECK((long)a.Q * a.q_rs < (1L << 31) && (long)a.Q * a.do_rs < (1L << 31) && (long)a.Q * a.o_rs < (1L << 31) &&
                 (long)a.Q * a.dq_rs < (1L << 31) && (long)k.nqt * k.lpad < (1L << 31),
             PETR_ERR_UNSUPPORTED, "mha_bwd: one (batch, head) slice of q / d_o / o / dq must span < 2^31 elements");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t ev0, ev1;   // null unless bench.py's profiler is on: then they carry this dispatch's begin/end
  petr_prof_claim(PETR_PROF_MHA_BWD + 16 * (a.L > a.Q ? 1 : 0), &ev0, &ev1);
  auto launch = [&](auto kern) { hipExtLaunchKernelGGL(kern, dim3((unsigned)k.G), dim3(256), 0, s, ev0, ev1, 0, k); };
  const int dmode = k.drop.thr ? (k.drop_bits ? 2 : 1) : 0;
#define PETR_SK_CASE(M, V)                                                                            \
  (dmode == 0 ? launch(mha_bwd_sk_kernel<M, V, 0>) : dmode == 1 ? launch(mha_bwd_sk_kernel<M, V, 1>) \
                                                                : launch(mha_bwd_sk_kernel<M, V, 2>))
  if (a.kpm) { if (vec) PETR_SK_CASE(true, true); else PETR_SK_CASE(true, false); }
  else { if (vec) PETR_SK_CASE(false, true); else PETR_SK_CASE(false, false); }
#undef PETR_SK_CASE
  PETR_LAUNCH_CHECK("mha_bwd");
  return PETR_OK;
}

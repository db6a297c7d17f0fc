// Multi-head attention forward for the PETR decoder (head_dim 32, fp32, flash-style).
//
// Replaces the bmm -> softmax -> bmm inside torch.nn.MultiheadAttention reached from
// PETRMultiheadAttention.forward (reference models/utils/petr_transformer.py:357-362): 900 object
// queries against L = N*H*W key/value tokens (4 224 ... 24 000), and the 900x900 self-attention.
// Scores are never written to memory.
//
// Mapping (one wave = 32 queries, workgroup = 4 waves = 128 queries, KV tile = 64 keys in LDS):
//   S^T[key][q] = K . Q^T   : A = K tile (row = key), B = Q^T  -> v_mfma_f32_32x32x2_f32 x16
//     ("swapped" product: the query lands on the LANE, the keys on the 16 accumulator registers,
//      so the softmax row reduction is 15 in-lane max/add + ONE v_permlane32_swap; no LDS, no shuffles)
//   O^T[d][q]  += V^T . P^T : the P accumulator registers are used AS the B operand of the second
//     product (k-order permuted identically on the V side: step s multiplies key mfma32_row(s,h)),
//     so P never leaves the register file.
// LDS images: K transposed [d][key] with pitch 65 (transposing ds_write_b32 of a float4 hit 32
// distinct banks; fragment reads are 32 consecutive floats), V natural [key][32].
// K/V tiles are register-prefetched one tile ahead (cdna guide T14).
// B=1 gives only 8 heads x 8 query blocks = 64 workgroups, so L is split over workgroups
// (flash-decoding): each split writes (O, m, l) partials and a second kernel merges them.
// Workgroup ids are XCD-remapped so that the query blocks sharing one (head, L-split) K/V slice
// run on the same XCD and share its L2.
#include "common.h"

namespace {

constexpr int KV_TILE = 64;
constexpr int KT_PITCH = 65;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float LAZY = 8.f;       // log2 of the slack allowed between a row's true and reference maximum
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct MhaFwdParams {
  petr_mha_fwd_args a;
  int nqb, n_split, n_sub;
  float* o_part;   // [n_split][B*H][Q][32]
  float* ml_part;  // [n_split][B*H][Q][2]
  int q_vec, kv_vec;
  int n_tiles, n_tickets;
  DropDev drop;    // dropout of the probabilities (thr == 0: off)
  int* sched;      // [B*H*nqb] tile tickets (dynamic mode), zero on entry, zero again on exit
  uint32_t* drop_bits;   // out (optional): the key-major packed dropout mask (petr_dropout_bits layout) for the backward
  int nqt32, lpad;       // its dimensions: ceil(Q/32) query tiles, 32 * ceil(L/32) keys per tile
  int pair;              // bf16 kernel: heads 2j / 2j+1 of one key range are neighbours in the workgroup order
};

// The keep decisions of one 32 query x 32 key block leave the kernel as they fall out of the comparisons: the ballot of
// accumulator register r (an SGPR pair) holds, in its low / high word, the 32 queries' bits for key mfma32_row(r, 0) /
// (r, 1) of the block.  They are written with SCALAR stores - no vector instruction is spent on them (collecting them in
// a vector register with v_writelane cost the forward 3-5 %) - as 16 x 2 words per block: word 2 r + half of block
// (query tile, key block) = key (r & 3) + 8 (r >> 2) + 4 half, i.e. key c sits at petr_bits_slot(c) (petr_hip.h).
// Scalar stores go through the scalar data cache: every wave writes it back (s_dcache_wb) before it ends.
template <int R>
__device__ __forceinline__ void bits_sstore(unsigned long long base, unsigned long long bal) {
  asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(bal), "s"(base), "n"(8 * R) : "memory");
}
__device__ __forceinline__ void bits_store_block(uint32_t* block, const unsigned long long (&bal)[16]) {
  const unsigned long long a = (unsigned long long)block;
  // (readfirstlane returns a signed int: widen through uint32_t, or a low half with bit 31 set smears over the high half)
  const unsigned long long base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) |
                                  (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  bits_sstore<0>(base, bal[0]);   bits_sstore<1>(base, bal[1]);   bits_sstore<2>(base, bal[2]);    bits_sstore<3>(base, bal[3]);
  bits_sstore<4>(base, bal[4]);   bits_sstore<5>(base, bal[5]);   bits_sstore<6>(base, bal[6]);    bits_sstore<7>(base, bal[7]);
  bits_sstore<8>(base, bal[8]);   bits_sstore<9>(base, bal[9]);   bits_sstore<10>(base, bal[10]);  bits_sstore<11>(base, bal[11]);
  bits_sstore<12>(base, bal[12]); bits_sstore<13>(base, bal[13]); bits_sstore<14>(base, bal[14]);  bits_sstore<15>(base, bal[15]);
}

#ifdef PETR_DIAG_CLOCK   // diagnostic build only (scripts/diag_clock.cpp): in-kernel clock of the main loop
__device__ unsigned long long g_diag[4 * 4 * 4096];   // per wave: cycles, realtime ticks, tiles | active<<32, hw_id | xcc_id<<32
#endif

// 8 waves: wave = (key half kh, query group qg).  The two waves of a query group take the two 32-key halves of
// every 64-key tile and keep their own running (m, l, O); they are merged through LDS once, after the loop.
// Why 8 and not 4: with ONE resident wave per query group a SIMD held 2 waves, and measured in-kernel
// (scripts/diag_clock.cpp) a wave alone reaches only 66 % of the MFMA issue rate (its softmax and LDS waits are
// exposed), so the SIMD idled whenever its two waves were in the same phase and in the tail; 4 half-size
// instruction streams per SIMD keep the matrix pipe fed.
//
// Tile scheduling.  The n_split workgroups of one (batch, head, query block) are WORKERS over that group's
// K/V tiles; each keeps one running (m, l, O) and writes one partial, so any assignment of tiles to workers is
// valid.  DYN = false: worker s takes the contiguous range [s*per, (s+1)*per).  DYN = true: worker s starts
// with tile s and then draws tickets from a per-group counter (one atomicAdd per tile, fetched one tile ahead
// of its use).  Measured reason (diag_clock): the SIMD arbiter favours the older of the two workgroups sharing a
// CU, which then runs at 7.1k cycles/tile while the younger gets 10.3k, so with equal static ranges the younger
// one finishes the last third of its range alone at 2/3 of the MFMA rate; with tickets the fast worker simply
// takes more tiles and both drain together (the same mechanism absorbs the ragged last range and the mostly
// padded last query block).  The worker whose ticket is the group's last resets the counter for the next launch.
template <bool HAS_MASK, bool VEC, bool DYN, bool DROP>
__global__ __launch_bounds__(512, 2) void mha_fwd_kernel(const MhaFwdParams p) {
  constexpr int MERGE_PITCH = 17;
  constexpr int IMG = 32 * KT_PITCH + KV_TILE * 32;   // one K^T + V image pair, floats
  __shared__ __attribute__((aligned(16))) float smem[2 * IMG];   // two image pairs; reused by the final merge
  __shared__ float bias_s[2][KV_TILE];
  __shared__ int sched_s[2];
  static_assert((32 * KT_PITCH) % 4 == 0 && IMG % 4 == 0, "V images must stay 16-byte aligned");
  static_assert(2 * IMG >= 4 * 64 * MERGE_PITCH + 2 * 4 * 32, "merge buffer larger than the tile images");

  const petr_mha_fwd_args& a = p.a;
  const int total = p.nqb * a.B * a.H * p.n_split;
  const int w = xcd_remap(blockIdx.x, total);
  const int qb = w % p.nqb;
  const int rest = w / p.nqb;
  const int split = rest % p.n_split;
  const int bh = rest / p.n_split;
  const int b = bh / a.H, hd = bh - b * a.H;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int qg = wave & 3, kh = wave >> 2;
  const int h = lane >> 5, c = lane & 31;
  const int q_row = qb * 128 + qg * 32 + c;
  const int q_ld = min(q_row, a.Q - 1);
  const bool wave_active = qb * 128 + qg * 32 < a.Q;
  // attention-probability dropout (nn.MultiheadAttention's attn_drop): element (row, col) = ((b*H+h)*Q + q, key);
  // the row sum l keeps the UNdropped probabilities (softmax first, dropout after), the 1/(1-p) is applied once
  // to the finished output
  const uint32_t drop_rk = DROP ? drop_row_key(p.drop, (uint32_t)(bh * a.Q + q_ld)) : 0u;
  const uint16_t thr16 = (uint16_t)p.drop.thr;        // thr <= 65535 (make_drop)

  // Static ranges are cut at 32-key granularity (132 halves over 8 workers = 16/17 each at c5, not 9,9,...,3
  // tiles): the worker's tiles are 64-key steps from ITS first key and its last tile may be half (the half-tile
  // machinery is the one the ragged end of L needs anyway).  Dynamic mode numbers the tiles of the whole key axis.
  int k_base = 0, k_end = a.L, n_tiles = p.n_tiles;
  if (!DYN) {
    const int base = p.n_sub / p.n_split, rem = p.n_sub - base * p.n_split;
    k_base = 32 * (split * base + min(split, rem));
    k_end = min(a.L, k_base + 32 * (base + (split < rem ? 1 : 0)));
    n_tiles = (k_end - k_base + KV_TILE - 1) / KV_TILE;      // host guarantees >= 1
  }
  int cur = DYN ? split : 0;
  const int my_end = n_tiles;
  int* ticket = DYN ? p.sched + (bh * p.nqb + qb) : nullptr;
  // tickets: tiles s and n_split+s are worker s' first two tiles (the pipeline is two tiles deep), ticket v names
  // tile 2*n_split + v.  Every worker draws until its first miss, so a group draws exactly p.n_tickets tickets and
  // the draw that returns p.n_tickets-1 is the last.
  // The ticket value is only looked at one tile later (publish()): using it right after the atomic would make
  // thread 0's wave sit out the whole L2 round trip in front of its MFMA work.
  // It is issued as inline asm with EXEC narrowed to lane 0 of wave 0: through atomicAdd() hipcc's atomic optimizer
  // consumes the result at once (s_waitcnt vmcnt(0) right behind the atomic, in front of the MFMA work).  The
  // compiler does not count this operation in vmcnt; memory operations return in order, so its own counted waits
  // only become stricter, and publish() waits for everything before it reads the ticket.
  const bool wave0 = __builtin_amdgcn_readfirstlane(t >> 6) == 0;
  int drawn = n_tiles;                    // lane 0 of wave 0: raw ticket for the tile after the next one
  auto draw = [&](bool on) {
    const int m = __builtin_amdgcn_readfirstlane(on ? 1 : 0);
    unsigned long long saved;
    const int one = 1;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b32 exec_lo, %[m]\n\t"
        "s_mov_b32 exec_hi, 0\n\t"
        "global_atomic_add %[r], %[a], %[o], off sc0\n\t"
        "s_mov_b64 exec, %[sv]"
        : [r] "+v"(drawn), [sv] "=&s"(saved)
        : [m] "s"(m), [a] "v"(ticket), [o] "v"(one)
        : "memory");
  };
  auto publish = [&](int slot) {
    if (wave0) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(drawn) : : "memory");   // "+v": no read of the ticket above this
      if (t == 0) {
        if (drawn == p.n_tickets - 1) atomicExch(ticket, 0);
        // a counter that was not zero on entry (contract violation) can only end the worker early, never index
        // outside [0, n_tiles)
        sched_s[slot] = (unsigned)drawn < (unsigned)n_tiles ? 2 * p.n_split + drawn : n_tiles;
      }
    }
  };

  // ---- Q fragment: lane (c,h) holds Q[q][16h .. 16h+15], pre-scaled by scale*log2(e) ----
  float qf[16];
  {
    const float* qp = a.q + (long)b * a.q_bs + (long)hd * a.q_hs + (long)q_ld * a.q_rs + 16 * h;
    const float sc = a.scale * LOG2E;
    if (VEC) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = reinterpret_cast<const float4*>(qp)[i];
        qf[4 * i] = v.x * sc; qf[4 * i + 1] = v.y * sc; qf[4 * i + 2] = v.z * sc; qf[4 * i + 3] = v.w * sc;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) qf[i] = qp[i] * sc;
    }
  }

  const float* kp = a.k + (long)b * a.k_bs + (long)hd * a.k_hs;
  const float* vp = a.v + (long)b * a.v_bs + (long)hd * a.v_hs;
  const uint8_t* mp = HAS_MASK ? a.kpm + (long)b * a.L : nullptr;

  float4 kreg, vreg;
  uint8_t mreg = 0;
  const int ld_key = t >> 3, ld_c4 = t & 7;
  // f32 MFMA issues on the same lanes as the vector ALU (SQ_VALU_MFMA_COEXEC_CYCLES = 0 for this kernel; removing
  // the softmax / the staging from a diagnostic build shortened it by exactly their share), so every VALU
  // instruction in this loop is paid in full: tile addresses are a wave-uniform base + a per-lane 32-bit offset
  // that is computed once, and nothing is selected or zero-filled per element.
  const int koff = ld_key * (int)a.k_rs + 4 * ld_c4, voff = ld_key * (int)a.v_rs + 4 * ld_c4;
  // Loads are unconditional, rows beyond L re-read row L-1 (any finite data will do: their scores get a -inf
  // bias, so their probabilities are exact zeros).  Nothing is done with a loaded register before lstore(): a
  // use right behind a load would make hipcc wait for it on the spot.
  auto gload = [&](int k0) {
    int ko = koff, vo = voff;
    if (k0 + KV_TILE > a.L) {   // wave-uniform: ragged last tile
      const int kk = min(ld_key, a.L - 1 - k0);
      ko = kk * (int)a.k_rs + 4 * ld_c4;
      vo = kk * (int)a.v_rs + 4 * ld_c4;
    }
    const float* ks = kp + (long)k0 * a.k_rs + ko;
    const float* vs = vp + (long)k0 * a.v_rs + vo;
    if (VEC) {
      kreg = *reinterpret_cast<const float4*>(ks);
      vreg = *reinterpret_cast<const float4*>(vs);
    } else {
      kreg = make_float4(ks[0], ks[1], ks[2], ks[3]);
      vreg = make_float4(vs[0], vs[1], vs[2], vs[3]);
    }
    if (HAS_MASK) mreg = mp[min(k0 + (t & (KV_TILE - 1)), a.L - 1)];
  };
  auto lstore = [&](int k0, int buf) {
    float* Kt = smem + buf * IMG;
    float* Vs = Kt + 32 * KT_PITCH;
    float* d = Kt + (4 * ld_c4) * KT_PITCH + ld_key;
    d[0] = kreg.x;
    d[KT_PITCH] = kreg.y;
    d[2 * KT_PITCH] = kreg.z;
    d[3 * KT_PITCH] = kreg.w;
    *reinterpret_cast<float4*>(Vs + ld_key * 32 + 4 * ld_c4) = vreg;
    if ((HAS_MASK || k0 + KV_TILE > k_end) && t < KV_TILE) {
      bool dead = k0 + t >= k_end;
      if (HAS_MASK) dead = dead || mreg != 0;
      bias_s[buf][t] = dead ? -INFINITY : 0.f;
    }
  };

  f32x16 O;
#pragma unroll
  for (int r = 0; r < 16; ++r) O[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // Two-tile pipeline with ONE barrier per tile: tile `cur` is multiplied out of image it&1 while tile `nxt` moves
  // from registers into the other image and tile `nn` is on its way from memory into the registers.
  int nxt = DYN ? p.n_split + split : cur + 1;
  if (nxt >= my_end) nxt = n_tiles;
  gload(k_base + cur * KV_TILE);
  if (DYN) draw(wave0);
  lstore(k_base + cur * KV_TILE, 0);
  if (nxt < n_tiles) gload(k_base + nxt * KV_TILE);
#ifdef PETR_DIAG_CLOCK
  const unsigned long long dg_t0 = __builtin_amdgcn_s_memtime(), dg_r0 = __builtin_amdgcn_s_memrealtime();
  int dg_tiles = 0;
#endif
  for (int it = 0;; ++it) {
    const int k0 = k_base + cur * KV_TILE;
    const int buf = it & 1;
    if (DYN) publish(buf);
#ifndef PETR_DIAG_NO_BARRIER
    __syncthreads();   // image `buf` is complete; every wave is done reading the other one
#endif
    int nn = n_tiles;
    if (nxt < n_tiles) {
      if (DYN) nn = __builtin_amdgcn_readfirstlane(sched_s[buf]);
      else nn = nxt + 1 < my_end ? nxt + 1 : n_tiles;
#ifndef PETR_DIAG_NO_STAGE
      lstore(k_base + nxt * KV_TILE, buf ^ 1);
      if (nn < n_tiles) gload(k_base + nn * KV_TILE);
#endif
    }
    if (DYN) draw(wave0 && nn < n_tiles);
    const float* Kt = smem + buf * IMG;
    const float* Vs = Kt + 32 * KT_PITCH;
#ifdef PETR_DIAG_CLOCK
    ++dg_tiles;
#endif
    const bool use_bias = HAS_MASK || (k0 + KV_TILE > k_end);
    // wave-uniform skips: a wave whose 32 queries are all padding, or whose key half lies beyond L,
    // only stages K/V
    if (wave_active && k0 + kh * 32 < k_end) {
    // all 16 K fragments are requested before the first MFMA so that one LDS latency, not sixteen,
    // sits in front of the dependent MFMA chain (the compiler otherwise emits read->wait->2 MFMA)
    float kfr[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) kfr[s] = Kt[(16 * h + s) * KT_PITCH + kh * 32 + c];
    __builtin_amdgcn_sched_barrier(0);
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) S = __builtin_amdgcn_mfma_f32_32x32x2f32(kfr[s], qf[s], S, 0, 0, 0);
    // V fragments: requested now, consumed after the softmax (their latency hides under its VALU work)
    float vfr[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) vfr[s] = Vs[(kh * 32 + mfma32_row(s, h)) * 32 + c];
    if (use_bias) {
#pragma unroll
      for (int r = 0; r < 16; ++r) S[r] += bias_s[buf][kh * 32 + mfma32_row(r, h)];
    }
#ifndef PETR_DIAG_NO_SOFTMAX
    float mx = S[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, S[r]);
    mx = xhalf_max(mx);
    // Lazy reference maximum: m_run only moves when some row's maximum has outgrown it by more than 2^LAZY (then
    // exp2(S - m_run) <= 2^LAZY, far from overflow).  (m, l, O) stay a consistent triple, so O / l and the LSE are
    // the same numbers; what is saved is the rescale of O and its exp2 in almost every step.
    if (__builtin_amdgcn_ballot_w64(mx > m_run + LAZY) != 0) {   // wave-uniform
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - ((m_new == -INFINITY) ? 0.f : m_new));
      const f32x2 al = {alpha, alpha};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x2 o2 = {O[2 * i], O[2 * i + 1]};
        o2 *= al;
        O[2 * i] = o2.x; O[2 * i + 1] = o2.y;
      }
      l_run *= alpha;
      m_run = m_new;
    }
    {
      const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
      const f32x2 mm = {m_use, m_use};
      f32x2 acc = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i) {   // packed (two floats per instruction) subtract and row sum
        f32x2 e = {S[2 * i], S[2 * i + 1]};
        e -= mm;
        e.x = __builtin_amdgcn_exp2f(e.x);
        e.y = __builtin_amdgcn_exp2f(e.y);
        S[2 * i] = e.x; S[2 * i + 1] = e.y;
        acc += e;
      }
      l_run += xhalf_sum(acc.x + acc.y);
    }
    if (DROP) {
      // accumulator registers (2i, 2i+1) hold the keys (2j, 2j+1) of one column pair: one hash per two probabilities
      unsigned long long bal[16];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t hsh = drop_pair_hash(drop_rk, (uint32_t)(k0 + kh * 32 + mfma32_row(2 * i, h)) >> 1);
        const bool k0b = (uint16_t)hsh >= thr16, k1b = (uint16_t)(hsh >> 16) >= thr16;     // SDWA half-word compares: no extraction
        S[2 * i] = k0b ? S[2 * i] : 0.f;
        S[2 * i + 1] = k1b ? S[2 * i + 1] : 0.f;
        bal[2 * i] = __builtin_amdgcn_ballot_w64(k0b);
        bal[2 * i + 1] = __builtin_amdgcn_ballot_w64(k1b);
      }
      if (p.drop_bits)               // wave-uniform: leave the mask for the backward (it then tests bits instead of hashing)
        bits_store_block(p.drop_bits + ((long)bh * p.nqt32 + (qb * 4 + qg)) * p.lpad + (k0 + kh * 32), bal);
    }
#endif
#pragma unroll
    for (int s = 0; s < 16; ++s) O = __builtin_amdgcn_mfma_f32_32x32x2f32(vfr[s], S[s], O, 0, 0, 0);
    }
    if (nxt >= n_tiles) break;
    cur = nxt;
    nxt = nn;
  }

#ifdef PETR_DIAG_CLOCK
  if (lane == 0 && blockIdx.x < 4096 && wave < 4) {
    unsigned long long* dg = g_diag + 4 * (4 * blockIdx.x + wave);
    dg[0] = __builtin_amdgcn_s_memtime() - dg_t0;
    dg[1] = __builtin_amdgcn_s_memrealtime() - dg_r0;
    dg[2] = (unsigned long long)dg_tiles | ((unsigned long long)wave_active << 16) |
            ((unsigned long long)w << 32);
    dg[3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
            ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
  }
#endif

  if (DROP && p.drop_bits) asm volatile("s_dcache_wb" ::: "memory");     // the scalar-stored mask words leave the scalar cache
  // ---- merge the two key halves of each query group (upper wave -> LDS -> lower wave) ----
  __syncthreads();   // every wave is done with the K/V images
  float* mo = smem + (qg * 64 + lane) * MERGE_PITCH;
  float* mml = smem + 4 * 64 * MERGE_PITCH + qg * 64 + 2 * c;   // (m, l) per query, written by the h == 0 half
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) mo[r] = O[r];
    if (h == 0) { mml[0] = m_run; mml[1] = l_run; }
  }
  __syncthreads();
  if (kh == 1 || q_row >= a.Q) return;
  {
    const float m2 = mml[0], l2 = mml[1];
    const float m_new = fmaxf(m_run, m2);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float a1 = __builtin_amdgcn_exp2f(m_run - m_use), a2 = __builtin_amdgcn_exp2f(m2 - m_use);
#pragma unroll
    for (int r = 0; r < 16; ++r) O[r] = O[r] * a1 + mo[r] * a2;
    l_run = l_run * a1 + l2 * a2;
    m_run = m_new;
  }
  if (p.n_split == 1) {
    const float inv = (DROP ? p.drop.scale : 1.f) / l_run;   // fully masked row: 0 * inf = NaN, as torch's softmax of all -inf
    float* op = a.o + (long)b * a.o_bs + (long)hd * a.o_hs + (long)q_row * a.o_rs;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 8 * g + 4 * h;
      op[d0] = O[4 * g] * inv;
      op[d0 + 1] = O[4 * g + 1] * inv;
      op[d0 + 2] = O[4 * g + 2] * inv;
      op[d0 + 3] = O[4 * g + 3] * inv;
    }
    if (a.lse && h == 0) a.lse[(long)bh * a.Q + q_row] = (m_run + log2f(l_run)) * LN2;
  } else {
    const long row = ((long)split * a.B * a.H + bh) * a.Q + q_row;
    float* op = p.o_part + row * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(op + 8 * g + 4 * h) = make_float4(O[4 * g], O[4 * g + 1], O[4 * g + 2], O[4 * g + 3]);
    if (h == 0) {
      p.ml_part[row * 2] = m_run;
      p.ml_part[row * 2 + 1] = l_run;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// bf16 K/V variant (BASELINE configs 3-5: "bf16 I/O, fp32 accumulate/softmax").  K and V are bf16 in memory, Q is
// read as fp32 and rounded to bf16 after the scale*log2(e) pre-multiplication, scores / softmax / O stay fp32.
// Same mapping as the fp32 kernel (S^T = K Q^T so the query sits on the lane; P^T feeds the second product from
// the accumulator registers), on v_mfma_f32_32x32x16_bf16: 4 matrix instructions per 32x32 score block instead
// of 32, so the kernel is bound by the softmax (16 v_exp_f32 per block are ~2x the MFMA cycles), not by MFMA.
//   tile = 128 keys (half a barrier per key compared with 64), wave (qg, kh) takes 32 queries x 64 keys of it
//   K image: natural [key][32] bf16, pitch 40 (80 B): the A fragment of a lane is 8 consecutive d = ONE
//            ds_read_b128, and 16 lanes x 80 B touch 64 distinct banks
//   V image: transposed [d][key] bf16, pitch 132 (66 dwords): a lane's 8 keys for one k-chunk are two 8-byte
//            reads (keys 16j+4h..+3 and +8: exactly the rows its P registers 8j..8j+7 hold); the transposing
//            store packs the same d of two adjacent keys with one v_perm_b32 into one ds_write_b32
//   k-slot order inside an MFMA is irrelevant as long as both operands agree, which is all the code relies on.
constexpr int BT = 128;          // keys per tile
constexpr int BK_PITCH = 40;     // bf16 elements per K row
constexpr int BV_PITCH = 132;    // bf16 elements per V^T row
constexpr int BIMG = BT * BK_PITCH + 32 * BV_PITCH;   // bf16 elements of one image pair
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool HAS_MASK, bool DROP>
__global__ __launch_bounds__(512, 2) void mha_fwd_bf16_kernel(const MhaFwdParams p) {
  constexpr int MERGE_PITCH = 17;
  __shared__ __attribute__((aligned(16))) uint16_t smem16[2 * BIMG];
  __shared__ float bias_s[2][BT];
  static_assert((BT * BK_PITCH) % 8 == 0 && BIMG % 8 == 0, "images must stay 16-byte aligned");
  static_assert(2 * BIMG * 2 >= (4 * 64 * MERGE_PITCH + 2 * 4 * 32) * 4, "merge buffer larger than the tile images");

  const petr_mha_fwd_args& a = p.a;
  const int total = p.nqb * a.B * a.H * p.n_split;
  const int w = xcd_remap(blockIdx.x, total);
  const int qb = w % p.nqb;
  int rest = w / p.nqb;
  // a bf16 head row is 64 bytes: one 128-byte line carries two heads.  Paired, the two heads of a key range sit on the
  // same XCD (xcd_remap keeps neighbours together), so one L2 fetches the line once instead of two L2s once each.
  const int h0 = p.pair ? (rest & 1) : 0;
  if (p.pair) rest >>= 1;
  const int split = rest % p.n_split;
  const int bh = p.pair ? 2 * (rest / p.n_split) + h0 : rest / p.n_split;
  const int b = bh / a.H, hd = bh - b * a.H;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int qg = wave & 3, kh = wave >> 2;
  const int h = lane >> 5, c = lane & 31;
  const int q_row = qb * 128 + qg * 32 + c;
  const int q_ld = min(q_row, a.Q - 1);
  const bool wave_active = qb * 128 + qg * 32 < a.Q;
  const uint32_t drop_rk = DROP ? drop_row_key(p.drop, (uint32_t)(bh * a.Q + q_ld)) : 0u;
  const uint16_t thr16 = (uint16_t)p.drop.thr;        // thr <= 65535 (make_drop)

  // static key ranges at 64-key granularity (one wave's share of a tile)
  const int base = p.n_sub / p.n_split, rem = p.n_sub - base * p.n_split;
  const int k_base = 64 * (split * base + min(split, rem));
  const int k_end = min(a.L, k_base + 64 * (base + (split < rem ? 1 : 0)));
  const int n_tiles = (k_end - k_base + BT - 1) / BT;      // host guarantees >= 1

  // ---- Q fragments (B operand of K Q^T): lane (c, h) holds Q[q][16j + 8h .. +7], j = 0, 1 ----
  bf16x8 qf[2];
  {
    const float* qp = a.q + (long)b * a.q_bs + (long)hd * a.q_hs + (long)q_ld * a.q_rs + 8 * h;
    const float sc = a.scale * LOG2E;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) qf[j][i] = (__bf16)(qp[16 * j + i] * sc);
  }

  const uint16_t* kp = reinterpret_cast<const uint16_t*>(a.k) + (long)b * a.k_bs + (long)hd * a.k_hs;
  const uint16_t* vp = reinterpret_cast<const uint16_t*>(a.v) + (long)b * a.v_bs + (long)hd * a.v_hs;
  const uint8_t* mp = HAS_MASK ? a.kpm + (long)b * a.L : nullptr;

  // Register ring of PF tiles between memory and LDS.  A tile of this kernel is ~1 us of work, about one loaded
  // memory latency: with the fp32 kernel's one-tile prefetch every iteration began by waiting for its loads (each
  // ablation of the arithmetic then "cost" 2-4 % only); the loads of tile T+PF+1 are issued while tile T is computed.
  struct Stage { uint4 k; uint2 v0, v1; uint8_t m; };
  constexpr int PF = 3;
  Stage st[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) { st[i].k = make_uint4(0, 0, 0, 0); st[i].v0 = st[i].v1 = make_uint2(0, 0); st[i].m = 0; }
  const int k_key = t >> 2, k_ch = t & 3;      // K: 16 bytes = 8 d of one key
  const int v_kp = t >> 3, v_d4 = t & 7;       // V: 4 d of the key pair (2 v_kp, 2 v_kp + 1)
  // addresses = wave-uniform 64-bit tile base (scalar ALU) + per-lane unsigned 32-bit BYTE offset, the form that maps to
  // global_load ... v_offset, s[base:base+1]: no 64-bit vector arithmetic per load
  const uint32_t koff = 2u * (uint32_t)(k_key * (int)a.k_rs + 8 * k_ch);
  const uint32_t voff0 = 2u * (uint32_t)((2 * v_kp) * (int)a.v_rs + 4 * v_d4), voff1 = voff0 + 2u * (uint32_t)a.v_rs;
  auto gload = [&](int k0, Stage& g) {
    uint32_t ko = koff, vo0 = voff0, vo1 = voff1;
    if (k0 + BT > a.L) {   // wave-uniform: ragged last tile, rows beyond L re-read row L-1 (they get a -inf bias)
      const int lim = a.L - 1 - k0;
      ko = 2u * (uint32_t)(min(k_key, lim) * (int)a.k_rs + 8 * k_ch);
      vo0 = 2u * (uint32_t)(min(2 * v_kp, lim) * (int)a.v_rs + 4 * v_d4);
      vo1 = 2u * (uint32_t)(min(2 * v_kp + 1, lim) * (int)a.v_rs + 4 * v_d4);
    }
    const char* kb = reinterpret_cast<const char*>(kp + (long)k0 * a.k_rs);
    const char* vb = reinterpret_cast<const char*>(vp + (long)k0 * a.v_rs);
    g.k = *reinterpret_cast<const uint4*>(kb + ko);
    g.v0 = *reinterpret_cast<const uint2*>(vb + vo0);
    g.v1 = *reinterpret_cast<const uint2*>(vb + vo1);
    if (HAS_MASK) g.m = mp[min(k0 + (t & (BT - 1)), a.L - 1)];
  };
  auto lstore = [&](int k0, int buf, const Stage& g) {
    uint16_t* Ks = smem16 + buf * BIMG;
    uint16_t* Vt = Ks + BT * BK_PITCH;
    *reinterpret_cast<uint4*>(Ks + k_key * BK_PITCH + 8 * k_ch) = g.k;
    uint32_t* vd = reinterpret_cast<uint32_t*>(Vt + (4 * v_d4) * BV_PITCH + 2 * v_kp);
    vd[0] = __builtin_amdgcn_perm(g.v1.x, g.v0.x, 0x05040100u);
    vd[BV_PITCH / 2] = __builtin_amdgcn_perm(g.v1.x, g.v0.x, 0x07060302u);
    vd[BV_PITCH] = __builtin_amdgcn_perm(g.v1.y, g.v0.y, 0x05040100u);
    vd[3 * BV_PITCH / 2] = __builtin_amdgcn_perm(g.v1.y, g.v0.y, 0x07060302u);
    if ((HAS_MASK || k0 + BT > k_end) && t < BT) {
      bool dead = k0 + t >= k_end;
      if (HAS_MASK) dead = dead || g.m != 0;
      bias_s[buf][t] = dead ? -INFINITY : 0.f;
    }
  };

  f32x16 O;
#pragma unroll
  for (int r = 0; r < 16; ++r) O[r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  float m_use = 0.f;        // m_run, or 0 while m_run is still -inf
  f32x16 NM;                // -m_use in every accumulator register
  f32x16 Lacc;              // !DROP: the running row sum, sixteen identical copies (see the second product)
  bf16x8 ones;
#pragma unroll
  for (int r = 0; r < 16; ++r) NM[r] = Lacc[r] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.f;

  // tile T travels in ring slot T % PF; iteration `it` multiplies tile `it` out of image it & 1, moves tile it+1 from
  // its slot into the other image and re-issues that slot for tile it+1+PF
  gload(k_base, st[0]);
  lstore(k_base, 0, st[0]);
  // Loads are issued UNCONDITIONALLY (past the range: the worker's last tile again, never stored): only then is the
  // number of loads in flight a compile-time constant and the wait in front of a slot's lstore a counted
  // s_waitcnt vmcnt(2 * loads-per-tile); with loads under `if (tile < n_tiles)` hipcc has to assume the younger ones
  // were skipped and waits for everything, which serialises the ring.
#pragma unroll
  for (int i = 1; i <= PF; ++i) {
    gload(k_base + min(i, n_tiles - 1) * BT, st[i % PF]);
    __builtin_amdgcn_sched_barrier(0);     // keep the issue order = the order the loop consumes the slots in
  }
  auto step = [&](const int it, Stage& g) -> bool {    // g = slot (it + 1) % PF
    const int k0 = k_base + it * BT;
    const int buf = it & 1;
#ifndef PETR_DIAG_BF16_NO_BARRIER
    __syncthreads();   // image `buf` is complete; every wave is done reading the other one
#endif
    const int kw = kh * 64;     // this wave's keys inside the tile
#ifndef PETR_DIAG_BF16_NO_STAGE
    lstore(k0 + BT, buf ^ 1, g);     // unconditional as well (after the last tile: into the image nobody reads any more)
    gload(k_base + min(it + 1 + PF, n_tiles - 1) * BT, g);
#endif
    const uint16_t* Ks = smem16 + buf * BIMG;
    const uint16_t* Vt = Ks + BT * BK_PITCH;
    const bool use_bias = HAS_MASK || (k0 + BT > k_end);
    if (wave_active && k0 + kw < k_end) {
      uint4 kfr[2][2];
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#ifdef PETR_DIAG_BF16_NO_LDSREAD
          kfr[sb][j] = make_uint4(lane + sb, lane + j, it, 0x3f803f80u);
#else
          kfr[sb][j] = *reinterpret_cast<const uint4*>(Ks + (kw + 32 * sb + c) * BK_PITCH + 16 * j + 8 * h);
#endif
      // The accumulator input of the first product is NM = -m (the row's reference maximum, sixteen copies that only
      // change on the rare rescale), so the scores come out of the matrix pipe already shifted: no subtract pass.
      f32x16 S[2];
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
#ifdef PETR_DIAG_BF16_NO_QK
        S[sb] = NM;
        S[sb][sb] += __uint_as_float(kfr[sb][0].x & 0x3f800000u) + (float)qf[0][0];
#else
        S[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kfr[sb][0]), qf[0], NM, 0, 0, 0);
        S[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kfr[sb][1]), qf[1], S[sb], 0, 0, 0);
#endif
      }
      if (use_bias) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
          for (int r = 0; r < 16; ++r) S[sb][r] += bias_s[buf][kw + 32 * sb + mfma32_row(r, h)];
      }
      float mx = fmaxf(S[0][0], S[1][0]);
#ifndef PETR_DIAG_BF16_NO_MAX
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, S[0][r]), S[1][r]);   // v_max3_f32
#endif
      mx = xhalf_max(mx);      // row maximum RELATIVE to the reference maximum
      // lazy reference maximum (see the fp32 kernel): it moves only when a row has outgrown it by 2^LAZY
      if (__builtin_amdgcn_ballot_w64(mx > LAZY || (m_run == -INFINITY && mx > -INFINITY)) != 0) {   // wave-uniform, rare
        const float m_new = fmaxf(m_run, m_use + mx);
        const float m_new_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float delta = m_new_use - m_use;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new_use);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          S[0][r] -= delta;
          S[1][r] -= delta;
          O[r] *= alpha;
          NM[r] = -m_new_use;
          if (!DROP) Lacc[r] *= alpha;
        }
        l_run *= alpha;
        m_run = m_new;
        m_use = m_new_use;
      }
#ifndef PETR_DIAG_BF16_NO_EXP
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[sb][r] = __builtin_amdgcn_exp2f(S[sb][r]);
#endif
      if (DROP) {
        // the row sum keeps the UNdropped probabilities: plain f32 adds in ONE dependent chain (packed f32 VALU, which
        // is what the SLP vectoriser makes of independent neighbours, costs more issue cycles next to MFMAs than the
        // two scalar instructions it replaces)
        float acc = 0.f;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc += S[sb][r];
        l_run += xhalf_sum(acc);
      }
      // V fragments: requested only now (16 registers that the softmax above needs; four waves per SIMD cover
      // the LDS latency)
      uint2 vfr[2][2][2];
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 2; ++e)
#ifdef PETR_DIAG_BF16_NO_LDSREAD
            vfr[sb][j][e] = make_uint2(0x3f803f80u + lane + sb + j, 0x3f803f80u + e + it);
#else
            vfr[sb][j][e] = *reinterpret_cast<const uint2*>(Vt + c * BV_PITCH + kw + 32 * sb + 16 * j + 4 * h + 8 * e);
#endif
      if (DROP) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
          unsigned long long bal[16];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const uint32_t hsh = drop_pair_hash(drop_rk, (uint32_t)(k0 + kw + 32 * sb + mfma32_row(2 * i, h)) >> 1);
            const bool k0b = (uint16_t)hsh >= thr16, k1b = (uint16_t)(hsh >> 16) >= thr16;     // SDWA half-word compares: no extraction
            S[sb][2 * i] = k0b ? S[sb][2 * i] : 0.f;
            S[sb][2 * i + 1] = k1b ? S[sb][2 * i + 1] : 0.f;
            bal[2 * i] = __builtin_amdgcn_ballot_w64(k0b);
            bal[2 * i + 1] = __builtin_amdgcn_ballot_w64(k1b);
          }
          if (p.drop_bits && k0 + kw + 32 * sb < p.lpad)        // wave-uniform: the mask for the backward
            bits_store_block(p.drop_bits + ((long)bh * p.nqt32 + (qb * 4 + qg)) * p.lpad + (k0 + kw + 32 * sb), bal);
        }
      }
#pragma unroll
      for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x8 pf;
#pragma unroll
          for (int i = 0; i < 8; ++i) pf[i] = (__bf16)S[sb][8 * j + i];
          // without dropout the row sums ride on the matrix pipe: ones x P^T adds sum_k P[k][q] (both lane halves'
          // keys) to every register of Lacc - one MFMA issue slot instead of eight adds, and the normaliser is the
          // sum of exactly the bf16 probabilities that multiply V
#ifndef PETR_DIAG_BF16_NO_LACC
          if (!DROP) Lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, Lacc, 0, 0, 0);
#else
          Lacc[j] += (float)pf[0];
#endif
          const uint4 vraw = make_uint4(vfr[sb][j][0].x, vfr[sb][j][0].y, vfr[sb][j][1].x, vfr[sb][j][1].y);
#ifdef PETR_DIAG_BF16_NO_PV
          O[4 * sb + j] += __uint_as_float((vraw.x ^ __builtin_bit_cast(uint4, pf).x) & 0x3f800000u);
#else
          O = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vraw), pf, O, 0, 0, 0);
#endif
        }
    }
    return it + 1 >= n_tiles;
  };
  for (int it = 0;; it += PF) {
    static_assert(PF == 3, "the ring is unrolled by hand");
    if (step(it, st[1])) break;
    if (step(it + 1, st[2])) break;
    if (step(it + 2, st[0])) break;
  }
  if (!DROP) l_run = Lacc[0];

  if (DROP && p.drop_bits) asm volatile("s_dcache_wb" ::: "memory");     // the scalar-stored mask words leave the scalar cache
  // ---- merge the two key halves of each query group (upper wave -> LDS -> lower wave) ----
  float* smem = reinterpret_cast<float*>(smem16);
  __syncthreads();
  float* mo = smem + (qg * 64 + lane) * MERGE_PITCH;
  float* mml = smem + 4 * 64 * MERGE_PITCH + qg * 64 + 2 * c;
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) mo[r] = O[r];
    if (h == 0) { mml[0] = m_run; mml[1] = l_run; }
  }
  __syncthreads();
  if (kh == 1 || q_row >= a.Q) return;
  {
    const float m2 = mml[0], l2 = mml[1];
    const float m_new = fmaxf(m_run, m2);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float a1 = __builtin_amdgcn_exp2f(m_run - m_use), a2 = __builtin_amdgcn_exp2f(m2 - m_use);
#pragma unroll
    for (int r = 0; r < 16; ++r) O[r] = O[r] * a1 + mo[r] * a2;
    l_run = l_run * a1 + l2 * a2;
    m_run = m_new;
  }
  if (p.n_split == 1) {
    const float inv = (DROP ? p.drop.scale : 1.f) / l_run;
    float* op = a.o + (long)b * a.o_bs + (long)hd * a.o_hs + (long)q_row * a.o_rs;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = 8 * g + 4 * h;
      op[d0] = O[4 * g] * inv;
      op[d0 + 1] = O[4 * g + 1] * inv;
      op[d0 + 2] = O[4 * g + 2] * inv;
      op[d0 + 3] = O[4 * g + 3] * inv;
    }
    if (a.lse && h == 0) a.lse[(long)bh * a.Q + q_row] = (m_run + log2f(l_run)) * LN2;
  } else {
    const long row = ((long)split * a.B * a.H + bh) * a.Q + q_row;
    float* op = p.o_part + row * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(op + 8 * g + 4 * h) = make_float4(O[4 * g], O[4 * g + 1], O[4 * g + 2], O[4 * g + 3]);
    if (h == 0) {
      p.ml_part[row * 2] = m_run;
      p.ml_part[row * 2 + 1] = l_run;
    }
  }
}

// fp32 -> bf16 (round to nearest even), 8 elements per thread
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, long n8, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n8) {
    const float4 u = reinterpret_cast<const float4*>(x)[2 * i], v = reinterpret_cast<const float4*>(x)[2 * i + 1];
    bf16x8 o = {(__bf16)u.x, (__bf16)u.y, (__bf16)u.z, (__bf16)u.w, (__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    reinterpret_cast<uint4*>(y)[i] = __builtin_bit_cast(uint4, o);
  } else if (i == n8) {
    for (long j = 8 * n8; j < n; ++j) y[j] = __builtin_bit_cast(uint16_t, (__bf16)x[j]);
  }
}

// merge the L-split partials: thread = (row, 4 consecutive d)
__global__ __launch_bounds__(256) void mha_combine_kernel(const MhaFwdParams p) {
  const petr_mha_fwd_args& a = p.a;
  const long rows = (long)a.B * a.H * a.Q;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * 8) return;
  const long row = idx >> 3;
  const int d4 = (int)(idx & 7);
  float M = -INFINITY;
  for (int s = 0; s < p.n_split; ++s) M = fmaxf(M, p.ml_part[((long)s * rows + row) * 2]);
  float L = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < p.n_split; ++s) {
    const float ms = p.ml_part[((long)s * rows + row) * 2];
    const float ls = p.ml_part[((long)s * rows + row) * 2 + 1];
    // M == -inf (every key of the row masked): exp2(-inf - -inf) = NaN, propagating like the reference
    const float wgt = __builtin_amdgcn_exp2f(ms - M);
    const float4 o = *reinterpret_cast<const float4*>(p.o_part + ((long)s * rows + row) * 32 + 4 * d4);
    L += ls * wgt;
    acc.x += o.x * wgt; acc.y += o.y * wgt; acc.z += o.z * wgt; acc.w += o.w * wgt;
  }
  const float inv = p.drop.scale / L;   // scale == 1 without dropout
  const int q = (int)(row % a.Q);
  const int bh = (int)(row / a.Q);
  const int b = bh / a.H, hd = bh - b * a.H;
  float* op = a.o + (long)b * a.o_bs + (long)hd * a.o_hs + (long)q * a.o_rs + 4 * d4;
  op[0] = acc.x * inv; op[1] = acc.y * inv; op[2] = acc.z * inv; op[3] = acc.w * inv;
  if (a.lse && d4 == 0) a.lse[row] = (M + log2f(L)) * LN2;
}

}  // namespace

extern "C" int petr_mha_choose_split(int B, int H, int Q, int L) {
  const long base = cdiv(Q, 128) * (long)B * H;
  const int tiles = (int)cdiv(L, KV_TILE);
  int best = 1;
  double best_cost = 1e30;
  for (int ns = 1; ns <= tiles && ns <= 64; ++ns) {
    const int per = (int)cdiv(tiles, ns);
    if ((long)(ns - 1) * per >= tiles) continue;   // would leave an empty split
    const long wgs = base * ns;
    double cost = (double)cdiv(wgs, 256) * per;
    if (wgs < 512) cost *= 1.5;                    // < 2 workgroups per CU: softmax cannot hide under MFMA
    if (ns > 1) cost += 0.25;                      // merge pass
    if (cost < best_cost - 1e-9) { best_cost = cost; best = ns; }
  }
  return best;
}

extern "C" size_t petr_mha_fwd_workspace_bytes(int B, int H, int Q, int L, int n_split) {
  if (n_split <= 0) n_split = petr_mha_choose_split(B, H, Q, L);
  if (n_split == 1) return 0;
  return (size_t)n_split * B * H * Q * (32 + 2) * sizeof(float);
}

extern "C" int petr_mha_fwd(const petr_mha_fwd_args* ap, void* stream) {
  PETR_CHECK(ap && ap->q && ap->k && ap->v && ap->o, PETR_ERR_INVALID, "mha_fwd: null pointer");
  PETR_CHECK(ap->B > 0 && ap->H > 0 && ap->Q > 0 && ap->L > 0, PETR_ERR_INVALID, "mha_fwd: bad shape");
  MhaFwdParams p;
  p.pair = 0;
  p.a = *ap;
  const petr_mha_fwd_args& a = p.a;
  p.nqb = (int)cdiv(a.Q, 128);
  int ns = a.n_split > 0 ? a.n_split : petr_mha_choose_split(a.B, a.H, a.Q, a.L);
  const int tiles = (int)cdiv(a.L, KV_TILE);
  if (ns > tiles) ns = tiles;
  p.n_split = ns;
  p.n_tiles = tiles;
  p.n_sub = (int)cdiv(a.L, 32);   // ns <= tiles <= n_sub: every static range owns at least one 32-key half tile
  p.sched = ns > 1 ? a.sched : nullptr;
  p.drop_bits = a.drop.p > 0.f ? a.drop_bits : nullptr;
  p.nqt32 = (int)cdiv(a.Q, 32);
  p.lpad = 32 * (int)cdiv(a.L, 32);
  PETR_CHECK(a.drop.p >= 0.f && a.drop.p < 1.f, PETR_ERR_INVALID, "mha_fwd: dropout p=%g outside [0,1)", (double)a.drop.p);
  PETR_CHECK((long)a.B * a.H * a.Q < (1L << 32), PETR_ERR_UNSUPPORTED, "mha_fwd: dropout row index needs B*H*Q < 2^32");
  p.drop = make_drop(a.drop);
  p.n_tickets = tiles - (2 * ns < tiles ? 2 * ns : tiles) + ns;   // successful draws + one miss per worker
  p.o_part = nullptr;
  p.ml_part = nullptr;
  if (ns > 1) {
    const size_t need = petr_mha_fwd_workspace_bytes(a.B, a.H, a.Q, a.L, ns);
    PETR_CHECK(a.ws && a.ws_bytes >= need && aligned16(a.ws), PETR_ERR_WORKSPACE,
               "mha_fwd: workspace %zu < %zu bytes", a.ws_bytes, need);
    p.o_part = (float*)a.ws;
    p.ml_part = p.o_part + (size_t)ns * a.B * a.H * a.Q * 32;
  }
  p.q_vec = aligned16(a.q) && !(a.q_bs & 3) && !(a.q_hs & 3) && !(a.q_rs & 3);
  p.kv_vec = aligned16(a.k) && aligned16(a.v) && !(a.k_bs & 3) && !(a.k_hs & 3) && !(a.k_rs & 3) && !(a.v_bs & 3) &&
             !(a.v_hs & 3) && !(a.v_rs & 3);
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)p.nqb * a.B * a.H * ns;
  PETR_CHECK(total < (1L << 31), PETR_ERR_UNSUPPORTED, "mha_fwd: grid too large");
  hipEvent_t ev0, ev1;   // null unless bench.py's profiler is on: then they carry this dispatch's begin/end
  petr_prof_claim(PETR_PROF_MHA_FWD + 16 * (a.L > a.Q ? 1 : 0), &ev0, &ev1);
  const bool vec = p.q_vec && p.kv_vec;
  auto launch = [&](auto kern) {
    hipExtLaunchKernelGGL(kern, dim3((unsigned)total), dim3(512), 0, s, ev0, ev1, 0, p);
  };
  const int variant = (p.drop.thr ? 8 : 0) | (a.kpm ? 4 : 0) | (vec ? 2 : 0) | (p.sched ? 1 : 0);
  switch (variant) {
    case 0: launch(mha_fwd_kernel<false, false, false, false>); break;
    case 1: launch(mha_fwd_kernel<false, false, true, false>); break;
    case 2: launch(mha_fwd_kernel<false, true, false, false>); break;
    case 3: launch(mha_fwd_kernel<false, true, true, false>); break;
    case 4: launch(mha_fwd_kernel<true, false, false, false>); break;
    case 5: launch(mha_fwd_kernel<true, false, true, false>); break;
    case 6: launch(mha_fwd_kernel<true, true, false, false>); break;
    case 7: launch(mha_fwd_kernel<true, true, true, false>); break;
    case 8: launch(mha_fwd_kernel<false, false, false, true>); break;
    case 9: launch(mha_fwd_kernel<false, false, true, true>); break;
    case 10: launch(mha_fwd_kernel<false, true, false, true>); break;
    case 11: launch(mha_fwd_kernel<false, true, true, true>); break;
    case 12: launch(mha_fwd_kernel<true, false, false, true>); break;
    case 13: launch(mha_fwd_kernel<true, false, true, true>); break;
    case 14: launch(mha_fwd_kernel<true, true, false, true>); break;
    default: launch(mha_fwd_kernel<true, true, true, true>); break;
  }
  PETR_LAUNCH_CHECK("mha_fwd");
  if (ns > 1 && !a.defer_merge) {
    const long n = (long)a.B * a.H * a.Q * 8;
    hipLaunchKernelGGL(mha_combine_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, p);
    PETR_LAUNCH_CHECK("mha_combine");
  }
  return PETR_OK;
}

static int bf16_split(int B, int H, int Q, int L) {
  const long base = cdiv(Q, 128) * (long)B * H;
  const long n_sub = cdiv(L, 64);
  long ns = cdiv(256, base);                 // one 8-wave workgroup per CU: the softmax saturates the SIMDs' issue slots
                                             // at two waves each (measured: 512 workgroups are 5-20 % slower than 256)
  if (ns > n_sub / 4) ns = n_sub / 4;        // but at least two tiles per worker
  if (ns > 64) ns = 64;
  return ns < 1 ? 1 : (int)ns;
}

extern "C" int petr_mha_fwd_bf16_choose_split(int B, int H, int Q, int L) { return bf16_split(B, H, Q, L); }

extern "C" size_t petr_mha_fwd_bf16_workspace_bytes(int B, int H, int Q, int L, int n_split) {
  if (n_split <= 0) n_split = bf16_split(B, H, Q, L);
  if (n_split == 1) return 0;
  return (size_t)n_split * B * H * Q * (32 + 2) * sizeof(float);
}

extern "C" int petr_mha_fwd_bf16(const petr_mha_fwd_bf16_args* ap, void* stream) {
  PETR_CHECK(ap && ap->q && ap->k && ap->v && ap->o, PETR_ERR_INVALID, "mha_fwd_bf16: null pointer");
  PETR_CHECK(ap->B > 0 && ap->H > 0 && ap->Q > 0 && ap->L > 0, PETR_ERR_INVALID, "mha_fwd_bf16: bad shape");
  static_assert(sizeof(petr_mha_fwd_bf16_args) == sizeof(petr_mha_fwd_args), "the two argument blocks share one layout");
  MhaFwdParams p;
  memcpy(&p.a, ap, sizeof(p.a));     // k / v are bf16 pointers with strides in bf16 elements
  const petr_mha_fwd_args& a = p.a;
  PETR_CHECK(aligned16(a.k) && aligned16(a.v) && !(a.k_bs & 7) && !(a.k_hs & 7) && !(a.k_rs & 7) && !(a.v_bs & 7) &&
                 !(a.v_hs & 7) && !(a.v_rs & 7),
             PETR_ERR_UNSUPPORTED, "mha_fwd_bf16: K/V rows must be 16-byte aligned (strides multiples of 8 elements)");
  PETR_CHECK((long)a.L * a.k_rs < (1L << 31) && (long)a.L * a.v_rs < (1L << 31), PETR_ERR_UNSUPPORTED,
             "mha_fwd_bf16: per-head K/V extent needs 32-bit element offsets");
  p.nqb = (int)cdiv(a.Q, 128);
  p.n_sub = (int)cdiv(a.L, 64);
  int ns = a.n_split > 0 ? a.n_split : bf16_split(a.B, a.H, a.Q, a.L);
  if (ns > p.n_sub) ns = p.n_sub;    // every static range owns at least one 64-key half tile
  p.n_split = ns;
  p.n_tiles = (int)cdiv(a.L, BT);
  p.n_tickets = 0;
  p.sched = nullptr;
  p.drop_bits = a.drop.p > 0.f ? a.drop_bits : nullptr;
  p.nqt32 = (int)cdiv(a.Q, 32);
  p.lpad = 32 * (int)cdiv(a.L, 32);
  PETR_CHECK(a.drop.p >= 0.f && a.drop.p < 1.f, PETR_ERR_INVALID, "mha_fwd_bf16: dropout p=%g outside [0,1)", (double)a.drop.p);
  PETR_CHECK((long)a.B * a.H * a.Q < (1L << 32), PETR_ERR_UNSUPPORTED, "mha_fwd_bf16: dropout row index needs B*H*Q < 2^32");
  p.drop = make_drop(a.drop);
  p.o_part = nullptr;
  p.ml_part = nullptr;
  if (ns > 1) {
    const size_t need = petr_mha_fwd_bf16_workspace_bytes(a.B, a.H, a.Q, a.L, ns);
    PETR_CHECK(a.ws && a.ws_bytes >= need && aligned16(a.ws), PETR_ERR_WORKSPACE,
               "mha_fwd_bf16: workspace %zu < %zu bytes", a.ws_bytes, need);
    p.o_part = (float*)a.ws;
    p.ml_part = p.o_part + (size_t)ns * a.B * a.H * a.Q * 32;
  }
  p.q_vec = p.kv_vec = 0;
  static const int pair_on = petr_tune("PETR_MHA16_PAIR", 1) != 0;
  p.pair = pair_on && !(a.H & 1) && a.k_hs == 32 && a.v_hs == 32;
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)p.nqb * a.B * a.H * ns;
  PETR_CHECK(total < (1L << 31), PETR_ERR_UNSUPPORTED, "mha_fwd_bf16: grid too large");
  hipEvent_t ev0, ev1;
  petr_prof_claim(PETR_PROF_MHA_FWD + 16 * (a.L > a.Q ? 1 : 0), &ev0, &ev1);
  auto launch = [&](auto kern) {
    hipExtLaunchKernelGGL(kern, dim3((unsigned)total), dim3(512), 0, s, ev0, ev1, 0, p);
  };
  switch ((p.drop.thr ? 2 : 0) | (a.kpm ? 1 : 0)) {
    case 0: launch(mha_fwd_bf16_kernel<false, false>); break;
    case 1: launch(mha_fwd_bf16_kernel<true, false>); break;
    case 2: launch(mha_fwd_bf16_kernel<false, true>); break;
    default: launch(mha_fwd_bf16_kernel<true, true>); break;
  }
  PETR_LAUNCH_CHECK("mha_fwd_bf16");
  if (ns > 1 && !a.defer_merge) {
    const long n = (long)a.B * a.H * a.Q * 8;
    hipLaunchKernelGGL(mha_combine_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, p);
    PETR_LAUNCH_CHECK("mha_combine");
  }
  return PETR_OK;
}

extern "C" int petr_cast_bf16(const float* x, uint16_t* y, long n, void* stream) {
  PETR_CHECK(x && y && n >= 0, PETR_ERR_INVALID, "cast_bf16: bad arguments");
  if (n == 0) return PETR_OK;
  PETR_CHECK(aligned16(x) && aligned16(y), PETR_ERR_UNSUPPORTED, "cast_bf16: buffers must be 16-byte aligned");
  const long n8 = n / 8;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)cdiv(n8 + 1, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n8, n);
  PETR_LAUNCH_CHECK("cast_bf16");
  return PETR_OK;
}

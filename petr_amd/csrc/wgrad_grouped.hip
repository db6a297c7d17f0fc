// Grouped weight gradients (petr_hip.h "Grouped weight gradients"): every parameter gradient of one decoder layer's
// linear maps - what autograd accumulates into .grad for the modules built at reference petr_transformer.py:158-224
// (nn.MultiheadAttention in / out projections, mmcv FFN) - as ONE launch of 64 x 64 output tiles.
//
// Why: at 900 queries each of these contractions is a 256 x 256 ... 2048 x 256 output over K = 900 rows (0.1-0.9 GFLOP).
// As separate petr_gemm launches (16-128 tiles x 14-32 K slices, float-atomic epilogues) they took 41 us apiece beside the
// decoder chain: 52 launches = 30 % of the c5 step's kernel time, and without them the chain itself ran 0.5 ms faster
// (profiles/r02_bench_c5_timed_only_kernel_stats.csv; round-3 A/B with the contractions skipped).  Here a workgroup owns an
// output tile for the whole K range: operands stream through a double-buffered LDS image one 32-row K step at a time
// ([k][column] rows, exactly as they lie in memory: both operands are K-major, nothing is transposed), four waves hold a
// 32 x 32 accumulator each, and the result is added to dw with plain loads / stores - no atomics, no partial slabs, one
// launch per layer stage.
#include "common.h"

namespace {

struct WgItem {
  const float* a;
  const float* b;
  float* c;
  float* colsum;
  int lda, ldb, ldc;
  int M, N, K;
  int ksplit, ksteps_per;   // K slices and 32-row steps per slice
  int wg0, tiles_n, tiles;  // first workgroup, tiles along N, tiles in all
};
struct WgGroup {
  WgItem it[PETR_WGRAD_MAX];
  int n;
};

// 512 threads = two K groups of four waves: group g walks the K steps step0 + g, step0 + g + 2, ... of the workgroup's K
// range with its own double-buffered LDS images, and the two 64 x 64 partial tiles are summed through LDS at the end - an
// in-workgroup K split: twice the waves per tile to hide the operand latency (a 900-row item is only 29 K steps long and
// 352 tiles put 1.4 workgroups on a CU), still one owner per tile, so no atomics and a bit-reproducible sum.
// Operands of a group's step i + 2 are requested while step i is multiplied (two register sets).
__global__ __launch_bounds__(512) void wgrad_grouped_kernel(const WgGroup grp) {
  __shared__ __attribute__((aligned(16))) float As[2][2][32 * 64];
  __shared__ __attribute__((aligned(16))) float Bs[2][2][32 * 64];

  // which item / tile / K slice (wave-uniform scalar search over <= 16 items)
  // Workgroup ids are dealt round-robin to the 8 XCDs: remapped so that CONSECUTIVE logical ids share an XCD, and tiles are
  // numbered with the shorter tile dimension fastest, a chunk of consecutive tiles touches all (few) column blocks of one
  // operand and only a few of the other - the ~46 tiles an XCD holds at 900 rows then read ~16 operand column blocks of
  // 230 KB (its 4 MB L2) instead of one pair per tile (64 x 64 tiles re-read each operand block up to 32 times).
  const int wid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  int ii = 0;
#pragma unroll 1
  for (int i = 1; i < grp.n; ++i)
    if (wid >= grp.it[i].wg0) ii = i;
  const WgItem& it = grp.it[ii];
  const int local = wid - it.wg0;
  const int ks = local / it.tiles;
  const int tile = local - ks * it.tiles;
  const int tiles_m = it.tiles / it.tiles_n;
  int tm, tn;
  if (tiles_m <= it.tiles_n) { tn = tile / tiles_m; tm = tile - tn * tiles_m; }
  else { tm = tile / it.tiles_n; tn = tile - tm * it.tiles_n; }
  const int m0 = tm * 64, n0 = tn * 64;
  const int step0 = ks * it.ksteps_per;
  const int ksteps_all = (it.K + 31) >> 5;
  const int step1 = min(ksteps_all, step0 + it.ksteps_per);

  const int grpk = threadIdx.x >> 8;                     // K group
  const int t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;
  const int h = lane >> 5, c = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  // staging: thread -> rows (t >> 4) and (t >> 4) + 16 of the K step, 4 columns at 4 * (t & 15)
  const int sr = t >> 4, sc = 4 * (t & 15);
  const float* ap = it.a + m0 + sc;
  const float* bp = it.b + n0 + sc;
  float* Ag = &As[grpk][0][0];
  float* Bg = &Bs[grpk][0][0];
  // (plain float4 variables and macros, not structs behind lambdas: hipcc kept those in scratch memory)
#define WG_GLOAD(A0, A1, B0, B1, STEP)                                                         \
  do {                                                                                         \
    const int k0__ = (STEP) * 32 + sr;                                                         \
    const int kc0__ = min(k0__, it.K - 1), kc1__ = min(k0__ + 16, it.K - 1);                   \
    A0 = *reinterpret_cast<const float4*>(ap + (size_t)kc0__ * it.lda);                        \
    A1 = *reinterpret_cast<const float4*>(ap + (size_t)kc1__ * it.lda);                        \
    B0 = *reinterpret_cast<const float4*>(bp + (size_t)kc0__ * it.ldb);                        \
    B1 = *reinterpret_cast<const float4*>(bp + (size_t)kc1__ * it.ldb);                        \
  } while (0)
// (component-wise selects: `ok ? float4 : float4` on the vector CLASS type makes hipcc select between two scratch addresses)
#define WG_SEL(OK, V) make_float4((OK) ? (V).x : 0.f, (OK) ? (V).y : 0.f, (OK) ? (V).z : 0.f, (OK) ? (V).w : 0.f)
#define WG_STAGE(A0, A1, B0, B1, STEP, BUF)                                                    \
  do {                                                                                         \
    const bool ok0__ = (STEP) * 32 + sr < it.K, ok1__ = (STEP) * 32 + sr + 16 < it.K;          \
    *reinterpret_cast<float4*>(Ag + (BUF) * 2048 + sr * 64 + sc) = WG_SEL(ok0__, A0);          \
    *reinterpret_cast<float4*>(Ag + (BUF) * 2048 + (sr + 16) * 64 + sc) = WG_SEL(ok1__, A1);   \
    *reinterpret_cast<float4*>(Bg + (BUF) * 2048 + sr * 64 + sc) = WG_SEL(ok0__, B0);          \
    *reinterpret_cast<float4*>(Bg + (BUF) * 2048 + (sr + 16) * 64 + sc) = WG_SEL(ok1__, B1);   \
  } while (0)

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float csum = 0.f;
  const bool do_colsum = it.colsum != nullptr && tn == 0 && wave == 0;

  // this group's steps: first + 2 j, j = 0 .. ; both groups run the same number of loop iterations (barriers)
  const int first = step0 + grpk;
  const int iters = (step1 - step0 + 1) >> 1;
  float4 pa0 = make_float4(0.f, 0.f, 0.f, 0.f), pa1 = pa0, pb0 = pa0, pb1 = pa0;
  float4 qa0 = pa0, qa1 = pa0, qb0 = pa0, qb1 = pa0;
  if (first < step1) WG_GLOAD(pa0, pa1, pb0, pb1, first);
  if (first + 2 < step1) WG_GLOAD(qa0, qa1, qb0, qb1, first + 2);
  // two register sets, loop unrolled by two: a set is re-filled (step + 4) right behind the barrier that published its step
#define WG_BODY(A0, A1, B0, B1, J, BUF)                                                        \
  do {                                                                                         \
    const int step__ = first + 2 * (J);                                                        \
    const bool on__ = step__ < step1; /* group-uniform */                                      \
    if (on__) WG_STAGE(A0, A1, B0, B1, step__, BUF);                                           \
    __syncthreads(); /* image BUF complete; every wave is done with the previous step's image */ \
    if (on__) {                                                                                \
      if (step__ + 4 < step1) WG_GLOAD(A0, A1, B0, B1, step__ + 4);                            \
      const float* Ab__ = Ag + (BUF) * 2048 + wm * 32 + c;                                     \
      const float* Bb__ = Bg + (BUF) * 2048 + wn * 32 + c;                                     \
      float af__[16], bf__[16]; /* all fragments of the step requested before the first product */ \
      _Pragma("unroll") for (int s__ = 0; s__ < 16; ++s__) {                                   \
        af__[s__] = Ab__[(2 * s__ + h) * 64];                                                  \
        bf__[s__] = Bb__[(2 * s__ + h) * 64];                                                  \
      }                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                       \
      _Pragma("unroll") for (int s__ = 0; s__ < 16; ++s__)                                     \
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af__[s__], bf__[s__], acc, 0, 0, 0);        \
      if (do_colsum) {                                                                         \
        _Pragma("unroll") for (int k__ = 0; k__ < 32; ++k__) csum += Ag[(BUF) * 2048 + k__ * 64 + lane]; \
      }                                                                                        \
    }                                                                                          \
  } while (0)
  for (int j = 0; j < iters; j += 2) {
    WG_BODY(pa0, pa1, pb0, pb1, j, 0);
    if (j + 1 < iters) WG_BODY(qa0, qa1, qb0, qb1, j + 1, 1);
  }
#undef WG_BODY
#undef WG_GLOAD
#undef WG_STAGE
#undef WG_SEL

  // ---- the odd group's partial tile (and bias sums) crosses LDS; the even group adds it and owns the epilogue ----
  __syncthreads();
  float* xch = &As[0][0][0];                     // 4 waves x 16 registers x 64 lanes = 16 KB: the A images are free now
  float* xcs = &Bs[0][0][0];
  if (grpk == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xch[(wave * 16 + r) * 64 + lane] = acc[r];
    if (do_colsum) xcs[lane] = csum;
  }
  __syncthreads();
  if (grpk == 1) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += xch[(wave * 16 + r) * 64 + lane];
  if (do_colsum) csum += xcs[lane];

  // ---- epilogue: dw += acc (one owner per tile) or float atomics (K slices) ----
  float* cp = it.c + (size_t)(m0 + wm * 32) * it.ldc + n0 + wn * 32 + c;
  if (it.ksplit > 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) atomicAdd(cp + (size_t)mfma32_row(r, h) * it.ldc, acc[r]);
  } else {
    float old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) old[r] = cp[(size_t)mfma32_row(r, h) * it.ldc];
#pragma unroll
    for (int r = 0; r < 16; ++r) cp[(size_t)mfma32_row(r, h) * it.ldc] = old[r] + acc[r];
  }
  if (do_colsum) atomicAdd(it.colsum + m0 + lane, csum);
}

}  // namespace

extern "C" int petr_wgrad_grouped(const petr_wgrad_item* items, int n, void* stream) {
  PETR_CHECK(items && n > 0 && n <= PETR_WGRAD_MAX, PETR_ERR_INVALID, "wgrad_grouped: 1..%d items", PETR_WGRAD_MAX);
  WgGroup g;
  memset(&g, 0, sizeof g);
  g.n = n;
  long wgs = 0;
  for (int i = 0; i < n; ++i) {
    const petr_wgrad_item& s = items[i];
    PETR_CHECK(s.dy && s.x && s.dw && s.M > 0 && s.N > 0 && s.K > 0, PETR_ERR_INVALID, "wgrad_grouped: item %d: bad argument", i);
    PETR_CHECK(!(s.M & 63) && !(s.N & 63) && !(s.lda & 3) && !(s.ldb & 3) && aligned16(s.dy) && aligned16(s.x) &&
                   s.lda >= s.M && s.ldb >= s.N && s.ldc >= s.N && s.lda < (1L << 31) && s.ldb < (1L << 31) && s.ldc < (1L << 31),
               PETR_ERR_UNSUPPORTED, "wgrad_grouped: item %d: M, N must be multiples of 64, lda / ldb multiples of 4, operands 16-byte aligned", i);
    WgItem& d = g.it[i];
    d.a = s.dy; d.b = s.x; d.c = s.dw; d.colsum = s.db;
    d.lda = (int)s.lda; d.ldb = (int)s.ldb; d.ldc = (int)s.ldc;
    d.M = s.M; d.N = s.N; d.K = s.K;
    const int ksteps = (s.K + 31) / 32;
    int ks = s.ksplit < 1 ? 1 : s.ksplit;
    if (ks > ksteps) ks = ksteps;
    d.ksteps_per = (ksteps + ks - 1) / ks;
    d.ksplit = (ksteps + d.ksteps_per - 1) / d.ksteps_per;     // no empty slice
    d.tiles_n = s.N / 64;
    d.tiles = (s.M / 64) * d.tiles_n;
    d.wg0 = (int)wgs;
    wgs += (long)d.tiles * d.ksplit;
    PETR_CHECK(wgs < (1L << 30), PETR_ERR_UNSUPPORTED, "wgrad_grouped: grid too large");
  }
  hipLaunchKernelGGL(wgrad_grouped_kernel, dim3((unsigned)wgs), dim3(512), 0, (hipStream_t)stream, g);
  PETR_LAUNCH_CHECK("wgrad_grouped");
  return PETR_OK;
}

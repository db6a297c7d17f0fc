// Shared helpers for the libpetr_hip kernels (gfx950 / CDNA4 only: wave64, MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/petr_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void petr_set_error(const char* fmt, ...);
// CU count of the current device (hardware constant, read once per device; <= 0 on error)
int petr_num_cus();

// opt-in event timing of tagged launches (api.cpp); no-ops unless petr_prof_begin() was called
enum { PETR_PROF_MHA_FWD = 1, PETR_PROF_MHA_BWD = 2, PETR_PROF_GEMM = 3, PETR_PROF_COORDS3D = 4 };
int petr_prof_open_record(int tag, void* stream);
int petr_prof_claim(int tag, hipEvent_t* start, hipEvent_t* stop);   // events to attach to ONE dispatch (null when off)
void petr_prof_close_record(int rec, void* stream);

// execution context (side streams + event ring), owned by the host through petr_ctx_create/destroy
#define PETR_CTX_MAX_SIDE 4
#define PETR_CTX_EVENTS 512
struct petr_ctx {
  int n_side;
  int next_event;
  hipStream_t side[PETR_CTX_MAX_SIDE];
  hipEvent_t ev[PETR_CTX_EVENTS];
};

// fork/join helper: with ctx == nullptr every side stream IS the main stream and fork/join are no-ops
struct Lanes {
  hipStream_t main;
  petr_ctx* ctx;
  void* m() const { return (void*)main; }
  void* side(int i) const { return ctx ? (void*)ctx->side[i % ctx->n_side] : (void*)main; }
  hipEvent_t next() const {
    hipEvent_t e = ctx->ev[ctx->next_event];
    ctx->next_event = (ctx->next_event + 1) % PETR_CTX_EVENTS;
    return e;
  }
  // side stream i starts after everything enqueued on main so far
  void fork(int i) const {
    if (!ctx) return;
    hipEvent_t e = next();
    (void)hipEventRecord(e, main);
    (void)hipStreamWaitEvent(ctx->side[i % ctx->n_side], e, 0);
  }
  // side streams 0..n-1 start after everything enqueued on main so far: ONE event record on main (each record is a
  // barrier packet in main's queue and costs it ~7 us on MI355X), n waits
  void fork_first(int n) const {
    if (!ctx) return;
    hipEvent_t e = next();
    (void)hipEventRecord(e, main);
    for (int i = 0; i < n && i < ctx->n_side; ++i) (void)hipStreamWaitEvent(ctx->side[i], e, 0);
  }
  // main continues after everything enqueued on side stream i so far
  void join(int i) const {
    if (!ctx) return;
    hipEvent_t e = next();
    (void)hipEventRecord(e, ctx->side[i % ctx->n_side]);
    (void)hipStreamWaitEvent(main, e, 0);
  }
  void join_all() const {
    if (!ctx) return;
    for (int i = 0; i < ctx->n_side; ++i) join(i);
  }
};

// ---- dropout (petr_hip.h "Dropout"): host-derived keys + the device hashes every kernel shares ----
// One 32-bit hash serves the two columns (2j, 2j+1) of a row: its low / high 16 bits are compared with a 16-bit
// threshold, so p is realised as round(p * 65536) / 65536 (0.1 -> 0.100006) and kept values are scaled by the
// matching 1/(1-p).  The hash uses full-rate instructions only (xor, shift, 24-bit multiply-add): on gfx950 the
// f32 MFMA shares the vector issue port, so every VALU cycle spent here is paid in full, and v_mul_lo_u32 is
// quarter rate.
struct DropDev {
  uint32_t k0, k1, thr;   // thr = p * 2^16 ; thr == 0: disabled
  float scale;            // 1 / (1 - thr / 2^16)
};
inline DropDev make_drop(const petr_dropout& d) {
  DropDev r = {0u, 0u, 0u, 1.f};
  if (!(d.p > 0.f)) return r;
  uint64_t z = d.seed + 0x9E3779B97F4A7C15ull * (uint64_t)(d.site + 1);   // splitmix64 of (seed, site)
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  r.k0 = (uint32_t)z;
  r.k1 = (uint32_t)(z >> 32);
  const double t = (double)d.p * 65536.0 + 0.5;
  r.thr = t >= 65535.0 ? 65535u : (uint32_t)t;
  if (r.thr == 0u) r.thr = 1u;
  r.scale = (float)(1.0 / (1.0 - (double)r.thr / 65536.0));
  return r;
}
// per row (computed once per row / lane, so 32-bit multiplies are fine here)
__device__ __forceinline__ uint32_t drop_row_key(const DropDev& d, uint32_t row) {
  uint32_t x = (row ^ d.k0) * 0x9E3779B1u;
  x ^= x >> 15;
  x = (x + d.k1) * 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
// hash of the column pair j = col >> 1 of a row: 7 full-rate VALU instructions (statistics: tests/test_ops_gpu.py)
__device__ __forceinline__ uint32_t drop_pair_hash(uint32_t row_key, uint32_t pair) {
  uint32_t x = row_key ^ pair;
  x = __umul24(x, 0xC2B2AEu) + (row_key >> 7);   // v_mad_u32_u24: low 24 bits of x times a 24-bit odd constant
  x ^= x >> 13;
  x = __umul24(x, 0x85EBCBu) + row_key;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool drop_keep(uint32_t row_key, uint32_t col, uint32_t thr) {
  const uint32_t hsh = drop_pair_hash(row_key, col >> 1);
  return (uint16_t)((col & 1u) ? (hsh >> 16) : hsh) >= (uint16_t)thr;
}

// position of key c (0..31) inside a 32-word block of the packed key-major dropout mask: the order in which the forward's
// accumulator ballots fall (register r, lane half) - see petr_dropout_bits in petr_hip.h
__host__ __device__ __forceinline__ int petr_bits_slot(int c) { return 2 * ((c & 3) + 4 * (c >> 3)) + ((c >> 2) & 1); }

#define PETR_CHECK(cond, code, ...)     \
  do {                                  \
    if (!(cond)) {                      \
      petr_set_error(__VA_ARGS__);      \
      return (code);                    \
    }                                   \
  } while (0)

#define PETR_LAUNCH_CHECK(what)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      petr_set_error("%s: launch failed: %s", what, hipGetErrorString(e__)); \
      return PETR_ERR_LAUNCH;                                                \
    }                                                                        \
  } while (0)

// Tuning switches of the A/B builds.  The PRODUCT library never reads the environment: every switch is its compiled-in
// default (petr_tune(name, dflt) folds to dflt), so the library has no hidden process-global inputs.  A diagnostic build
// (make EXTRA=-DPETR_TUNING_ENV, scripts/ab_build.sh) reads the PETR_* variable once per call site instead: the same-box
// A/B timing of the alternatives DESIGN.md records (devices differ by several per cent, so only same-box runs rank variants).
#ifdef PETR_TUNING_ENV
#include <stdlib.h>
static inline int petr_tune(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
static inline constexpr int petr_tune(const char*, int dflt) { return dflt; }
#endif

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: raised once per (kernel, device) - one bit
// per device ordinal in a per-call-site word (racing first calls set the same attribute twice, harmlessly)
#include <atomic>
struct PetrLdsLimit { std::atomic<unsigned long long> done{0ull}; };
static inline void petr_raise_lds_limit(PetrLdsLimit& st, const void* kern, int bytes) {
  int dev = 0;
  const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
  const unsigned long long bit = known ? 1ull << dev : 0ull;
  if (known && (st.done.load(std::memory_order_relaxed) & bit)) return;
  (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (known) st.done.fetch_or(bit, std::memory_order_relaxed);
}

static inline long cdiv(long a, long b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Row index held by accumulator register r of a 32x32 MFMA result in lane-half h
// (C/D map of v_mfma_f32_32x32x*: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
__device__ __forceinline__ int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// value of the partner lane (lane ^ 32) combined by max / sum: one v_permlane32_swap, no LDS.
__device__ __forceinline__ float xhalf_max(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// XCD-aware remap of a linear workgroup id (cdna guide T1, bijective form): consecutive
// remapped ids run on the same XCD (round-robin dispatch: id % 8 labels the XCD group).
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, s = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

// Library-level entry points: version, thread-local error text, device capabilities.
#include "common.h"
#include <atomic>

static thread_local char g_err[512] = "";

void petr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int petr_version(void) { return PETR_HIP_VERSION; }
extern "C" const char* petr_last_error(void) { return g_err; }

extern "C" int petr_device_caps(int* num_cu, char* arch, int arch_len) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  PETR_CHECK(e == hipSuccess, PETR_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  PETR_CHECK(e == hipSuccess, PETR_ERR_LAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (num_cu) *num_cu = prop.multiProcessorCount;
  if (arch && arch_len > 0) snprintf(arch, (size_t)arch_len, "%s", prop.gcnArchName);
  return PETR_OK;
}

// CU count per device: a hardware constant, cached after the first query of each device (relaxed atomics: racing first
// calls store the same value)
int petr_num_cus() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (dev >= 0 && dev < 64) {
    const int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
  }
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  if (dev >= 0 && dev < 64 && n > 0) cache[dev].store(n, std::memory_order_relaxed);
  return n;
}

// ---------------------------------------------------------------------------------------------
// Opt-in kernel timing with HIP events on the launch stream (bench.py's roofline leg).  Disabled by
// default: the hot path then records nothing.  Events are created in petr_prof_begin (outside any
// launch path) and resolved in petr_prof_end after a synchronise.  A tagged launch either attaches its
// two events to the dispatch itself (petr_prof_claim + hipExtLaunchKernelGGL: the events then carry the
// kernel's own begin/end timestamps, the same interval rocprofv3 --kernel-trace reports) or, for a group
// of launches, records them around the group (open/close: includes the inter-packet gaps).
// ---------------------------------------------------------------------------------------------
namespace {
struct Prof {
  bool on = false;
  int cap = 0, n = 0;
  hipEvent_t* ev = nullptr;   // 2 per record
  int* tag = nullptr;
} g_prof;
}  // namespace

int petr_prof_open_record(int tag, void* stream) {
  if (!g_prof.on || g_prof.n >= g_prof.cap) return -1;
  const int i = g_prof.n++;
  g_prof.tag[i] = tag;
  (void)hipEventRecord(g_prof.ev[2 * i], (hipStream_t)stream);
  return i;
}

int petr_prof_claim(int tag, hipEvent_t* start, hipEvent_t* stop) {
  *start = nullptr;
  *stop = nullptr;
  if (!g_prof.on || g_prof.n >= g_prof.cap) return -1;
  const int i = g_prof.n++;
  g_prof.tag[i] = tag;
  *start = g_prof.ev[2 * i];
  *stop = g_prof.ev[2 * i + 1];
  return i;
}

void petr_prof_close_record(int i, void* stream) {
  if (i >= 0) (void)hipEventRecord(g_prof.ev[2 * i + 1], (hipStream_t)stream);
}

extern "C" int petr_prof_begin(int capacity) {
  PETR_CHECK(capacity > 0 && capacity <= (1 << 20), PETR_ERR_INVALID, "prof_begin: bad capacity");
  if (g_prof.cap < capacity) {
    for (int i = 0; i < 2 * g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev;
    delete[] g_prof.tag;
    g_prof.ev = new hipEvent_t[2 * capacity];
    g_prof.tag = new int[capacity];
    for (int i = 0; i < 2 * capacity; ++i) {
      hipError_t e = hipEventCreate(&g_prof.ev[i]);
      PETR_CHECK(e == hipSuccess, PETR_ERR_LAUNCH, "prof_begin: hipEventCreate: %s", hipGetErrorString(e));
    }
    g_prof.cap = capacity;
  }
  g_prof.n = 0;
  g_prof.on = true;
  return PETR_OK;
}

extern "C" int petr_prof_end(float* ms, int* tags, int cap, int* n_out) {
  g_prof.on = false;
  PETR_CHECK(ms && tags && n_out, PETR_ERR_INVALID, "prof_end: null pointer");
  const int n = g_prof.n < cap ? g_prof.n : cap;
  for (int i = 0; i < n; ++i) {
    hipError_t e = hipEventSynchronize(g_prof.ev[2 * i + 1]);
    PETR_CHECK(e == hipSuccess, PETR_ERR_LAUNCH, "prof_end: %s", hipGetErrorString(e));
    float t = 0.f;
    (void)hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
    ms[i] = t;
    tags[i] = g_prof.tag[i];
  }
  *n_out = n;
  return PETR_OK;
}

// ---------------------------------------------------------------------------------------------
// Execution context: side streams + event ring (see petr_hip.h).
// ---------------------------------------------------------------------------------------------
extern "C" int petr_ctx_create(petr_ctx** out, int n_side_streams) {
  PETR_CHECK(out && n_side_streams >= 1 && n_side_streams <= PETR_CTX_MAX_SIDE, PETR_ERR_INVALID,
             "ctx_create: 1..%d side streams", PETR_CTX_MAX_SIDE);
  petr_ctx* c = new petr_ctx();
  c->n_side = n_side_streams;
  c->next_event = 0;
  for (int i = 0; i < n_side_streams; ++i) {
    hipError_t e = hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking);
    PETR_CHECK(e == hipSuccess, PETR_ERR_LAUNCH, "ctx_create: hipStreamCreate: %s", hipGetErrorString(e));
  }
  for (int i = 0; i < PETR_CTX_EVENTS; ++i) {
    hipError_t e = hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming);
    PETR_CHECK(e == hipSuccess, PETR_ERR_LAUNCH, "ctx_create: hipEventCreate: %s", hipGetErrorString(e));
  }
  *out = c;
  return PETR_OK;
}

extern "C" int petr_ctx_join_into(petr_ctx* c, void* main_stream, void* target_stream) {
  // `target_stream` waits for everything enqueued so far on `main_stream` and on every side stream of `c`
  hipStream_t tgt = (hipStream_t)target_stream;
  const Lanes ln{(hipStream_t)main_stream, c};
  hipEvent_t e;
  if (c) {
    e = ln.next();
  } else {
    static thread_local hipEvent_t fallback = nullptr;      // no context: one event is enough (record/wait are ordered)
    if (!fallback) {
      hipError_t r = hipEventCreateWithFlags(&fallback, hipEventDisableTiming);
      PETR_CHECK(r == hipSuccess, PETR_ERR_LAUNCH, "ctx_join_into: hipEventCreate: %s", hipGetErrorString(r));
    }
    e = fallback;
  }
  hipError_t r = hipEventRecord(e, ln.main);
  if (r == hipSuccess) r = hipStreamWaitEvent(tgt, e, 0);
  for (int i = 0; c && i < c->n_side && r == hipSuccess; ++i) {
    hipEvent_t es = ln.next();
    r = hipEventRecord(es, c->side[i]);
    if (r == hipSuccess) r = hipStreamWaitEvent(tgt, es, 0);
  }
  PETR_CHECK(r == hipSuccess, PETR_ERR_LAUNCH, "ctx_join_into: %s", hipGetErrorString(r));
  return PETR_OK;
}

extern "C" int petr_ctx_side_stream(petr_ctx* c, int i, void** stream) {
  PETR_CHECK(c && stream && i >= 0 && i < c->n_side, PETR_ERR_INVALID, "ctx_side_stream: bad argument");
  *stream = (void*)c->side[i];
  return PETR_OK;
}

extern "C" int petr_ctx_destroy(petr_ctx* c) {
  if (!c) return PETR_OK;
  for (int i = 0; i < c->n_side; ++i) (void)hipStreamDestroy(c->side[i]);
  for (int i = 0; i < PETR_CTX_EVENTS; ++i) (void)hipEventDestroy(c->ev[i]);
  delete c;
  return PETR_OK;
}

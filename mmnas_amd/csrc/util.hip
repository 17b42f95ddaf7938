// Error plumbing, ABI version, and the small utility kernels of the C ABI:
// dropout-mask materialisation (tests / mask replay), segment pack/unpack for the data-parallel
// gradient exchange, fused Adam and sum-of-squares for the optimizer step.
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>
#include "common.h"

namespace mmnas {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return MMNAS_E_LAUNCH;
  }
  return MMNAS_OK;
}

// ---- optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline) ----
namespace prof {
struct Rec { hipEvent_t a, b; int kind; double flops, bytes; char tag[96]; };
static std::mutex mu;
static bool enabled = false;
static std::vector<Rec> recs;
static std::vector<hipEvent_t> pool;
static size_t pool_next = 0;

static hipEvent_t get_event() {
  if (pool_next == pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) e = nullptr;   // (a null event makes the launch carry no timestamp)
    pool.push_back(e);
  }
  return pool[pool_next++];
}
}  // namespace prof

bool prof_enabled() { return prof::enabled; }

struct ActiveScope { long idx; bool started; };
static thread_local ActiveScope g_active = {-1, false};

ProfScope::ProfScope(int kind, double flops, double bytes, hipStream_t st, const char* tag) : idx_(-1), st_(st) {
  if (!prof::enabled || g_active.idx >= 0) return;  // (scopes do not nest: the outer one keeps the launches)
  std::lock_guard<std::mutex> g(prof::mu);
  prof::Rec r;
  r.a = prof::get_event(); r.b = prof::get_event(); r.kind = kind; r.flops = flops; r.bytes = bytes;
  r.tag[0] = 0;
  if (tag) { strncpy(r.tag, tag, sizeof(r.tag) - 1); r.tag[sizeof(r.tag) - 1] = 0; }
  idx_ = (long)prof::recs.size();
  prof::recs.push_back(r);
  g_active.idx = idx_;
  g_active.started = false;
}

ProfScope::~ProfScope() {
  if (idx_ < 0) return;
  if (!g_active.started) {  // no kernel was launched inside the scope: drop the record
    std::lock_guard<std::mutex> g(prof::mu);
    prof::recs[idx_].kind = -1;
  }
  g_active.idx = -1;
}

ProfEvents prof_launch_events() {
  ProfEvents e = {nullptr, nullptr};
  if (g_active.idx < 0) return e;
  std::lock_guard<std::mutex> g(prof::mu);
  const prof::Rec& r = prof::recs[g_active.idx];
  e.start = g_active.started ? nullptr : r.a;
  e.stop = r.b;
  g_active.started = true;
  return e;
}

__global__ void dropout_mask_kernel(float* out, size_t n, DropCfg c) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = c.thresh ? drop_mult(c, (uint32_t)i) : 1.0f;
}

__global__ void pack_kernel(const mmnas_segment* segs, float* staging, float scale, int direction) {
  const mmnas_segment s = segs[blockIdx.y];
  float* stg = staging + s.offset;
  const size_t n4 = (((uintptr_t)s.ptr & 15) == 0 && ((uintptr_t)stg & 15) == 0) ? (s.n >> 2) : 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float4* p4 = reinterpret_cast<float4*>(s.ptr);
  float4* s4 = reinterpret_cast<float4*>(stg);
  for (size_t i = t0; i < n4; i += stride) {
    if (direction == 0) { float4 v = p4[i]; v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; s4[i] = v; }
    else { float4 v = s4[i]; v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; p4[i] = v; }
  }
  for (size_t i = (n4 << 2) + t0; i < s.n; i += stride) {
    if (direction == 0) stg[i] = s.ptr[i] * scale; else s.ptr[i] = stg[i] * scale;
  }
}

struct PackArgs { mmnas_segment s[96]; };
__global__ void pack_args_kernel(PackArgs a, float* staging, float scale, int direction) {
  const mmnas_segment s = a.s[blockIdx.y];
  float* stg = staging + s.offset;
  const size_t n4 = (((uintptr_t)s.ptr & 15) == 0 && ((uintptr_t)stg & 15) == 0) ? (s.n >> 2) : 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float4* p4 = reinterpret_cast<float4*>(s.ptr);
  float4* s4 = reinterpret_cast<float4*>(stg);
  for (size_t i = t0; i < n4; i += stride) {
    if (direction == 0) { float4 v = p4[i]; v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; s4[i] = v; }
    else { float4 v = s4[i]; v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; p4[i] = v; }
  }
  for (size_t i = (n4 << 2) + t0; i < s.n; i += stride) {
    if (direction == 0) stg[i] = s.ptr[i] * scale; else s.ptr[i] = stg[i] * scale;
  }
}

__global__ void adam_kernel(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1,
                            float b2, float eps, float wd, const float* sumsq, float max_norm, float c1, float c2) {
  float gscale = 1.f;
  if (sumsq) gscale = fminf(1.f, max_norm / (sqrtf(*sumsq) + 1e-6f));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    // torch.optim.Adam: denom = sqrt(v)/sqrt(1-b2^t) + eps ; p -= lr/(1-b1^t) * m/denom
    p[i] = pi - (lr / c1) * mi / (sqrtf(vi) / c2 + eps);
  }
}

__global__ void sumsq_kernel(const float* g, size_t n, float* out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s += g[i] * g[i];
  s = wave_sum(s);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace mmnas

using namespace mmnas;

extern "C" int mmnas_prof_enable(int on) {
  std::lock_guard<std::mutex> g(prof::mu);
  prof::enabled = on != 0;
  prof::recs.clear();
  prof::pool_next = 0;
  return MMNAS_OK;
}

extern "C" int mmnas_prof_collect(mmnas_prof_stat* stats) {
  MMNAS_REQUIRE(stats, MMNAS_E_ARG, "prof_collect: null output");
  std::lock_guard<std::mutex> g(prof::mu);
  for (int k = 0; k < MMNAS_K_COUNT; ++k) { stats[k].ms = 0; stats[k].flops = 0; stats[k].bytes = 0; stats[k].launches = 0; }
  // tuning aid: MMNAS_PROF_DUMP=<file> appends one "kind,tag,ms,flops,bytes" row per bracketed launch
  const char* dump = getenv("MMNAS_PROF_DUMP");
  FILE* df = dump && dump[0] ? fopen(dump, "a") : nullptr;
  for (const prof::Rec& r : prof::recs) {
    if (r.kind < 0) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    if (df) fprintf(df, "%d,%s,%.6f,%.0f,%.0f\n", r.kind, r.tag, ms, r.flops, r.bytes);
    mmnas_prof_stat& s = stats[r.kind];
    s.ms += ms; s.flops += r.flops; s.bytes += r.bytes; s.launches += 1;
  }
  if (df) fclose(df);
  prof::recs.clear();
  prof::pool_next = 0;
  return MMNAS_OK;
}

extern "C" int mmnas_abi_version(void) { return MMNAS_ABI_VERSION; }
extern "C" const char* mmnas_last_error(void) { return g_err; }

extern "C" int mmnas_dropout_mask(float* out, size_t n, float p, uint64_t seed, uint32_t site, void* stream) {
  MMNAS_REQUIRE(out || n == 0, MMNAS_E_ARG, "mmnas_dropout_mask: null output");
  if (n == 0) return MMNAS_OK;
  MMNAS_REQUIRE(n < (1ull << 32), MMNAS_E_SHAPE, "mmnas_dropout_mask: n too large");
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  MMNAS_LAUNCH(dropout_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, n,
                     make_drop(p, seed, site));
  return check_launch("dropout_mask");
}

extern "C" int mmnas_pack_segments(const mmnas_segment* segs, int nseg, float* staging, float scale,
                                   int direction, void* stream) {
  if (nseg <= 0) return MMNAS_OK;
  MMNAS_REQUIRE(segs && staging, MMNAS_E_ARG, "mmnas_pack_segments: null pointer");
  MMNAS_LAUNCH(pack_kernel, dim3(64, nseg), dim3(256), 0, (hipStream_t)stream, segs, staging, scale, direction);
  return check_launch("pack_segments");
}

extern "C" int mmnas_pack_segments_host(const mmnas_segment* segs_host, int nseg, float* staging, float scale,
                                        int direction, void* stream) {
  if (nseg <= 0) return MMNAS_OK;
  MMNAS_REQUIRE(segs_host && staging, MMNAS_E_ARG, "mmnas_pack_segments_host: null pointer");
  for (int base = 0; base < nseg; base += 96) {
    const int n = nseg - base < 96 ? nseg - base : 96;
    PackArgs a;
    memset(&a, 0, sizeof(a));
    memcpy(a.s, segs_host + base, (size_t)n * sizeof(mmnas_segment));
    MMNAS_LAUNCH(pack_args_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, a, staging, scale, direction);
  }
  return check_launch("pack_segments_host");
}

extern "C" int mmnas_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, const float* sumsq, float max_norm,
                               int step, void* stream) {
  if (n == 0) return MMNAS_OK;
  MMNAS_REQUIRE(p && g && m && v && step >= 1, MMNAS_E_ARG, "mmnas_adam_step: bad arguments");
  const float c1 = 1.f - powf(beta1, (float)step);
  const float c2 = sqrtf(1.f - powf(beta2, (float)step));
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  MMNAS_LAUNCH(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, sumsq, max_norm, c1, c2);
  return check_launch("adam_step");
}

extern "C" int mmnas_sumsq(const float* g, size_t n, float* out, void* stream) {
  if (n == 0) return MMNAS_OK;
  MMNAS_REQUIRE(g && out, MMNAS_E_ARG, "mmnas_sumsq: null pointer");
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  MMNAS_LAUNCH(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, out);
  return check_launch("sumsq");
}

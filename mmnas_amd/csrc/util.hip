// Error plumbing, ABI version, and the small utility kernels of the C ABI:
// dropout-mask materialisation (tests / mask replay), segment pack/unpack for the data-parallel
// gradient exchange, fused Adam and sum-of-squares for the optimizer step.
#include <stdarg.h>
#include <string.h>
#include <mutex>
#include <vector>
#include <algorithm>
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

// Gather / scatter of gradient segments (DDP's bucket copies).  One 1-D grid over all segments: segment sizes differ by
// 100x (a candidate's 1 MB of weights beside the 24 MB embedding table), so the work is cut into equal pieces of
// PACK_PIECE floats and each workgroup takes pieces round-robin; a piece never crosses a segment (pieces are counted per
// segment).  (Round 2 launched 64 workgroups per segment: the largest segment set the kernel's duration, 50 us for a
// 24 MB bucket.)
constexpr int PACK_PIECE = 8192;   // floats per piece: 8 float4 per thread
__device__ __forceinline__ void pack_piece(const mmnas_segment s, float* staging, size_t piece, float scale, int direction) {
  float* stg = staging + s.offset;
  const size_t lo = piece * PACK_PIECE, hi = lo + PACK_PIECE < s.n ? lo + PACK_PIECE : s.n;
  if ((((uintptr_t)s.ptr | (uintptr_t)stg) & 15) == 0) {
    float4* p4 = reinterpret_cast<float4*>(s.ptr);
    float4* s4 = reinterpret_cast<float4*>(stg);
    const size_t l4 = lo >> 2, h4 = hi >> 2;
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // all loads of the piece in flight before the first store
      const size_t i = l4 + threadIdx.x + 256 * k;
      if (i < h4) v[k] = direction == 0 ? p4[i] : s4[i];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t i = l4 + threadIdx.x + 256 * k;
      if (i < h4) {
        float4 w = v[k];
        w.x *= scale; w.y *= scale; w.z *= scale; w.w *= scale;
        if (direction == 0) s4[i] = w; else p4[i] = w;
      }
    }
    for (size_t i = (h4 << 2) + threadIdx.x; i < hi; i += 256) {
      if (direction == 0) stg[i] = s.ptr[i] * scale; else s.ptr[i] = stg[i] * scale;
    }
  } else {
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
      if (direction == 0) stg[i] = s.ptr[i] * scale; else s.ptr[i] = stg[i] * scale;
    }
  }
}

// piece q of the whole table -> (segment, piece inside it); nseg <= 96: a linear walk of per-segment piece counts
template <typename Table>
__device__ __forceinline__ void pack_body(const Table& segs, int nseg, size_t npieces, float* staging, float scale, int direction) {
  if (npieces == 0)   // (table on the device: the launcher could not count)
    for (int k = 0; k < nseg; ++k) npieces += (segs[k].n + PACK_PIECE - 1) / PACK_PIECE;
  for (size_t q = blockIdx.x; q < npieces; q += gridDim.x) {
    size_t rest = q;
    int k = 0;
    for (; k < nseg; ++k) {
      const size_t np = (segs[k].n + PACK_PIECE - 1) / PACK_PIECE;
      if (rest < np) break;
      rest -= np;
    }
    if (k < nseg) pack_piece(segs[k], staging, rest, scale, direction);
  }
}

__global__ void pack_kernel(const mmnas_segment* segs, int nseg, size_t npieces, float* staging, float scale, int direction) {
  pack_body(segs, nseg, npieces, staging, scale, direction);
}

struct PackArgs { mmnas_segment s[96]; };
__global__ void pack_args_kernel(PackArgs a, int nseg, size_t npieces, float* staging, float scale, int direction) {
  pack_body(a.s, nseg, npieces, staging, scale, direction);
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

// the same arithmetic on four consecutive elements per thread and pass: 16-byte loads / stores (the scalar form moved
// 4 bytes per lane and instruction: 0.45 ms for the supernet's 37 M parameters where 1.04 GB of traffic is 0.21 ms at 5 TB/s)
__device__ __forceinline__ void adam_one(float& pi, float gi, float& mi, float& vi, float gscale, float lr, float b1, float b2,
                                         float eps, float wd, float c1, float c2) {
  gi *= gscale;
  if (wd != 0.f) gi += wd * pi;
  mi = b1 * mi + (1.f - b1) * gi;
  vi = b2 * vi + (1.f - b2) * gi * gi;
  pi = pi - (lr / c1) * mi / (sqrtf(vi) / c2 + eps);
}
__global__ void __launch_bounds__(256) adam4_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                    float4* __restrict__ v, size_t n4, float lr, float b1, float b2, float eps,
                                                    float wd, const float* sumsq, float max_norm, float c1, float c2) {
  float gscale = 1.f;
  if (sumsq) gscale = fminf(1.f, max_norm / (sqrtf(*sumsq) + 1e-6f));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 pi = p[i], mi = m[i], vi = v[i];
    const float4 gi = g[i];
    adam_one(pi.x, gi.x, mi.x, vi.x, gscale, lr, b1, b2, eps, wd, c1, c2);
    adam_one(pi.y, gi.y, mi.y, vi.y, gscale, lr, b1, b2, eps, wd, c1, c2);
    adam_one(pi.z, gi.z, mi.z, vi.z, gscale, lr, b1, b2, eps, wd, c1, c2);
    adam_one(pi.w, gi.w, mi.w, vi.w, gscale, lr, b1, b2, eps, wd, c1, c2);
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
}
__global__ void __launch_bounds__(256) sumsq4_kernel(const float4* __restrict__ g, size_t n4, float* out) {
  float s0 = 0.f, s1 = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x = g[i];
    s0 += x.x * x.x + x.y * x.y; s1 += x.z * x.z + x.w * x.w;
  }
  float s = wave_sum(s0 + s1);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
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

static int pack_grid(size_t npieces) { return (int)(npieces < 2048 ? (npieces ? npieces : 1) : 2048); }

extern "C" int mmnas_pack_segments(const mmnas_segment* segs, int nseg, float* staging, float scale,
                                   int direction, void* stream) {
  if (nseg <= 0) return MMNAS_OK;
  MMNAS_REQUIRE(segs && staging, MMNAS_E_ARG, "mmnas_pack_segments: null pointer");
  // (the table lives on the device: the kernel counts the pieces itself)
  MMNAS_LAUNCH(pack_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, segs, nseg, (size_t)0, staging, scale, direction);
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
    size_t npieces = 0;
    for (int k = 0; k < n; ++k) npieces += (a.s[k].n + PACK_PIECE - 1) / PACK_PIECE;
    MMNAS_LAUNCH(pack_args_kernel, dim3(pack_grid(npieces)), dim3(256), 0, (hipStream_t)stream, a, n, npieces, staging, scale, direction);
  }
  return check_launch("pack_segments_host");
}

// ---- ragged batches: padded [B, S, d] <-> packed [sum n_b, d] rows (sample b: its first n_b = off[b+1] - off[b] rows) ----
namespace mmnas {
template <bool PACK>
__global__ void __launch_bounds__(256) ragged_rows_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                          const int* __restrict__ off, int B, int S, int d4) {
  const size_t n = (size_t)B * S * d4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t row = i / d4;
    const int c = (int)(i - row * d4);
    const int b = (int)(row / S), s = (int)(row - (size_t)b * S);
    const int o = off[b], nb = off[b + 1] - o;
    if (PACK) {
      if (s < nb) dst[(size_t)(o + s) * d4 + c] = src[i];
    } else {
      dst[i] = s < nb ? src[(size_t)(o + s) * d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);   // padding rows: zero
    }
  }
}
}  // namespace mmnas

extern "C" int mmnas_pack_rows(const float* x, const int* off, float* packed, int B, int S, int d, void* stream) {
  MMNAS_REQUIRE(x && off && packed && B > 0 && S > 0 && d > 0 && d % 4 == 0, MMNAS_E_ARG, "pack_rows: bad arguments (d %% 4 == 0)");
  const size_t n = (size_t)B * S * (d / 4);
  MMNAS_LAUNCH((mmnas::ragged_rows_kernel<true>), dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
               (const float4*)x, (float4*)packed, off, B, S, d / 4);
  return mmnas::check_launch("pack_rows");
}

extern "C" int mmnas_unpack_rows(const float* packed, const int* off, float* x, int B, int S, int d, void* stream) {
  MMNAS_REQUIRE(x && off && packed && B > 0 && S > 0 && d > 0 && d % 4 == 0, MMNAS_E_ARG, "unpack_rows: bad arguments (d %% 4 == 0)");
  const size_t n = (size_t)B * S * (d / 4);
  MMNAS_LAUNCH((mmnas::ragged_rows_kernel<false>), dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream,
               (const float4*)packed, (float4*)x, off, B, S, d / 4);
  return mmnas::check_launch("unpack_rows");
}

extern "C" int mmnas_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, const float* sumsq, float max_norm,
                               int step, void* stream) {
  if (n == 0) return MMNAS_OK;
  MMNAS_REQUIRE(p && g && m && v && step >= 1, MMNAS_E_ARG, "mmnas_adam_step: bad arguments");
  const float c1 = 1.f - powf(beta1, (float)step);
  const float c2 = sqrtf(1.f - powf(beta2, (float)step));
  // 16-byte path for the aligned body, the scalar kernel for a misaligned call or the tail (< 4 elements)
  const bool al16 = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  const size_t n4 = al16 ? n / 4 : 0, done = 4 * n4;
  if (n4) {
    const int blocks4 = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    MMNAS_LAUNCH(adam4_kernel, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, (float4*)p, (const float4*)g, (float4*)m, (float4*)v, n4,
                 lr, beta1, beta2, eps, weight_decay, sumsq, max_norm, c1, c2);
  }
  if (done < n) {
    const size_t r = n - done;
    const int blocks = (int)((r + 255) / 256 < 4096 ? (r + 255) / 256 : 4096);
    MMNAS_LAUNCH(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p + done, g + done, m + done, v + done, r, lr, beta1,
                 beta2, eps, weight_decay, sumsq, max_norm, c1, c2);
  }
  return check_launch("adam_step");
}

extern "C" int mmnas_sumsq(const float* g, size_t n, float* out, void* stream) {
  if (n == 0) return MMNAS_OK;
  MMNAS_REQUIRE(g && out, MMNAS_E_ARG, "mmnas_sumsq: null pointer");
  const size_t n4 = (((uintptr_t)g & 15) == 0) ? n / 4 : 0, done = 4 * n4;
  if (n4) {
    const int blocks4 = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
    MMNAS_LAUNCH(sumsq4_kernel, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, (const float4*)g, n4, out);
  }
  if (done < n) {
    const size_t r = n - done;
    const int blocks = (int)((r + 255) / 256 < 1024 ? (r + 255) / 256 : 1024);
    MMNAS_LAUNCH(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g + done, r, out);
  }
  return check_launch("sumsq");
}

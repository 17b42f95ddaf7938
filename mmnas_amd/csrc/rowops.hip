// Row-wise and element-wise kernels of the operator epilogues: the reference's hand-written
// LayerNorm (modules.py:44-56: Bessel-corrected std, eps added to the std), bias-gradient column
// sums, the activation registry entries (modules.py:96-119, ops_adapter.py:25-29) and nn.GLU.
// All are HBM-bound: one pass over the data, 16-byte accesses, one wave per row.
#include "common.h"

namespace mmnas {

// ---------------------------------------------------------------- LayerNorm forward
template <int NV>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                     const float* __restrict__ b, float* __restrict__ y,
                                                     int M, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * d;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    v[i] = (c < d) ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)d;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < d) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
  }
  const float sd = sqrtf(wave_sum(ss) / (float)(d - 1));
  const float inv = 1.0f / (sd + eps);
  float* yr = y + (size_t)row * d;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < d) {
      const float4 av = *reinterpret_cast<const float4*>(a + c);
      const float4 bv = *reinterpret_cast<const float4*>(b + c);
      float4 o;
      o.x = av.x * v[i].x * inv + bv.x; o.y = av.y * v[i].y * inv + bv.y;
      o.z = av.z * v[i].z * inv + bv.z; o.w = av.w * v[i].w * inv + bv.w;
      *reinterpret_cast<float4*>(yr + c) = o;
    }
  }
}

// ---------------------------------------------------------------- LayerNorm backward
// dx = (g - mean(g))/s - c * sum(g*c) / ((n-1) * sd * s^2),  g = dy*a, c = x-mean, s = sd+eps
// (pinned by oracle.layer_norm_backward against the reference's autograd).
template <int NV>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                     const float* __restrict__ dy, float* __restrict__ dx,
                                                     float* __restrict__ da, float* __restrict__ db,
                                                     float* __restrict__ ddrop, float* __restrict__ dcol,
                                                     float* __restrict__ part, DropCfg drop, int M, int d,
                                                     float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 acc_a[NV], acc_b[NV], acc_c[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    acc_a[i] = make_float4(0.f, 0.f, 0.f, 0.f); acc_b[i] = acc_a[i]; acc_c[i] = acc_a[i];
  }
  float4 av[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    av[i] = (c < d) ? *reinterpret_cast<const float4*>(a + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // a wave walks its rows with the NEXT row's loads in flight during the current row's three wave reductions
  // (one wave per SIMD and no prefetch left every row's ~2 us of load latency exposed: 15-20 us per launch)
  auto fetch = [&](int row, float4* vv, float4* gg) {
    const int rc = row < M ? row : M - 1;   // clamped: the loads are unconditional
    const float* xr = x + (size_t)rc * d;
    const float* gr = dy + (size_t)rc * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < d) {
        vv[i] = *reinterpret_cast<const float4*>(xr + c);
        gg[i] = *reinterpret_cast<const float4*>(gr + c);
      } else {
        vv[i] = make_float4(0.f, 0.f, 0.f, 0.f); gg[i] = vv[i];
      }
    }
  };
  float4 vn[NV], gn[NV];
  const int row0 = blockIdx.x * 4 + wave, rstep = gridDim.x * 4;
  fetch(row0, vn, gn);
  for (int row = row0; row < M; row += rstep) {
    float4 v[NV], g[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i] = vn[i]; g[i] = gn[i]; }
    fetch(row + rstep, vn, gn);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) / (float)d;
    float ss = 0.f, sg = 0.f, sgc = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < d) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        const float gx = g[i].x * av[i].x, gy = g[i].y * av[i].y, gz = g[i].z * av[i].z, gw = g[i].w * av[i].w;
        sg += (gx + gy) + (gz + gw);
        sgc += (gx * v[i].x + gy * v[i].y) + (gz * v[i].z + gw * v[i].w);
      }
    }
    ss = wave_sum(ss); sg = wave_sum(sg); sgc = wave_sum(sgc);
    const float sd = sqrtf(ss / (float)(d - 1));
    const float sden = sd + eps;
    const float inv = 1.0f / sden;
    const float mg = sg / (float)d;
    const float k2 = sgc / ((float)(d - 1) * sd * sden * sden);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < d) {
        float4 o;
        o.x = (g[i].x * av[i].x - mg) * inv - v[i].x * k2;
        o.y = (g[i].y * av[i].y - mg) * inv - v[i].y * k2;
        o.z = (g[i].z * av[i].z - mg) * inv - v[i].z * k2;
        o.w = (g[i].w * av[i].w - mg) * inv - v[i].w * k2;
        *reinterpret_cast<float4*>(dx + (size_t)row * d + c) = o;
        acc_a[i].x += g[i].x * v[i].x * inv; acc_a[i].y += g[i].y * v[i].y * inv;
        acc_a[i].z += g[i].z * v[i].z * inv; acc_a[i].w += g[i].w * v[i].w * inv;
        acc_b[i].x += g[i].x; acc_b[i].y += g[i].y; acc_b[i].z += g[i].z; acc_b[i].w += g[i].w;
        if (ddrop) {
          if (drop.thresh) {
            const uint32_t base = (uint32_t)row * (uint32_t)d + (uint32_t)c;
            o.x *= drop_mult(drop, base); o.y *= drop_mult(drop, base + 1);
            o.z *= drop_mult(drop, base + 2); o.w *= drop_mult(drop, base + 3);
          }
          *reinterpret_cast<float4*>(ddrop + (size_t)row * d + c) = o;
          acc_c[i].x += o.x; acc_c[i].y += o.y; acc_c[i].z += o.z; acc_c[i].w += o.w;
        }
      }
    }
  }
  // The per-wave column partials meet in LDS as 16-byte rows; thread (which = wave < 3, lane) adds the 4 partials of its 4
  // columns with 16-byte reads (conflict-free) and leaves one 16-byte partial-row store (or 4 atomics).  Rounds 1-5 had wave
  // 0 read them 4 bytes per lane at a stride of 16 bytes -- a 4-way bank conflict on every read (SQ_LDS_BANK_CONFLICT /
  // SQ_LDS_IDX_ACTIVE = 0.49) -- and write 12 scalars per lane.
  __shared__ __attribute__((aligned(16))) float red[3][4][64 * 4];  // [which][wave][lane*4 + j], reused per i
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    __syncthreads();
    *reinterpret_cast<float4*>(&red[0][wave][lane * 4]) = acc_a[i];
    *reinterpret_cast<float4*>(&red[1][wave][lane * 4]) = acc_b[i];
    *reinterpret_cast<float4*>(&red[2][wave][lane * 4]) = acc_c[i];
    __syncthreads();
    if (wave < 3 && c < d) {
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(&red[wave][0][lane * 4]);
      const f32x4 p1 = *reinterpret_cast<const f32x4*>(&red[wave][1][lane * 4]);
      const f32x4 p2 = *reinterpret_cast<const f32x4*>(&red[wave][2][lane * 4]);
      const f32x4 p3 = *reinterpret_cast<const f32x4*>(&red[wave][3][lane * 4]);
      const f32x4 s4 = (p0 + p1) + (p2 + p3);
      if (part) {  // per-block partial rows, summed by ln_bwd_reduce_kernel / the next pair launch (no atomics, deterministic)
        *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.x * 3 + wave) * d + c) = s4;
      } else {
        float* out = wave == 0 ? da : (wave == 1 ? db : dcol);
        if (out) {
          atomicAdd(out + c, s4.x); atomicAdd(out + c + 1, s4.y); atomicAdd(out + c + 2, s4.z); atomicAdd(out + c + 3, s4.w);
        }
      }
    }
  }
}

// out_w[c] += sum over blocks of part[block][w][c]; grid (ceil(d/16), 3), block 1024 = 16 columns x 64
// block-slices: every partial row is touched by one lane group, 4 independent loads in flight per thread
__global__ void __launch_bounds__(1024) ln_bwd_reduce_kernel(const float* __restrict__ part, int nblocks, int d,
                                                             float* __restrict__ da, float* __restrict__ db,
                                                             float* __restrict__ dcol) {
  const int w = blockIdx.y;
  float* out = w == 0 ? da : (w == 1 ? db : dcol);
  if (!out) return;
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;  // g in [0,64)
  const int c = blockIdx.x * 16 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < d) {
    int b = g;
    for (; b + 192 < nblocks; b += 256) {
      s0 += part[((size_t)b * 3 + w) * d + c];
      s1 += part[((size_t)(b + 64) * 3 + w) * d + c];
      s2 += part[((size_t)(b + 128) * 3 + w) * d + c];
      s3 += part[((size_t)(b + 192) * 3 + w) * d + c];
    }
    for (; b < nblocks; b += 64) s0 += part[((size_t)b * 3 + w) * d + c];
  }
  __shared__ float red[64][17];
  red[g][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && c < d) {
    float t = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) t += red[i][cl];
    out[c] += t;
  }
}

// ---------------------------------------------------------------- column sums
__global__ void __launch_bounds__(256) colsum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                     int M, int N, int ldx, int rows_per_block) {
  // block = 64 columns x 4 row-slices; 8 independent loads in flight per thread, LDS reduce of the
  // 4 slices, one atomic per column per block
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.f;
  if (col < N) {
    int r = r0 + g;
    for (; r + 28 < r1; r += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += x[(size_t)(r + 4 * u) * ldx + col];
    }
    for (; r < r1; r += 4) s[0] += x[(size_t)r * ldx + col];
  }
  __shared__ float red[4][64];
  red[g][cl] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (g == 0 && col < N) atomicAdd(out + col, (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]));
}

// ---------------------------------------------------------------- element-wise
__device__ __forceinline__ float act_fwd(int kind, float x) {
  switch (kind) {
    case 0: return x * 0.f;
    case 1: return fmaxf(x, 0.f);
    case 2: return x > 0.f ? x : 0.01f * x;
    default: {
      const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
      return 0.5f * x * (1.f + tanhf(u));
    }
  }
}
__device__ __forceinline__ float act_bwd(int kind, float x) {
  switch (kind) {
    case 0: return 0.f;
    case 1: return x > 0.f ? 1.f : 0.f;
    case 2: return x > 0.f ? 1.f : 0.01f;
    default: {
      const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
      const float t = tanhf(u);
      return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * 0.7978845608028654f * (1.f + 3.f * 0.044715f * x * x);
    }
  }
}

__global__ void eltwise_fwd_kernel(int kind, const float* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = act_fwd(kind, x[i]);
}
__global__ void eltwise_bwd_kernel(int kind, const float* __restrict__ x, const float* __restrict__ dy,
                                   float* __restrict__ dx, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dx[i] = dy[i] * act_bwd(kind, x[i]);
}
__global__ void drop_add_kernel(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y,
                                size_t n, DropCfg drop) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v = x[i];
    if (drop.thresh) v *= drop_mult(drop, (uint32_t)i);
    y[i] = res ? res[i] + v : v;
  }
}

// y = sum_j src[j] (n <= ADD_MANY_MAX sources, fixed order): the key / value source gradients of a chain's guided operators
struct AddManyK { const float* src[ADD_MANY_MAX]; int n; };
__global__ void add_many_kernel(const AddManyK p, float* __restrict__ y, size_t count4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += (size_t)gridDim.x * blockDim.x) {
    float4 acc = reinterpret_cast<const float4*>(p.src[0])[i];
    for (int j = 1; j < p.n; ++j) {
      const float4 v = reinterpret_cast<const float4*>(p.src[j])[i];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4*>(y)[i] = acc;
  }
}

__global__ void glu_fwd_kernel(const float* __restrict__ h, float* __restrict__ y, int M, int C, int relu,
                               DropCfg drop) {
  const size_t n = (size_t)M * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / C, c = i - m * C;
    const float a = h[m * 2 * C + c], b = h[m * 2 * C + C + c];
    float v = a / (1.f + expf(-b));
    if (relu) v = fmaxf(v, 0.f);
    if (drop.thresh) v *= drop_mult(drop, (uint32_t)i);
    y[i] = v;
  }
}
__global__ void glu_bwd_kernel(const float* __restrict__ h, const float* __restrict__ dy, float* __restrict__ dh,
                               int M, int C, int relu, DropCfg drop) {
  const size_t n = (size_t)M * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / C, c = i - m * C;
    const float a = h[m * 2 * C + c], b = h[m * 2 * C + C + c];
    const float sg = 1.f / (1.f + expf(-b));
    float g = dy[i];
    if (drop.thresh) g *= drop_mult(drop, (uint32_t)i);
    if (relu && !(a * sg > 0.f)) g = 0.f;
    dh[m * 2 * C + c] = g * sg;
    dh[m * 2 * C + C + c] = g * a * sg * (1.f - sg);
  }
}

static inline int blocks_for(size_t n) { size_t b = (n + 255) / 256; return (int)(b < 2048 ? (b ? b : 1) : 2048); }

}  // namespace mmnas

using namespace mmnas;

extern "C" int mmnas_layernorm_fwd(const float* x, const float* a, const float* b, float* y, int M, int d,
                                   float eps, void* stream) {
  MMNAS_REQUIRE(x && a && b && y, MMNAS_E_ARG, "layernorm_fwd: null pointer");
  MMNAS_REQUIRE(M > 0 && d >= 4 && d % 4 == 0 && d <= 2048, MMNAS_E_SHAPE,
                "layernorm_fwd: M=%d d=%d (need d %% 4 == 0, 4 <= d <= 2048)", M, d);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(cdiv(M, 4)), block(256);
  const int nv = cdiv(d, 256);
  ProfScope ps(MMNAS_K_ROWOPS, 8.0 * M * d, 8.0 * M * d, st);
  if (nv <= 1) MMNAS_LAUNCH(ln_fwd_kernel<1>, grid, block, 0, st, x, a, b, y, M, d, eps);
  else if (nv <= 2) MMNAS_LAUNCH(ln_fwd_kernel<2>, grid, block, 0, st, x, a, b, y, M, d, eps);
  else if (nv <= 4) MMNAS_LAUNCH(ln_fwd_kernel<4>, grid, block, 0, st, x, a, b, y, M, d, eps);
  else MMNAS_LAUNCH(ln_fwd_kernel<8>, grid, block, 0, st, x, a, b, y, M, d, eps);
  return check_launch("layernorm_fwd");
}

static int ln_bwd_blocks(int M) {   // <= 2 workgroups (8 waves) per CU; each wave prefetches its next row
  int nb = cdiv(M, 4);
  return nb > 512 ? 512 : nb;   // (256..1024 measured within 5 %: the kernel moves 52 MB at ~3 TB/s, read+write mix)
}

extern "C" size_t mmnas_layernorm_bwd_ws_floats(int M, int d) { return (size_t)ln_bwd_blocks(M) * 3 * d; }

namespace mmnas {
int layernorm_bwd_deferred(const float* x, const float* a, const float* dy, float* dx, float* da, float* db, float* ddrop,
                           float* dcol, float* ws, float drop_p, uint64_t seed, uint32_t site, int M, int d, float eps,
                           hipStream_t st, AuxReduce* aux) {
  MMNAS_REQUIRE(x && a && dy && dx, MMNAS_E_ARG, "layernorm_bwd: null pointer");
  MMNAS_REQUIRE(M > 0 && d >= 4 && d % 4 == 0 && d <= 2048, MMNAS_E_SHAPE,
                "layernorm_bwd: M=%d d=%d (need d %% 4 == 0, 4 <= d <= 2048)", M, d);
  MMNAS_REQUIRE(dcol == nullptr || ddrop != nullptr, MMNAS_E_ARG, "layernorm_bwd: dcol needs ddrop");
  const int nb = ln_bwd_blocks(M);
  dim3 grid(nb), block(256);
  const DropCfg dc = make_drop(drop_p, seed, site);
  const int nv = cdiv(d, 256);
  ProfScope ps(MMNAS_K_ROWOPS, 16.0 * M * d, 4.0 * M * d * (ddrop ? 4.0 : 3.0), st);
#define LNB(NV) MMNAS_LAUNCH(ln_bwd_kernel<NV>, grid, block, 0, st, x, a, dy, dx, da, db, ddrop, dcol, ws, dc, M, d, eps)
  if (nv <= 1) LNB(1); else if (nv <= 2) LNB(2); else if (nv <= 4) LNB(4); else LNB(8);
#undef LNB
  aux->part = ws; aux->nrows = nb; aux->d = d;
  aux->out[0] = da; aux->out[1] = db; aux->out[2] = dcol;
  if (!ws || (!da && !db && !dcol)) aux->part = nullptr;
  return check_launch("layernorm_bwd");
}

int launch_aux_reduce(const AuxReduce& a, hipStream_t st) {
  if (!a.part) return MMNAS_OK;
  ProfScope ps(MMNAS_K_ROWOPS, 0.0, 4.0 * a.nrows * 3.0 * a.d, st);
  MMNAS_LAUNCH(ln_bwd_reduce_kernel, dim3(cdiv(a.d, 16), 3), dim3(1024), 0, st, a.part, a.nrows, a.d, a.out[0], a.out[1], a.out[2]);
  return check_launch("layernorm_bwd_reduce");
}
}  // namespace mmnas

extern "C" int mmnas_layernorm_bwd(const float* x, const float* a, const float* dy, float* dx, float* da,
                                   float* db, float* ddrop, float* dcol, float* ws, float drop_p, uint64_t seed,
                                   uint32_t site, int M, int d, float eps, void* stream) {
  AuxReduce aux;
  const int rc = layernorm_bwd_deferred(x, a, dy, dx, da, db, ddrop, dcol, ws, drop_p, seed, site, M, d, eps, (hipStream_t)stream, &aux);
  if (rc) return rc;
  return launch_aux_reduce(aux, (hipStream_t)stream);
}

extern "C" int mmnas_colsum(const float* x, float* out, int M, int N, int ldx, void* stream) {
  MMNAS_REQUIRE(x && out && M > 0 && N > 0 && ldx >= N, MMNAS_E_ARG, "colsum: bad arguments");
  int splits = cdiv(M, 64);
  const int colblocks = cdiv(N, 64);
  const int want = cdiv(1024, colblocks);   // ~4 workgroups per CU in total
  if (splits > want) splits = want;
  if (splits < 1) splits = 1;
  const int rpb = cdiv(M, splits);
  MMNAS_LAUNCH(colsum_kernel, dim3(colblocks, cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, x, out, M,
                     N, ldx, rpb);
  return check_launch("colsum");
}

extern "C" int mmnas_eltwise_fwd(int kind, const float* x, float* y, size_t n, void* stream) {
  MMNAS_REQUIRE(kind >= 0 && kind <= 3 && x && y, MMNAS_E_ARG, "eltwise_fwd: bad arguments");
  if (n == 0) return MMNAS_OK;
  MMNAS_LAUNCH(eltwise_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, kind, x, y, n);
  return check_launch("eltwise_fwd");
}
extern "C" int mmnas_eltwise_bwd(int kind, const float* x, const float* dy, float* dx, size_t n, void* stream) {
  MMNAS_REQUIRE(kind >= 0 && kind <= 3 && x && dy && dx, MMNAS_E_ARG, "eltwise_bwd: bad arguments");
  if (n == 0) return MMNAS_OK;
  MMNAS_LAUNCH(eltwise_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, kind, x, dy, dx, n);
  return check_launch("eltwise_bwd");
}
extern "C" int mmnas_drop_add(const float* x, const float* res, float* y, size_t n, float drop_p, uint64_t seed,
                              uint32_t site, void* stream) {
  MMNAS_REQUIRE(x && y && n < (1ull << 32), MMNAS_E_ARG, "drop_add: bad arguments");
  if (n == 0) return MMNAS_OK;
  MMNAS_LAUNCH(drop_add_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, res, y, n,
                     make_drop(drop_p, seed, site));
  return check_launch("drop_add");
}
namespace mmnas {
int add_many(const float* const* srcs, int n, float* y, size_t count, hipStream_t st) {
  MMNAS_REQUIRE(srcs && y && n >= 1 && n <= ADD_MANY_MAX && count % 4 == 0, MMNAS_E_ARG, "add_many: %d sources, %zu elements", n, count);
  AddManyK k;
  k.n = n;
  for (int j = 0; j < n; ++j) { MMNAS_REQUIRE(srcs[j], MMNAS_E_ARG, "add_many: null source %d", j); k.src[j] = srcs[j]; }
  if (count == 0) return MMNAS_OK;
  ProfScope ps(MMNAS_K_ROWOPS, 0.0, 4.0 * (double)count * (n + 1), st);
  MMNAS_LAUNCH(add_many_kernel, dim3(blocks_for(count / 4)), dim3(256), 0, st, k, y, count / 4);
  return check_launch("add_many");
}
}  // namespace mmnas

extern "C" int mmnas_glu_fwd(const float* h, float* y, int M, int C, int relu, float drop_p, uint64_t seed,
                             uint32_t site, void* stream) {
  MMNAS_REQUIRE(h && y && M > 0 && C > 0, MMNAS_E_ARG, "glu_fwd: bad arguments");
  MMNAS_LAUNCH(glu_fwd_kernel, dim3(blocks_for((size_t)M * C)), dim3(256), 0, (hipStream_t)stream, h, y, M, C,
                     relu, make_drop(drop_p, seed, site));
  return check_launch("glu_fwd");
}
extern "C" int mmnas_glu_bwd(const float* h, const float* dy, float* dh, int M, int C, int relu, float drop_p,
                             uint64_t seed, uint32_t site, void* stream) {
  MMNAS_REQUIRE(h && dy && dh && M > 0 && C > 0, MMNAS_E_ARG, "glu_bwd: bad arguments");
  MMNAS_LAUNCH(glu_bwd_kernel, dim3(blocks_for((size_t)M * C)), dim3(256), 0, (hipStream_t)stream, h, dy, dh, M,
                     C, relu, make_drop(drop_p, seed, site));
  return check_launch("glu_bwd");
}

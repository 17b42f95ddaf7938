// Supernet node plumbing of the architecture step (MixedOp, mmnas/model/mixed.py):
//   * the gated sum of a node's candidate outputs (mixed.py:59-68) and its backward -- the gradient of every gate is
//     the inner product <dL/dout, o_j>, which the reference obtains from ~10 ATen kernels per candidate
//     (select, mul, add, and their autograd: expand, mul, sum);
//   * the architecture-parameter update of all nodes in one launch: dL/dalpha from dL/dgate (mixed.py:171-198,
//     'full' mode) followed by the Adam step of alpha_optim (search_vqa.py:194,331-332).
// HBM-bound streaming kernels (one pass over the candidate outputs).
#include <string.h>
#include "common.h"

namespace mmnas {

constexpr int MAXC = MMNAS_MIXED_MAX;
#define MMNAS_MIX_REDUCE_MAX 32
static_assert(MAXC == 8, "mixed_sum_reduce_kernel assumes 8 candidate slots");

struct MixArgs {
  const float* o[MAXC];
  int n;
};

__global__ void __launch_bounds__(256) mixed_sum_fwd_kernel(MixArgs a, const float* __restrict__ gate, float* __restrict__ out,
                                                            size_t n4) {
  float g[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) g[j] = j < a.n ? gate[j] : 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // every candidate's loads of an element are issued before the first is used (the compiler hoists them out of
  // the unrolled loop): n independent 16-byte loads in flight per thread, one pass over each tensor
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      v[j] = (j < a.n && a.o[j]) ? reinterpret_cast<const float4*>(a.o[j])[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < MAXC; ++j) { acc.x += g[j] * v[j].x; acc.y += g[j] * v[j].y; acc.z += g[j] * v[j].z; acc.w += g[j] * v[j].w; }
    reinterpret_cast<float4*>(out)[i] = acc;
  }
}

// partial inner products <dout, o_j> of this workgroup's slice -> part[blockIdx.x][MAXC]; d_active = gate[active] * dout
__global__ void __launch_bounds__(256) mixed_sum_bwd_kernel(MixArgs a, const float* __restrict__ gate, const float* __restrict__ dout,
                                                            float* __restrict__ d_active, int active, float* __restrict__ part,
                                                            size_t n4) {
  __shared__ float red[4][MAXC];
  float s[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) s[j] = 0.f;
  const float ga = d_active ? gate[active] : 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 d = reinterpret_cast<const float4*>(dout)[i];
    float4 v[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      v[j] = (j < a.n && a.o[j]) ? reinterpret_cast<const float4*>(a.o[j])[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < MAXC; ++j) s[j] += (d.x * v[j].x + d.y * v[j].y) + (d.z * v[j].z + d.w * v[j].w);
    if (d_active) reinterpret_cast<float4*>(d_active)[i] = make_float4(ga * d.x, ga * d.y, ga * d.z, ga * d.w);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const float t = wave_sum(s[j]);
    if (lane == 0) red[wave][j] = t;
  }
  __syncthreads();
  if (threadIdx.x < MAXC)
    part[(size_t)blockIdx.x * MAXC + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// dgate[j] += sum over workgroups, in a fixed order (bitwise reproducible): 256 threads = 8 candidates x 32 strided
// partial sums, then a fixed-order tree through LDS.  (A single thread per candidate walking all 2048 partials was a
// chain of 2048 dependent loads: 105 us per node, a fifth of the architecture step.)
__global__ void __launch_bounds__(256) mixed_sum_reduce_kernel(const float* __restrict__ part, int nwg, int n, float* __restrict__ dgate) {
  __shared__ float red[32][MAXC + 1];
  const int j = threadIdx.x & (MAXC - 1), g = threadIdx.x / MAXC;   // MAXC = 8: 32 groups
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  int b = g;
  for (; b + 96 < nwg; b += 128) {
    t0 += part[(size_t)b * MAXC + j];
    t1 += part[(size_t)(b + 32) * MAXC + j];
    t2 += part[(size_t)(b + 64) * MAXC + j];
    t3 += part[(size_t)(b + 96) * MAXC + j];
  }
  for (; b < nwg; b += 32) t0 += part[(size_t)b * MAXC + j];
  red[g][j] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (threadIdx.x < n) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red[i][threadIdx.x];
    dgate[threadIdx.x] += t;
  }
}

// The same reduction for MANY nodes in one launch (the backbone chain of the architecture step defers its 30 nodes'
// reductions to the end of its backward: one workgroup per node instead of 30 one-workgroup launches of ~5 us each).
struct MixReduceJobs {
  const float* part[MMNAS_MIX_REDUCE_MAX];
  float* dgate[MMNAS_MIX_REDUCE_MAX];
  int nwg[MMNAS_MIX_REDUCE_MAX];
  int n[MMNAS_MIX_REDUCE_MAX];
};
__global__ void __launch_bounds__(256) mixed_sum_reduce_many_kernel(const MixReduceJobs jobs) {
  __shared__ float red[32][MAXC + 1];
  const float* __restrict__ part = jobs.part[blockIdx.x];
  float* __restrict__ dgate = jobs.dgate[blockIdx.x];
  const int nwg = jobs.nwg[blockIdx.x], n = jobs.n[blockIdx.x];
  const int j = threadIdx.x & (MAXC - 1), g = threadIdx.x / MAXC;
  float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
  int b = g;
  for (; b + 96 < nwg; b += 128) {
    t0 += part[(size_t)b * MAXC + j];
    t1 += part[(size_t)(b + 32) * MAXC + j];
    t2 += part[(size_t)(b + 64) * MAXC + j];
    t3 += part[(size_t)(b + 96) * MAXC + j];
  }
  for (; b < nwg; b += 32) t0 += part[(size_t)b * MAXC + j];
  red[g][j] = (t0 + t1) + (t2 + t3);
  __syncthreads();
  if (threadIdx.x < n) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) t += red[i][threadIdx.x];
    dgate[threadIdx.x] += t;
  }
}

// ---- node epilogue of the architecture step: every candidate's LayerNorm AND the gated sum in one pass ----
// A supernet node in modes 'full' / 'two' evaluates n candidates on the same input; each ends in its own LayerNorm
// (modules.py:52-56, wrapper :266-268) and the node's output is sum_j gate_j LN_j(z_j) (mixed.py:59-68).  As separate
// launches that is n LayerNorm kernels (read z_j, write y_j) plus the gated sum (read every y_j, write out): 2n + 1
// passes over [M, d] and n + 1 launches per node, 30 nodes per step.  Here: one wave per row reads the n pre-LayerNorm
// rows, normalises each in registers and writes only the node output -- n + 1 passes, one launch; the candidates' own
// outputs y_j are never stored.  The backward needs them once more for the gate gradients <dout, y_j>: it recomputes
// them from z_j the same way.  A candidate without LayerNorm (or one whose kernel normalised already) passes
// ln_a = NULL and its output as z.
struct NodeMixArgs {
  const float* z[MAXC];
  const float* a[MAXC];
  const float* b[MAXC];
  int n;
};

template <int NV>
__device__ __forceinline__ void node_ln_row(float4 (&v)[NV], const float* __restrict__ a, const float* __restrict__ b, int lane, int d,
                                            float eps) {
  // (the arithmetic of ln_fwd_kernel, rowops.hip, operation for operation)
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
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
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < d) {
      const float4 av = *reinterpret_cast<const float4*>(a + c);
      const float4 bv = *reinterpret_cast<const float4*>(b + c);
      v[i].x = av.x * v[i].x * inv + bv.x; v[i].y = av.y * v[i].y * inv + bv.y;
      v[i].z = av.z * v[i].z * inv + bv.z; v[i].w = av.w * v[i].w * inv + bv.w;
    }
  }
}

template <int NV>
__global__ void __launch_bounds__(256) node_mix_fwd_kernel(NodeMixArgs p, const float* __restrict__ gate, float* __restrict__ out,
                                                           int M, int d, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float4 acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v[MAXC][NV];
#pragma unroll
  for (int j = 0; j < MAXC; ++j)     // every candidate's row in flight before the first reduction
    if (j < p.n && p.z[j]) {         // (z == NULL: a candidate of the node that was not evaluated -- mode 'two')
      const float* zr = p.z[j] + (size_t)row * d;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        v[j][i] = (c < d) ? *reinterpret_cast<const float4*>(zr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
  for (int j = 0; j < MAXC; ++j)
    if (j < p.n && p.z[j]) {
      if (p.a[j]) node_ln_row<NV>(v[j], p.a[j], p.b[j], lane, d, eps);
      const float g = gate[j];
#pragma unroll
      for (int i = 0; i < NV; ++i) { acc[i].x += g * v[j][i].x; acc[i].y += g * v[j][i].y; acc[i].z += g * v[j][i].z; acc[i].w += g * v[j][i].w; }
    }
  float* o = out + (size_t)row * d;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < d) *reinterpret_cast<float4*>(o + c) = acc[i];
  }
}

// part[blockIdx.x][j] = sum over this workgroup's rows of <dout, LN_j(z_j)>; d_active = gate[active] * dout.
// LNB (round 6): the sampled candidate's LayerNorm BACKWARD happens here as well -- its row z_active is already in registers
// for the gate's inner product, and d_active = gate[active] * dout would only be written to be read back by the candidate's
// own ln_bwd launch (8 us + a launch boundary per node of the architecture step).  With LNB the kernel writes dz (gradient
// wrt z_active), dt (the same behind the candidate's output dropout, replayed; NULL: none) and leaves the LayerNorm
// parameter partials [workgroup][3][d] (dln_a, dln_b, column sums of dt) exactly as ln_bwd_kernel (rowops.hip) does -- the
// same arithmetic operation for operation, the same row -> workgroup map (<= 512 workgroups) -- and d_active is not written.
template <int NV, bool LNB>
__global__ void __launch_bounds__(256) node_mix_bwd_kernel(NodeMixArgs p, const float* __restrict__ gate, const float* __restrict__ dout,
                                                           float* __restrict__ d_active, int active, float* __restrict__ part,
                                                           int M, int d, float eps, float* __restrict__ ln_dz, float* __restrict__ ln_dt,
                                                           float* __restrict__ ln_part, DropCfg ln_drop) {
  __shared__ float red[4][MAXC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float s[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) s[j] = 0.f;
  const float ga = (d_active || LNB) ? gate[active] : 0.f;
  float4 acc_a[NV], acc_b[NV], acc_c[NV], lnav[NV], lnbv[NV];
  if (LNB) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      acc_a[i] = make_float4(0.f, 0.f, 0.f, 0.f); acc_b[i] = acc_a[i]; acc_c[i] = acc_a[i];
      lnav[i] = (c < d) ? *reinterpret_cast<const float4*>(p.a[active] + c) : acc_a[i];
      lnbv[i] = (c < d) ? *reinterpret_cast<const float4*>(p.b[active] + c) : acc_a[i];
    }
  }
  for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
    float4 g4[NV];
    const float* dr = dout + (size_t)row * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      g4[i] = (c < d) ? *reinterpret_cast<const float4*>(dr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 v[MAXC][NV];
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      if (j < p.n && p.z[j]) {
        const float* zr = p.z[j] + (size_t)row * d;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int c = (lane + 64 * i) * 4;
          v[j][i] = (c < d) ? *reinterpret_cast<const float4*>(zr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    if (d_active && !LNB) {
      float* ar = d_active + (size_t)row * d;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) *reinterpret_cast<float4*>(ar + c) = make_float4(ga * g4[i].x, ga * g4[i].y, ga * g4[i].z, ga * g4[i].w);
      }
    }
#pragma unroll
    for (int j = 0; j < MAXC; ++j)
      if (j < p.n && p.z[j]) {
        if (LNB && j == active) {
          // LayerNorm forward (for the gate's inner product) AND backward of the sampled candidate: ln_fwd / ln_bwd_kernel's
          // arithmetic on the row in registers; g = ga * dout is the gradient of the candidate's output
          float4* vv = v[j];
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < NV; ++i) t += (vv[i].x + vv[i].y) + (vv[i].z + vv[i].w);
          const float mean = wave_sum(t) / (float)d;
          float ss = 0.f, sg = 0.f, sgc = 0.f;
          float4 gy[NV];
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const int c = (lane + 64 * i) * 4;
            gy[i] = make_float4(ga * g4[i].x, ga * g4[i].y, ga * g4[i].z, ga * g4[i].w);
            if (c < d) {
              vv[i].x -= mean; vv[i].y -= mean; vv[i].z -= mean; vv[i].w -= mean;
              ss += (vv[i].x * vv[i].x + vv[i].y * vv[i].y) + (vv[i].z * vv[i].z + vv[i].w * vv[i].w);
              const float gx = gy[i].x * lnav[i].x, gyy = gy[i].y * lnav[i].y, gz = gy[i].z * lnav[i].z, gw = gy[i].w * lnav[i].w;
              sg += (gx + gyy) + (gz + gw);
              sgc += (gx * vv[i].x + gyy * vv[i].y) + (gz * vv[i].z + gw * vv[i].w);
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
              // the candidate's output row (ln_fwd's expression) for the gate gradient
              s[j] += (g4[i].x * (lnav[i].x * vv[i].x * inv + lnbv[i].x) + g4[i].y * (lnav[i].y * vv[i].y * inv + lnbv[i].y)) +
                      (g4[i].z * (lnav[i].z * vv[i].z * inv + lnbv[i].z) + g4[i].w * (lnav[i].w * vv[i].w * inv + lnbv[i].w));
              float4 o;
              o.x = (gy[i].x * lnav[i].x - mg) * inv - vv[i].x * k2;
              o.y = (gy[i].y * lnav[i].y - mg) * inv - vv[i].y * k2;
              o.z = (gy[i].z * lnav[i].z - mg) * inv - vv[i].z * k2;
              o.w = (gy[i].w * lnav[i].w - mg) * inv - vv[i].w * k2;
              *reinterpret_cast<float4*>(ln_dz + (size_t)row * d + c) = o;
              acc_a[i].x += gy[i].x * vv[i].x * inv; acc_a[i].y += gy[i].y * vv[i].y * inv;
              acc_a[i].z += gy[i].z * vv[i].z * inv; acc_a[i].w += gy[i].w * vv[i].w * inv;
              acc_b[i].x += gy[i].x; acc_b[i].y += gy[i].y; acc_b[i].z += gy[i].z; acc_b[i].w += gy[i].w;
              if (ln_dt) {
                if (ln_drop.thresh) {
                  const uint32_t base = (uint32_t)row * (uint32_t)d + (uint32_t)c;
                  o.x *= drop_mult(ln_drop, base); o.y *= drop_mult(ln_drop, base + 1);
                  o.z *= drop_mult(ln_drop, base + 2); o.w *= drop_mult(ln_drop, base + 3);
                }
                *reinterpret_cast<float4*>(ln_dt + (size_t)row * d + c) = o;
                acc_c[i].x += o.x; acc_c[i].y += o.y; acc_c[i].z += o.z; acc_c[i].w += o.w;
              }
            }
          }
          continue;
        }
        if (p.a[j]) node_ln_row<NV>(v[j], p.a[j], p.b[j], lane, d, eps);
#pragma unroll
        for (int i = 0; i < NV; ++i)
          s[j] += (g4[i].x * v[j][i].x + g4[i].y * v[j][i].y) + (g4[i].z * v[j][i].z + g4[i].w * v[j][i].w);
      }
  }
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const float t = wave_sum(s[j]);
    if (lane == 0) red[wave][j] = t;
  }
  __syncthreads();
  if (threadIdx.x < MAXC)
    part[(size_t)blockIdx.x * MAXC + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (LNB) {   // the LayerNorm parameter partials of this workgroup: ln_bwd_kernel's reduction (16-byte reads, one row store per wave < 3)
    __shared__ __attribute__((aligned(16))) float lred[3][4][64 * 4];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + 64 * i) * 4;
      __syncthreads();
      *reinterpret_cast<float4*>(&lred[0][wave][lane * 4]) = acc_a[i];
      *reinterpret_cast<float4*>(&lred[1][wave][lane * 4]) = acc_b[i];
      *reinterpret_cast<float4*>(&lred[2][wave][lane * 4]) = acc_c[i];
      __syncthreads();
      if (wave < 3 && c < d) {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(&lred[wave][0][lane * 4]);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(&lred[wave][1][lane * 4]);
        const f32x4 p2 = *reinterpret_cast<const f32x4*>(&lred[wave][2][lane * 4]);
        const f32x4 p3 = *reinterpret_cast<const f32x4*>(&lred[wave][3][lane * 4]);
        *reinterpret_cast<f32x4*>(ln_part + ((size_t)blockIdx.x * 3 + wave) * d + c) = (p0 + p1) + (p2 + p3);
      }
    }
  }
}

// one thread per node (row): 'full'-mode architecture gradient + Adam
__global__ void alpha_full_step_kernel(float* __restrict__ prob, const float* __restrict__ gate_grad, float* __restrict__ m,
                                       float* __restrict__ v, float* __restrict__ prob_grad, int rows, int width, float lr,
                                       float b1, float b2, float eps, float c1, float c2s) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float* a = prob + (size_t)r * width;
  const float* g = gate_grad + (size_t)r * width;
  float mx = -INFINITY;
  for (int i = 0; i < width; ++i) mx = fmaxf(mx, a[i]);
  float den = 0.f;
  for (int i = 0; i < width; ++i) den += expf(a[i] - mx);   // padding columns hold -inf: exp = 0
  float dot = 0.f;
  for (int i = 0; i < width; ++i) dot += g[i] * (expf(a[i] - mx) / den);
  for (int i = 0; i < width; ++i) {
    const float p = expf(a[i] - mx) / den;
    const float grad = p * (g[i] - dot);                    // sum_j g_j p_j (delta_ij - p_i), mixed.py:194-198
    if (prob_grad) prob_grad[(size_t)r * width + i] = grad;
    if (p == 0.f && !(a[i] > -INFINITY)) continue;          // padding column: stays -inf
    const size_t o = (size_t)r * width + i;
    const float mi = b1 * m[o] + (1.f - b1) * grad;
    const float vi = b2 * v[o] + (1.f - b2) * grad * grad;
    m[o] = mi; v[o] = vi;
    a[i] -= (lr / c1) * mi / (sqrtf(vi) / c2s + eps);
  }
}

}  // namespace mmnas

using namespace mmnas;

static int mix_grid(size_t n4) {
  const size_t b = (n4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

extern "C" size_t mmnas_mixed_sum_ws_floats(void) { return (size_t)2048 * MAXC; }

extern "C" int mmnas_mixed_sum_fwd(const float* const* outs_host, int n, const float* gate, float* out, size_t count,
                                   void* stream) {
  MMNAS_REQUIRE(outs_host && gate && out, MMNAS_E_ARG, "mmnas_mixed_sum_fwd: null pointer");
  MMNAS_REQUIRE(n >= 1 && n <= MAXC, MMNAS_E_SHAPE, "mmnas_mixed_sum_fwd: 1..%d candidates, got %d", MAXC, n);
  MMNAS_REQUIRE(count % 4 == 0, MMNAS_E_SHAPE, "mmnas_mixed_sum_fwd: element count must be a multiple of 4");
  MixArgs a;
  a.n = n;
  for (int j = 0; j < MAXC; ++j) {
    a.o[j] = j < n ? outs_host[j] : nullptr;
    MMNAS_REQUIRE(((uintptr_t)a.o[j] & 15) == 0, MMNAS_E_ARG, "mmnas_mixed_sum_fwd: candidate %d unaligned", j);
  }
  if (count == 0) return MMNAS_OK;
  ProfScope ps(MMNAS_K_ROWOPS, 2.0 * n * count, 4.0 * (n + 1) * count, (hipStream_t)stream, "mixed_sum_fwd");
  MMNAS_LAUNCH(mixed_sum_fwd_kernel, dim3(mix_grid(count / 4)), dim3(256), 0, (hipStream_t)stream, a, gate, out, count / 4);
  return check_launch("mixed_sum_fwd");
}

extern "C" int mmnas_mixed_sum_bwd(const float* const* outs_host, int n, const float* gate, const float* dout,
                                   float* d_active, int active, float* dgate, float* ws, size_t count, void* stream) {
  MMNAS_REQUIRE(outs_host && gate && dout && dgate && ws, MMNAS_E_ARG, "mmnas_mixed_sum_bwd: null pointer");
  MMNAS_REQUIRE(n >= 1 && n <= MAXC, MMNAS_E_SHAPE, "mmnas_mixed_sum_bwd: 1..%d candidates, got %d", MAXC, n);
  MMNAS_REQUIRE(count % 4 == 0, MMNAS_E_SHAPE, "mmnas_mixed_sum_bwd: element count must be a multiple of 4");
  MMNAS_REQUIRE(!d_active || (active >= 0 && active < n), MMNAS_E_ARG, "mmnas_mixed_sum_bwd: active index out of range");
  MixArgs a;
  a.n = n;
  for (int j = 0; j < MAXC; ++j) a.o[j] = j < n ? outs_host[j] : nullptr;
  if (count == 0) return MMNAS_OK;
  const int g = mix_grid(count / 4);
  ProfScope ps(MMNAS_K_ROWOPS, 2.0 * n * count, 4.0 * (n + 2) * count, (hipStream_t)stream, "mixed_sum_bwd");
  MMNAS_LAUNCH(mixed_sum_bwd_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, a, gate, dout, d_active, active, ws, count / 4);
  MMNAS_LAUNCH(mixed_sum_reduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, g, n, dgate);
  return check_launch("mixed_sum_bwd");
}

extern "C" int mmnas_alpha_full_step(float* prob, const float* gate_grad, float* m, float* v, float* prob_grad, int rows,
                                     int width, float lr, float beta1, float beta2, float eps, int step, void* stream) {
  MMNAS_REQUIRE(prob && gate_grad && m && v && step >= 1 && rows >= 0 && width >= 1, MMNAS_E_ARG, "mmnas_alpha_full_step: bad arguments");
  if (rows == 0) return MMNAS_OK;
  const float c1 = 1.f - powf(beta1, (float)step);
  const float c2s = sqrtf(1.f - powf(beta2, (float)step));
  MMNAS_LAUNCH(alpha_full_step_kernel, dim3(cdiv(rows, 64)), dim3(64), 0, (hipStream_t)stream, prob, gate_grad, m, v, prob_grad,
               rows, width, lr, beta1, beta2, eps, c1, c2s);
  return check_launch("alpha_full_step");
}


namespace mmnas {
static int node_args(NodeMixArgs& a, const float* const* z, const float* const* ln_a, const float* const* ln_b, int n, int d, const char* who) {
  MMNAS_REQUIRE(z && n >= 1 && n <= MAXC, MMNAS_E_SHAPE, "%s: 1..%d candidates, got %d", who, MAXC, n);
  MMNAS_REQUIRE(d >= 4 && d % 4 == 0 && d <= 1024, MMNAS_E_SHAPE, "%s: d=%d (multiple of 4, <= 1024)", who, d);
  memset(&a, 0, sizeof(a));
  a.n = n;
  for (int j = 0; j < n; ++j) {
    a.z[j] = z[j];
    a.a[j] = ln_a ? ln_a[j] : nullptr;
    a.b[j] = ln_b ? ln_b[j] : nullptr;
    MMNAS_REQUIRE(((uintptr_t)a.z[j] & 15) == 0, MMNAS_E_ARG, "%s: candidate %d: unaligned input", who, j);   // (NULL: not evaluated)
    MMNAS_REQUIRE(!a.a[j] || (a.b[j] && (((uintptr_t)a.a[j] | (uintptr_t)a.b[j]) & 15) == 0), MMNAS_E_ARG, "%s: candidate %d: LayerNorm parameters", who, j);
    if (d < 2) MMNAS_REQUIRE(!a.a[j], MMNAS_E_SHAPE, "%s: LayerNorm needs d >= 2", who);
  }
  return MMNAS_OK;
}
}  // namespace mmnas

extern "C" int mmnas_node_mix_fwd(const float* const* z, const float* const* ln_a, const float* const* ln_b, int n, const float* gate,
                                  float* out, int M, int d, float eps, void* stream) {
  NodeMixArgs a;
  int rc = node_args(a, z, ln_a, ln_b, n, d, "mmnas_node_mix_fwd");
  if (rc) return rc;
  MMNAS_REQUIRE(gate && out && M >= 0, MMNAS_E_ARG, "mmnas_node_mix_fwd: null pointer");
  if (M == 0) return MMNAS_OK;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(MMNAS_K_ROWOPS, (2.0 * n + 8.0 * n) * M * d, 4.0 * (n + 1) * M * d, st, "node_mix_fwd");
  const dim3 grid(cdiv(M, 4)), block(256);
  if (d <= 256) MMNAS_LAUNCH((node_mix_fwd_kernel<1>), grid, block, 0, st, a, gate, out, M, d, eps);
  else if (d <= 512) MMNAS_LAUNCH((node_mix_fwd_kernel<2>), grid, block, 0, st, a, gate, out, M, d, eps);
  else MMNAS_LAUNCH((node_mix_fwd_kernel<4>), grid, block, 0, st, a, gate, out, M, d, eps);
  return check_launch("node_mix_fwd");
}

namespace mmnas {
// reduce == false: the partial sums stay in ws[0 .. *nwg_out * MAXC) for mixed_reduce_many()
int mixed_reduce_many(const float* const* parts, float* const* dgates, const int* nwg, const int* n, int count, hipStream_t st) {
  for (int i0 = 0; i0 < count; i0 += MMNAS_MIX_REDUCE_MAX) {
    MixReduceJobs jobs;
    const int c = count - i0 < MMNAS_MIX_REDUCE_MAX ? count - i0 : MMNAS_MIX_REDUCE_MAX;
    for (int i = 0; i < c; ++i) { jobs.part[i] = parts[i0 + i]; jobs.dgate[i] = dgates[i0 + i]; jobs.nwg[i] = nwg[i0 + i]; jobs.n[i] = n[i0 + i]; }
    ProfScope ps(MMNAS_K_ROWOPS, 0.0, 0.0, st, "mixed_reduce_many");
    MMNAS_LAUNCH(mixed_sum_reduce_many_kernel, dim3(c), dim3(256), 0, st, jobs);
  }
  return check_launch("mixed_reduce_many");
}
}  // namespace mmnas

extern "C" int mmnas_node_mix_bwd(const float* const* z, const float* const* ln_a, const float* const* ln_b, int n, const float* gate,
                                  const float* dout, float* d_active, int active, float* dgate, float* ws, int M, int d, float eps,
                                  void* stream) {
  return mmnas::node_mix_bwd_impl(z, ln_a, ln_b, n, gate, dout, d_active, active, dgate, ws, M, d, eps, (hipStream_t)stream, true, nullptr);
}

int mmnas::node_mix_bwd_impl(const float* const* z, const float* const* ln_a, const float* const* ln_b, int n, const float* gate,
                             const float* dout, float* d_active, int active, float* dgate, float* ws, int M, int d, float eps,
                             hipStream_t stream, bool reduce, int* nwg_out, const NodeLnBwd* lnb) {
  NodeMixArgs a;
  int rc = node_args(a, z, ln_a, ln_b, n, d, "mmnas_node_mix_bwd");
  if (rc) return rc;
  MMNAS_REQUIRE(gate && dout && dgate && ws && M >= 0, MMNAS_E_ARG, "mmnas_node_mix_bwd: null pointer");
  MMNAS_REQUIRE(!d_active || (active >= 0 && active < n), MMNAS_E_ARG, "mmnas_node_mix_bwd: active index out of range");
  if (nwg_out) *nwg_out = 0;
  if (M == 0) return MMNAS_OK;
  hipStream_t st = stream;
  const bool fuse = lnb && lnb->dz;
  if (fuse) MMNAS_REQUIRE(active >= 0 && active < n && z[active] && ln_a[active] && ln_b[active] && lnb->part && d <= 256, MMNAS_E_ARG,
                          "mmnas_node_mix_bwd: the fused LayerNorm backward needs a normalised active candidate (d <= 256)");
  int g = cdiv(M, 4);
  if (g > 2048) g = 2048;
  if (fuse && g > 512) g = 512;      // ln_bwd_kernel's row -> workgroup map: the partial rows fit mmnas_layernorm_bwd_ws_floats
  if (nwg_out) *nwg_out = g;
  ProfScope ps(MMNAS_K_ROWOPS, (2.0 * n + 8.0 * n) * M * d + (fuse ? 16.0 * M * d : 0.0), 4.0 * (n + 2 + (fuse ? 1 : 0)) * M * d, st, "node_mix_bwd");
  const dim3 grid(g), block(256);
  float* const ndz = fuse ? lnb->dz : nullptr; float* const ndt = fuse ? lnb->dt : nullptr; float* const npart = fuse ? lnb->part : nullptr;
  const DropCfg ndrop = fuse ? lnb->drop : make_drop(0.f, 0, 0);
  // (d <= 256 only: the two-chunk instantiation <2, true> crashes this compiler's machine copy propagation, as a [2][NV] row
  //  array did in rowops.hip; the supernet -- the one user of mixed chains -- is 256 wide)
  if (fuse) MMNAS_LAUNCH((node_mix_bwd_kernel<1, true>), grid, block, 0, st, a, gate, dout, d_active, active, ws, M, d, eps, ndz, ndt, npart, ndrop);
  else if (d <= 256) MMNAS_LAUNCH((node_mix_bwd_kernel<1, false>), grid, block, 0, st, a, gate, dout, d_active, active, ws, M, d, eps, ndz, ndt, npart, ndrop);
  else if (d <= 512) MMNAS_LAUNCH((node_mix_bwd_kernel<2, false>), grid, block, 0, st, a, gate, dout, d_active, active, ws, M, d, eps, ndz, ndt, npart, ndrop);
  else MMNAS_LAUNCH((node_mix_bwd_kernel<4, false>), grid, block, 0, st, a, gate, dout, d_active, active, ws, M, d, eps, ndz, ndt, npart, ndrop);
  if (reduce) MMNAS_LAUNCH(mixed_sum_reduce_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, g, n, dgate);
  return check_launch("node_mix_bwd");
}

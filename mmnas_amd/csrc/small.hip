// Short-sequence operators (the language stream of the VQA / VGD nets: 14 tokens per sample, M = B*S = 896 rows).
// On the general path an operator of this size is 4 dependent launches of 56..224 workgroups forward (QKV projection,
// attention core, merge projection + residual, LayerNorm) and 4 backward, each at the ~10 us floor of a latency-bound
// launch on a quarter of the CUs.  Here ONE launch does the whole forward of SelfAtt (modules.py:248-270 with
// MHAtt.forward / .att, modules.py:178-199) for sequences of <= 16 rows:
//
//   grid (H, B): a workgroup owns one head of one sample.
//     1. x_b [S, d] -> LDS.  Q_h, K_h, V_h = x_b W_{q,k,v}[64h : 64h+64, :]^T on v_mfma_f32_16x16x4_f32 (the 16 MFMA rows
//        are the sequence: no padding beyond 14 -> 16); the weight rows stream from L2 straight into B fragments, one
//        16-byte load per lane feeding 4 MFMAs (k order permuted identically for both operands).
//     2. scores, mask, softmax, attention dropout, A V for the head: 32 MFMAs, every wave redundantly does the 16x16
//        score tile and owns 16 of the 64 output columns.
//     3. the head's share of the merge projection  O_h Wm[:, 64h : 64h+64]^T  [S, d]  goes to a workspace slot with
//        write-through stores; the LAST workgroup of a sample to arrive (arrival counter, agent scope) adds the H
//        shares in head order (bitwise reproducible), applies the output dropout, the residual and the LayerNorm.
//   Hand-off = the stream-K recipe of gemm.hip: sc1 stores -> s_waitcnt vmcnt(0) -> barrier -> one atomic; the
//   finisher reads with agent-scope loads.  Nobody waits: a workgroup that is not last simply exits.
// The saved block (Q, K, V, attention output, row statistics, pre-LayerNorm sum) is written exactly as the general path
// writes it, so either backward may follow.
#include <string.h>
#include "common.h"

namespace mmnas {

int sk_workspace(hipStream_t st, float** ws, size_t* ws_floats, int** cnt, int* ncnt);   // gemm.hip

typedef unsigned long long u64_;
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

struct SaSmallK {
  int B, S, H, flags;
  const float* x; const uint8_t* mask;
  const float* Wq; const float* Wk; const float* Wv; const float* Wm;
  const float* ln_a; const float* ln_b;
  float* Q; float* K; float* V; float* att; float* stats; float* z; float* y;
  float* part; int* cnt;
  DropCfg drop_att, drop_out;
  float eps;
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
#define MFMA16x4(ACC, AF, BF)        \
  ACC = mfma16(AF.x, BF.x, ACC);     \
  ACC = mfma16(AF.y, BF.y, ACC);     \
  ACC = mfma16(AF.z, BF.z, ACC);     \
  ACC = mfma16(AF.w, BF.w, ACC);

__device__ __forceinline__ void st_agent_f(float* ptr, float v) {
  __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64_ ld_agent_u64(const u64_* ptr) {
  return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int D>
__global__ void __launch_bounds__(256, 1) sa_small_fwd_kernel(const SaSmallK p) {
  constexpr int LDX = D + 4, LDH = 68;
  constexpr int KC = 128, NCH = D / KC, SPC = KC / 16;   // reduction chunks of the QKV projection; 16-wide k steps per chunk
  constexpr int NTM = D / 64;                            // merge-projection column tiles per wave
  __shared__ __attribute__((aligned(16))) float xs[16 * LDX];
  __shared__ __attribute__((aligned(16))) float qs[16 * LDH];
  __shared__ __attribute__((aligned(16))) float ks[16 * LDH];
  __shared__ __attribute__((aligned(16))) float vs[16 * LDH];
  __shared__ __attribute__((aligned(16))) float os[16 * LDH];
  __shared__ float smask[16];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y, S = p.S, H = p.H;
  const size_t row0 = (size_t)b * S;

  // ---- weight fragments of the first reduction chunk in flight before anything else ----
  const float* const wrow[3] = {p.Wq + (size_t)(64 * h + 16 * w + l15) * D + 4 * kq,
                                p.Wk + (size_t)(64 * h + 16 * w + l15) * D + 4 * kq,
                                p.Wv + (size_t)(64 * h + 16 * w + l15) * D + 4 * kq};
  float4 wb[2][3][SPC];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int s = 0; s < SPC; ++s) wb[0][m][s] = *reinterpret_cast<const float4*>(wrow[m] + 16 * s);

  // ---- x_b -> LDS (rows >= S are zero) ----
  {
    constexpr int F4 = D / 4, N = 16 * F4 / 256;
    float4 xv[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int f = tid + 256 * i, r = f / F4, c4 = f - r * F4;
      const float4 t = *reinterpret_cast<const float4*>(p.x + (row0 + (r < S ? r : 0)) * D + 4 * c4);
      xv[i] = r < S ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int f = tid + 256 * i, r = f / F4, c4 = f - r * F4;
      *reinterpret_cast<float4*>(xs + r * LDX + 4 * c4) = xv[i];
    }
    if (tid < 16) smask[tid] = (p.mask && tid < S && p.mask[row0 + tid]) ? 1.f : 0.f;
  }
  __syncthreads();

  // ---- Q_h, K_h, V_h: wave w owns columns 16w..16w+15 of each ----
  f32x4 acc[3];
#pragma unroll
  for (int m = 0; m < 3; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c + 1 < NCH) {
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int s = 0; s < SPC; ++s)
          wb[(c + 1) & 1][m][s] = *reinterpret_cast<const float4*>(wrow[m] + KC * (c + 1) + 16 * s);
    }
    __builtin_amdgcn_sched_barrier(0);   // the next chunk's loads stay in front of this chunk's MFMAs
#pragma unroll
    for (int s = 0; s < SPC; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(xs + l15 * LDX + KC * c + 16 * s + 4 * kq);
#pragma unroll
      for (int m = 0; m < 3; ++m) { MFMA16x4(acc[m], a, wb[c & 1][m][s]) }
    }
  }
  // merge-projection fragments: rows n = 64 t + 16 w + l15 of Wm, columns 64h + 16 s + 4 kq .. +3 (in flight during the core)
  float4 wm[NTM][4];
#pragma unroll
  for (int t = 0; t < NTM; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s)
      wm[t][s] = *reinterpret_cast<const float4*>(p.Wm + (size_t)(64 * t + 16 * w + l15) * D + 64 * h + 16 * s + 4 * kq);
  __builtin_amdgcn_sched_barrier(0);
  {
    float* const dst[3] = {qs, ks, vs};
    float* const gdst[3] = {p.Q, p.K, p.V};
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s = 4 * kq + r;
        dst[m][s * LDH + 16 * w + l15] = acc[m][r];
        if (s < S) gdst[m][(row0 + s) * D + 64 * h + 16 * w + l15] = acc[m][r];
      }
  }
  __syncthreads();

  // ---- scores^T[key][query] = K Q^T (every wave), softmax over the keys of the lane's query ----
  f32x4 sc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 kf = *reinterpret_cast<const float4*>(ks + l15 * LDH + 16 * s + 4 * kq);
    const float4 qf = *reinterpret_cast<const float4*>(qs + l15 * LDH + 16 * s + 4 * kq);
    MFMA16x4(sc, kf, qf)
  }
  const int qi = l15;
  const size_t bh = (size_t)b * H + h;
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int key = 4 * kq + r;
    float v = sc[r] * 0.125f;
    if (key < S) { if (smask[key] != 0.f) v = -1e9f; } else v = -INFINITY;
    sc[r] = v;
    mx = fmaxf(mx, v);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) { sc[r] = __expf(sc[r] - mx); sum += sc[r]; }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  if (w == 0 && kq == 0 && qi < S) {
    p.stats[(bh * S + qi) * 2] = mx;
    p.stats[(bh * S + qi) * 2 + 1] = inv;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float a = sc[r] * inv;
    if (p.drop_att.thresh) a *= drop_mult(p.drop_att, (uint32_t)((bh * S + qi) * S + 4 * kq + r));
    sc[r] = a;
  }
  // ---- O[query][16w + n] = sum_key P[query][key] V[key][16w + n]: k slot kq of step r stands for key 4 kq + r ----
  f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) o = mfma16(sc[r], vs[(4 * kq + r) * LDH + 16 * w + l15], o);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int s = 4 * kq + r;
    os[s * LDH + 16 * w + l15] = o[r];
    if (s < S) p.att[(row0 + s) * D + 64 * h + 16 * w + l15] = o[r];
  }
  __syncthreads();

  // ---- this head's share of the merge projection: columns 64 t + 16 w + l15 ----
  f32x4 pm[NTM];
#pragma unroll
  for (int t = 0; t < NTM; ++t) pm[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 a = *reinterpret_cast<const float4*>(os + l15 * LDH + 16 * s + 4 * kq);
#pragma unroll
    for (int t = 0; t < NTM; ++t) { MFMA16x4(pm[t], a, wm[t][s]) }
  }
  float* const slot = p.part + (bh * 16) * D;
#pragma unroll
  for (int t = 0; t < NTM; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) st_agent_f(slot + (4 * kq + r) * D + 64 * t + 16 * w + l15, pm[t][r]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) s_last = __hip_atomic_fetch_add(p.cnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == H - 1;
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) __hip_atomic_store(p.cnt + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch

  // ---- finisher: z = x + drop(sum_h share_h); y = LN(z).  Wave w takes rows w, w+4, w+8, w+12; a lane 4 floats per 256
  //      columns.  Every share of a row group is in flight at once (sc1 loads: the shares come from other XCDs' L2s) ----
  constexpr int HH = D / 64;      // heads (di == d, heads of 64)
  constexpr int NV = D / 256;     // 16-byte words per lane and row
  constexpr int RG = D == 256 ? 4 : 2;   // rows per load batch
  const bool norm = p.flags & MMNAS_F_NORM, resid = p.flags & MMNAS_F_RESIDUAL;
  const __amdgpu_buffer_rsrc_t prsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.part + (size_t)b * HH * 16 * D), 0, (unsigned)(HH * 16 * D * 4), 0x00020000);
#pragma unroll 1
  for (int g = 0; g < 4 / RG; ++g) {
    u32x4s t[RG][HH][NV];
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
      const int s = min(w + 4 * (g * RG + rr), 15);
#pragma unroll
      for (int hh = 0; hh < HH; ++hh)
#pragma unroll
        for (int i = 0; i < NV; ++i)
          t[rr][hh][i] = __builtin_amdgcn_raw_buffer_load_b128(prsrc, (unsigned)(((hh * 16 + s) * D + 4 * (lane + 64 * i)) * 4), 0, 16 /* sc1 */);
    }
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
      const int s = w + 4 * (g * RG + rr);
      if (s >= S) continue;     // wave-uniform
      float v[4 * NV];
#pragma unroll
      for (int i = 0; i < 4 * NV; ++i) v[i] = 0.f;
#pragma unroll
      for (int hh = 0; hh < HH; ++hh)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          v[4 * i] += __uint_as_float(t[rr][hh][i].x); v[4 * i + 1] += __uint_as_float(t[rr][hh][i].y);
          v[4 * i + 2] += __uint_as_float(t[rr][hh][i].z); v[4 * i + 3] += __uint_as_float(t[rr][hh][i].w);
        }
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int col = 4 * (lane + 64 * i) + e;
          float x = v[4 * i + e];
          if (p.drop_out.thresh) x *= drop_mult(p.drop_out, (uint32_t)(row0 + s) * (uint32_t)D + (uint32_t)col);
          if (resid) x += xs[s * LDX + col];
          v[4 * i + e] = x;
          sm += x;
        }
      float* const yr = p.y + (row0 + s) * D;
      if (!norm) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
          *reinterpret_cast<float4*>(yr + 4 * (lane + 64 * i)) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        continue;
      }
      float* const zr = p.z + (row0 + s) * D;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        *reinterpret_cast<float4*>(zr + 4 * (lane + 64 * i)) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
      const float mean = wave_sum(sm) / (float)D;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 4 * NV; ++i) { v[i] -= mean; ss += v[i] * v[i]; }
      const float sd = sqrtf(wave_sum(ss) / (float)(D - 1));
      const float invs = 1.0f / (sd + p.eps);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int col = 4 * (lane + 64 * i);
        const float4 av = *reinterpret_cast<const float4*>(p.ln_a + col);
        const float4 bv = *reinterpret_cast<const float4*>(p.ln_b + col);
        *reinterpret_cast<float4*>(yr + col) = make_float4(av.x * v[4 * i] * invs + bv.x, av.y * v[4 * i + 1] * invs + bv.y,
                                                           av.z * v[4 * i + 2] * invs + bv.z, av.w * v[4 * i + 3] * invs + bv.w);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward twin: the data-gradient chain of SelfAtt's backward for <= 16 rows per sample in ONE launch.  On the general path
// it is four dependent launches (LayerNorm backward; d(att) = dt Wm paired with dWm; the attention core's backward; dx
// = dQ Wq + dK Wk + dV Wv + dz paired with dWq/dWk/dWv), each 56..224 workgroups at the latency floor.  Here, grid (H, B):
//   1. every workgroup of a sample runs LayerNorm backward over the sample's S rows (redundantly: it needs the WHOLE rows
//      of dt as the A operand of step 2; 14 rows x d is nothing) -- the arithmetic of ln_bwd_kernel in its order, the
//      same column -> lane mapping, so dz comes out as that kernel writes it; dt = dz o output-dropout mask.  Workgroup
//      h writes columns 64h.. of dt (the A operand of the dWm product) and, h == 0, the sample's row of LayerNorm
//      parameter partials.
//   2. d(att)_h = dt Wm[:, 64h : 64h+64]  [S, 64]: the reduction runs over Wm's ROWS, so a lane's 4 k-values are four
//      4-byte loads 1 KB apart (16 lanes x 4 rows = four 64-byte segments per instruction).
//   3. the head's softmax backward on 16x16 tiles in BOTH orientations (scores^T and scores; dA^T and dA: 32 MFMAs each
//      pair): the accumulator layout of one orientation is the A operand of dQ = dZ K, the other's of dK = dZ^T Q and
//      dV = A^T d(att).  delta = sum_key dA A from the tile itself.
//   4. the head's share of dx = dQ_h Wq[64h.., :] + dK_h Wk[64h.., :] + dV_h Wv[64h.., :]: weight rows are contiguous
//      along the OUTPUT column here, so a lane loads 16 bytes = one k, 4 consecutive columns, and the four 16-column
//      MFMA tiles of a 64-column block take columns = e mod 4 (e = 0..3): one load feeds 4 MFMAs, and a lane's four
//      results per row are 16 contiguous bytes of the share.
//   5. hand-off as forward: shares to the workspace, the last workgroup of the sample adds them in head order + dz.
// What remains are the parameter gradients: dWm, dWq, dWk, dWv as ONE grouped launch (4 x [d, d], reduction over the
// B*S rows) that also carries the LayerNorm parameter reduction (ops.hip: att_bwd_impl).
struct SaSmallBwdK {
  int B, S, H, flags;
  const float* dy; const float* z; const float* ln_a;
  const uint8_t* mask;
  const float* Wq; const float* Wk; const float* Wv; const float* Wm;
  const float* Q; const float* K; const float* V; const float* stats;
  float* dt;       // [M, d] gradient wrt the core's output, written for the dWm product (NULL: dy itself is it)
  float* dQ; float* dK; float* dV; float* dx;
  float* lnpart;   // [B][3][d] LayerNorm parameter partial rows (w = 0: a, 1: b; NULL without NORM)
  float* part; int* cnt;
  DropCfg drop_att, drop_out;
  float eps;
};

__device__ __forceinline__ void st_agent_f2(float* ptr, float a, float b) {
  const u64_ v = (u64_)__float_as_uint(a) | ((u64_)__float_as_uint(b) << 32);
  __hip_atomic_store(reinterpret_cast<u64_*>(ptr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int D>
__global__ void __launch_bounds__(256, 1) sa_small_bwd_kernel(const SaSmallBwdK p) {
  constexpr int LDX = D + 4, LDH = 68, NV = D / 256, HH = D / 64;
  constexpr int KC = 128, NCH = D / KC, SPC = KC / 16;
  constexpr int NTB = D / 256;   // 64-column blocks of dx per wave
  __shared__ __attribute__((aligned(16))) float dts[16 * LDX];
  __shared__ __attribute__((aligned(16))) float dzs[16 * LDX];
  __shared__ __attribute__((aligned(16))) float qs[16 * LDH], ks[16 * LDH], vs[16 * LDH], gs[16 * LDH];
  __shared__ __attribute__((aligned(16))) float dqs[16 * LDH], dks[16 * LDH], dvs[16 * LDH];
  __shared__ __attribute__((aligned(16))) float red[2][4][D];
  __shared__ float smask[16], smx[16], sinv[16], sdel[4][16];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.y, S = p.S, H = p.H;
  const size_t row0 = (size_t)b * S;
  const size_t bh = (size_t)b * H + h;
  const bool norm = p.flags & MMNAS_F_NORM, resid = p.flags & MMNAS_F_RESIDUAL;
  const bool drop_o = p.drop_out.thresh != 0;

  // ---- in flight first: the sample's dy / z rows of this wave, the head's Q / K / V tiles, Wm fragments of chunk 0 ----
  float4 zv[4][NV], gv[4][NV], av[NV];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int s = min(w + 4 * g, S - 1);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      gv[g][i] = *reinterpret_cast<const float4*>(p.dy + (row0 + s) * D + c);
      zv[g][i] = norm ? *reinterpret_cast<const float4*>(p.z + (row0 + s) * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) av[i] = norm ? *reinterpret_cast<const float4*>(p.ln_a + 4 * (lane + 64 * i)) : make_float4(1.f, 1.f, 1.f, 1.f);
  {
    const int r = tid >> 4, c4 = tid & 15;
    const size_t o = (row0 + (r < S ? r : 0)) * D + 64 * h + 4 * c4;
    float4 q4 = *reinterpret_cast<const float4*>(p.Q + o), k4 = *reinterpret_cast<const float4*>(p.K + o),
           v4 = *reinterpret_cast<const float4*>(p.V + o);
    const bool in = r < S;
    q4.x = in ? q4.x : 0.f; q4.y = in ? q4.y : 0.f; q4.z = in ? q4.z : 0.f; q4.w = in ? q4.w : 0.f;
    k4.x = in ? k4.x : 0.f; k4.y = in ? k4.y : 0.f; k4.z = in ? k4.z : 0.f; k4.w = in ? k4.w : 0.f;
    v4.x = in ? v4.x : 0.f; v4.y = in ? v4.y : 0.f; v4.z = in ? v4.z : 0.f; v4.w = in ? v4.w : 0.f;
    *reinterpret_cast<float4*>(qs + r * LDH + 4 * c4) = q4;
    *reinterpret_cast<float4*>(ks + r * LDH + 4 * c4) = k4;
    *reinterpret_cast<float4*>(vs + r * LDH + 4 * c4) = v4;
    if (tid < 16) {
      smask[tid] = (p.mask && tid < S && p.mask[row0 + tid]) ? 1.f : 0.f;
      smx[tid] = tid < S ? p.stats[(bh * S + tid) * 2] : 0.f;
      sinv[tid] = tid < S ? p.stats[(bh * S + tid) * 2 + 1] : 0.f;
    }
  }
  const float* const wmcol = p.Wm + (size_t)(4 * kq) * D + 64 * h + 16 * w + l15;
  float wmf[2][SPC][4];
#pragma unroll
  for (int s = 0; s < SPC; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) wmf[0][s][j] = wmcol[(size_t)(16 * s + j) * D];

  // ---- 1. LayerNorm backward of rows w, w+4, w+8, w+12 (ln_bwd_kernel's arithmetic) -> dzs, dts ----
  float4 acc_a[NV], acc_b[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { acc_a[i] = make_float4(0.f, 0.f, 0.f, 0.f); acc_b[i] = acc_a[i]; }
  auto ln_row = [&](const int g, const float4 (&zrow)[NV], const float4 (&grow)[NV]) __attribute__((always_inline)) {
    const int s = w + 4 * g;
    float4 o[NV];
    if (s >= S) {      // wave-uniform: rows of the 16-row MFMA tile beyond the sequence
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        *reinterpret_cast<float4*>(dts + s * LDX + 4 * (lane + 64 * i)) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(dzs + s * LDX + 4 * (lane + 64 * i)) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
    if (norm) {
      float4 v[NV];
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) { v[i] = zrow[i]; sm += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
      const float mean = wave_sum(sm) / (float)D;
      float ss = 0.f, sg = 0.f, sgc = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        const float gx = grow[i].x * av[i].x, gy = grow[i].y * av[i].y, gz = grow[i].z * av[i].z, gw = grow[i].w * av[i].w;
        sg += (gx + gy) + (gz + gw);
        sgc += (gx * v[i].x + gy * v[i].y) + (gz * v[i].z + gw * v[i].w);
      }
      ss = wave_sum(ss); sg = wave_sum(sg); sgc = wave_sum(sgc);
      const float sd = sqrtf(ss / (float)(D - 1));
      const float sden = sd + p.eps;
      const float inv = 1.0f / sden;
      const float mg = sg / (float)D;
      const float k2 = sgc / ((float)(D - 1) * sd * sden * sden);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 gg = grow[i];
        o[i].x = (gg.x * av[i].x - mg) * inv - v[i].x * k2;
        o[i].y = (gg.y * av[i].y - mg) * inv - v[i].y * k2;
        o[i].z = (gg.z * av[i].z - mg) * inv - v[i].z * k2;
        o[i].w = (gg.w * av[i].w - mg) * inv - v[i].w * k2;
        acc_a[i].x += gg.x * v[i].x * inv; acc_a[i].y += gg.y * v[i].y * inv;
        acc_a[i].z += gg.z * v[i].z * inv; acc_a[i].w += gg.w * v[i].w * inv;
        acc_b[i].x += gg.x; acc_b[i].y += gg.y; acc_b[i].z += gg.z; acc_b[i].w += gg.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) o[i] = grow[i];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = 4 * (lane + 64 * i);
      *reinterpret_cast<float4*>(dzs + s * LDX + c) = o[i];
      if (drop_o) {
        const uint32_t base = (uint32_t)(row0 + s) * (uint32_t)D + (uint32_t)c;
        o[i].x *= drop_mult(p.drop_out, base); o[i].y *= drop_mult(p.drop_out, base + 1);
        o[i].z *= drop_mult(p.drop_out, base + 2); o[i].w *= drop_mult(p.drop_out, base + 3);
      }
      *reinterpret_cast<float4*>(dts + s * LDX + c) = o[i];
    }
  };
  ln_row(0, zv[0], gv[0]); ln_row(1, zv[1], gv[1]); ln_row(2, zv[2], gv[2]); ln_row(3, zv[3], gv[3]);
  if (norm && h == 0) {   // (workgroup-uniform)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      *reinterpret_cast<float4*>(&red[0][w][4 * (lane + 64 * i)]) = acc_a[i];
      *reinterpret_cast<float4*>(&red[1][w][4 * (lane + 64 * i)]) = acc_b[i];
    }
  }
  __syncthreads();
  if (norm && h == 0 && p.lnpart) {
    for (int idx = tid; idx < 2 * (D / 4); idx += 256) {
      const int wh = idx / (D / 4), c = 4 * (idx - wh * (D / 4));
      const float4 r0 = *reinterpret_cast<const float4*>(&red[wh][0][c]), r1 = *reinterpret_cast<const float4*>(&red[wh][1][c]);
      const float4 r2 = *reinterpret_cast<const float4*>(&red[wh][2][c]), r3 = *reinterpret_cast<const float4*>(&red[wh][3][c]);
      *reinterpret_cast<float4*>(p.lnpart + ((size_t)b * 3 + wh) * D + c) =
          make_float4((r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y), (r0.z + r1.z) + (r2.z + r3.z), (r0.w + r1.w) + (r2.w + r3.w));
    }
  }
  if (p.dt) {   // this head's 64 columns of dt, for the dWm product
    const int r = tid >> 4, c4 = tid & 15;
    if (r < S) *reinterpret_cast<float4*>(p.dt + (row0 + r) * D + 64 * h + 4 * c4) = *reinterpret_cast<const float4*>(dts + r * LDX + 64 * h + 4 * c4);
  }

  // ---- 2. d(att)_h = dt Wm[:, 64h + 16w + l15] ----
  f32x4 ga = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c + 1 < NCH) {
#pragma unroll
      for (int s = 0; s < SPC; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) wmf[(c + 1) & 1][s][j] = wmcol[(size_t)(KC * (c + 1) + 16 * s + j) * D];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < SPC; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(dts + l15 * LDX + KC * c + 16 * s + 4 * kq);
      ga = mfma16(a.x, wmf[c & 1][s][0], ga);
      ga = mfma16(a.y, wmf[c & 1][s][1], ga);
      ga = mfma16(a.z, wmf[c & 1][s][2], ga);
      ga = mfma16(a.w, wmf[c & 1][s][3], ga);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) gs[(4 * kq + r) * LDH + 16 * w + l15] = ga[r];

  // dx-share weight fragments, first stage (Wq, k rows 64h .. 64h+31) in flight during the core
  //   stage u = 2 m + half: matrix m (q, k, v), k rows 64h + 32 half + 16 s2 + 4 kq + j  (s2 = 0, 1; j = 0..3)
  const float* const wsrc[3] = {p.Wq, p.Wk, p.Wv};
  float4 wf[2][2][4][NTB];
  auto wload = [&](int u, float4 (&dst)[2][4][NTB]) __attribute__((always_inline)) {
    const float* const base = wsrc[u >> 1] + (size_t)(64 * h + 32 * (u & 1) + 4 * kq) * D + 4 * l15;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb)
          dst[s2][j][tb] = *reinterpret_cast<const float4*>(base + (size_t)(16 * s2 + j) * D + 64 * (w + 4 * tb));
  };
  wload(0, wf[0]);
  __syncthreads();

  // ---- 3. softmax backward of the head, both orientations ----
  f32x4 scT = (f32x4){0.f, 0.f, 0.f, 0.f}, scN = scT, dAT = scT, dAN = scT;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 kf = *reinterpret_cast<const float4*>(ks + l15 * LDH + 16 * s + 4 * kq);
    const float4 qf = *reinterpret_cast<const float4*>(qs + l15 * LDH + 16 * s + 4 * kq);
    const float4 vf = *reinterpret_cast<const float4*>(vs + l15 * LDH + 16 * s + 4 * kq);
    const float4 gf = *reinterpret_cast<const float4*>(gs + l15 * LDH + 16 * s + 4 * kq);
    MFMA16x4(scT, kf, qf)
    MFMA16x4(scN, qf, kf)
    MFMA16x4(dAT, vf, gf)
    MFMA16x4(dAN, gf, vf)
  }
  float dzT[4], aN[4], dzN[4];
  {   // rows = key 4 kq + r, column = query l15
    const float mx = smx[l15], inv = sinv[l15];
    float pr[4], dm[4], dl = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 4 * kq + r;
      const bool ok = l15 < S && key < S;
      float v = scT[r] * 0.125f;
      if (smask[key] != 0.f) v = -1e9f;
      pr[r] = ok ? __expf(v - mx) * inv : 0.f;
      dm[r] = drop_mult(p.drop_att, (uint32_t)((bh * S + l15) * S + key));
      dl += dAT[r] * (pr[r] * dm[r]);
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 4 * kq + r;
      const bool live = l15 < S && key < S && smask[key] == 0.f;
      dzT[r] = live ? pr[r] * (dAT[r] * dm[r] - dl) * 0.125f : 0.f;
    }
    if (kq == 0) sdel[w][l15] = dl;
  }
  __syncthreads();
  {   // rows = query 4 kq + r, column = key l15
    const bool masked = smask[l15] != 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = 4 * kq + r;
      const bool ok = l15 < S && q < S;
      float v = scN[r] * 0.125f;
      if (masked) v = -1e9f;
      const float pr = ok ? __expf(v - smx[q]) * sinv[q] : 0.f;
      const float dm = drop_mult(p.drop_att, (uint32_t)((bh * S + q) * S + l15));
      aN[r] = pr * dm;
      dzN[r] = (ok && !masked) ? pr * (dAN[r] * dm - sdel[w][q]) * 0.125f : 0.f;
    }
  }
  f32x4 dq = (f32x4){0.f, 0.f, 0.f, 0.f}, dk = dq, dv = dq;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = (4 * kq + r) * LDH + 16 * w + l15;
    dv = mfma16(aN[r], gs[o], dv);
    dq = mfma16(dzT[r], ks[o], dq);
    dk = mfma16(dzN[r], qs[o], dk);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int s = 4 * kq + r, o = s * LDH + 16 * w + l15;
    dqs[o] = dq[r]; dks[o] = dk[r]; dvs[o] = dv[r];
    if (s < S) {
      const size_t go = (row0 + s) * D + 64 * h + 16 * w + l15;
      p.dQ[go] = dq[r]; p.dK[go] = dk[r]; p.dV[go] = dv[r];
    }
  }
  __syncthreads();

  // ---- 4. the head's share of dx: column block 64 (w + 4 tb), tile e = columns 4 l15 + e of the block ----
  f32x4 px[NTB][4];
#pragma unroll
  for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
    for (int e = 0; e < 4; ++e) px[tb][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* const asrc[3] = {dqs, dks, dvs};
#pragma unroll
  for (int u = 0; u < 6; ++u) {
    if (u + 1 < 6) wload(u + 1, wf[(u + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float4 a = *reinterpret_cast<const float4*>(asrc[u >> 1] + l15 * LDH + 32 * (u & 1) + 16 * s2 + 4 * kq);
      const float aj[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
          const float4 bw = wf[u & 1][s2][j][tb];
          px[tb][0] = mfma16(aj[j], bw.x, px[tb][0]);
          px[tb][1] = mfma16(aj[j], bw.y, px[tb][1]);
          px[tb][2] = mfma16(aj[j], bw.z, px[tb][2]);
          px[tb][3] = mfma16(aj[j], bw.w, px[tb][3]);
        }
    }
  }
  float* const slot = p.part + (bh * 16) * D;
#pragma unroll
  for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* const dst = slot + (4 * kq + r) * D + 64 * (w + 4 * tb) + 4 * l15;
      st_agent_f2(dst, px[tb][0][r], px[tb][1][r]);
      st_agent_f2(dst + 2, px[tb][2][r], px[tb][3][r]);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) s_last = __hip_atomic_fetch_add(p.cnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == H - 1;
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) __hip_atomic_store(p.cnt + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- 5. finisher: dx = sum_h share_h (+ dz, the residual branch) ----
  constexpr int RG = D == 256 ? 4 : 2;
  const __amdgpu_buffer_rsrc_t prsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.part + (size_t)b * HH * 16 * D), 0, (unsigned)(HH * 16 * D * 4), 0x00020000);
#pragma unroll 1
  for (int g = 0; g < 4 / RG; ++g) {
    u32x4s t[RG][HH][NV];
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
      const int s = min(w + 4 * (g * RG + rr), 15);
#pragma unroll
      for (int hh = 0; hh < HH; ++hh)
#pragma unroll
        for (int i = 0; i < NV; ++i)
          t[rr][hh][i] = __builtin_amdgcn_raw_buffer_load_b128(prsrc, (unsigned)(((hh * 16 + s) * D + 4 * (lane + 64 * i)) * 4), 0, 16 /* sc1 */);
    }
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
      const int s = w + 4 * (g * RG + rr);
      if (s >= S) continue;     // wave-uniform
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int hh = 0; hh < HH; ++hh) {
          v.x += __uint_as_float(t[rr][hh][i].x); v.y += __uint_as_float(t[rr][hh][i].y);
          v.z += __uint_as_float(t[rr][hh][i].z); v.w += __uint_as_float(t[rr][hh][i].w);
        }
        if (resid) {
          const float4 zz = *reinterpret_cast<const float4*>(dzs + s * LDX + 4 * (lane + 64 * i));
          v.x += zz.x; v.y += zz.y; v.z += zz.z; v.w += zz.w;
        }
        *reinterpret_cast<float4*>(p.dx + (row0 + s) * D + 4 * (lane + 64 * i)) = v;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// FeedForward forward (modules.py:328-347: LayerNorm(x + drop(W2 drop(relu(W1 x + b1)) + b2))) of a short sequence in ONE
// launch, the same recipe: grid (4, B) -- a workgroup owns one 256-wide slice of the 4d hidden units of one sample.
//   1. x_b -> LDS; h_j = drop(relu(x_b W1[256j.., :]^T + b1)) [S, 256] (16 x 16 x 4 MFMAs, weight rows streamed as B
//      fragments, 16 bytes a lane feeding 4 MFMAs); h_j goes to the saved block and to LDS.
//   2. the slice's share of the second layer, h_j W2[:, 256j..]^T [S, d], to the workspace; the last workgroup of the sample
//      adds the four shares in slice order, b2, the output dropout, the residual and the LayerNorm.
// Replaces 3 dependent launches (two ~225-workgroup products and the LayerNorm: 13 + 10.5 + 4 us at M = 896).  The saved
// block (h, z) is the general path's, whose backward follows.
// MEASURED NEUTRAL, hence opt-in (MMNAS_SMALL_FFN=1 / mmnas_set_small_ffn(1)): 26.5 us per operator against the 27.5 us of the
// three launches; supernet step 4.666 vs 4.666 ms (profiles/r05_small_ffn_ab.txt).  A workgroup streams 512 KB of weights (twice
// the SelfAtt kernel's) through its CU's 64 B/clk L1 path and runs 2 x 3.4 us of fp32 MFMAs at one wave per SIMD; the general
// path's products run on the split-bf16 pipe at 5x the rate and pay their three launch boundaries instead.
struct FfnSmallK {
  int B, M, flags;      // B = groups of 16 consecutive rows (the operator is row-wise: sample boundaries do not matter)
  const float* x; const float* W1; const float* b1; const float* W2; const float* b2;
  const float* ln_a; const float* ln_b;
  float* h; float* z; float* y;
  float* part; int* cnt;
  DropCfg drop_h, drop_out;
  float eps;
};

template <int D, int HS>     // HS: hidden units per workgroup (256: 4 workgroups per row group; 128: 8, two per CU)
__global__ void __launch_bounds__(256, HS == 256 ? 1 : 2) ffn_small_fwd_kernel(const FfnSmallK p) {
  constexpr int LDX = D + 4, FF = 4 * D, LDHS = HS + 4, NSL = FF / HS;
  constexpr int KC = 64, SPC = KC / 16, NCH1 = D / KC, NCH2 = HS / KC;
  constexpr int NT1 = HS / 64;    // hidden-column tiles per wave (layer 1)
  constexpr int NT2 = D / 64;     // output-column tiles per wave (layer 2)
  __shared__ __attribute__((aligned(16))) float xs[16 * LDX];
  __shared__ __attribute__((aligned(16))) float hs[16 * LDHS];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  const int j = blockIdx.x, b = blockIdx.y;
  const size_t row0 = (size_t)b * 16;
  const int S = min(16, p.M - 16 * b);      // rows of this group

  const float* w1row[NT1];
#pragma unroll
  for (int m = 0; m < NT1; ++m) w1row[m] = p.W1 + (size_t)(HS * j + 64 * m + 16 * w + l15) * D + 4 * kq;
  float4 wb[2][NT1][SPC];
#pragma unroll
  for (int m = 0; m < NT1; ++m)
#pragma unroll
    for (int s = 0; s < SPC; ++s) wb[0][m][s] = *reinterpret_cast<const float4*>(w1row[m] + 16 * s);
  {
    constexpr int F4 = D / 4, N = 16 * F4 / 256;
    float4 xv[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int f = tid + 256 * i, r = f / F4, c4 = f - r * F4;
      const float4 t = *reinterpret_cast<const float4*>(p.x + (row0 + (r < S ? r : 0)) * D + 4 * c4);
      xv[i] = r < S ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int f = tid + 256 * i, r = f / F4, c4 = f - r * F4;
      *reinterpret_cast<float4*>(xs + r * LDX + 4 * c4) = xv[i];
    }
  }
  __syncthreads();

  // ---- layer 1: hidden columns 256 j + 64 m + 16 w + l15 ----
  f32x4 acc[NT1];
#pragma unroll
  for (int m = 0; m < NT1; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH1; ++c) {
    if (c + 1 < NCH1) {
#pragma unroll
      for (int m = 0; m < NT1; ++m)
#pragma unroll
        for (int s = 0; s < SPC; ++s) wb[(c + 1) & 1][m][s] = *reinterpret_cast<const float4*>(w1row[m] + KC * (c + 1) + 16 * s);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < SPC; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(xs + l15 * LDX + KC * c + 16 * s + 4 * kq);
#pragma unroll
      for (int m = 0; m < NT1; ++m) { MFMA16x4(acc[m], a, wb[c & 1][m][s]) }
    }
  }
  // layer-2 fragments of the first chunk: rows n = 64 t + 16 w + l15 of W2, columns 256 j + 16 s + 4 kq .. +3
  const float* w2row[NT2];
#pragma unroll
  for (int t = 0; t < NT2; ++t) w2row[t] = p.W2 + (size_t)(64 * t + 16 * w + l15) * FF + HS * j + 4 * kq;
  float4 wm[2][NT2][SPC];
#pragma unroll
  for (int t = 0; t < NT2; ++t)
#pragma unroll
    for (int s = 0; s < SPC; ++s) wm[0][t][s] = *reinterpret_cast<const float4*>(w2row[t] + 16 * s);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < NT1; ++m) {
    const int col = 64 * m + 16 * w + l15, gcol = HS * j + col;
    const float bv = p.b1 ? p.b1[gcol] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int s = 4 * kq + r;
      float v = fmaxf(acc[m][r] + bv, 0.f);
      if (p.drop_h.thresh) v *= drop_mult(p.drop_h, (uint32_t)(row0 + s) * (uint32_t)FF + (uint32_t)gcol);
      hs[s * LDHS + col] = v;
      if (s < S) p.h[(row0 + s) * FF + gcol] = v;
    }
  }
  __syncthreads();

  // ---- layer 2: this slice's share of the output, columns 64 t + 16 w + l15 ----
  f32x4 pm[NT2];
#pragma unroll
  for (int t = 0; t < NT2; ++t) pm[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH2; ++c) {
    if (c + 1 < NCH2) {
#pragma unroll
      for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int s = 0; s < SPC; ++s) wm[(c + 1) & 1][t][s] = *reinterpret_cast<const float4*>(w2row[t] + KC * (c + 1) + 16 * s);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < SPC; ++s) {
      const float4 a = *reinterpret_cast<const float4*>(hs + l15 * LDHS + KC * c + 16 * s + 4 * kq);
#pragma unroll
      for (int t = 0; t < NT2; ++t) { MFMA16x4(pm[t], a, wm[c & 1][t][s]) }
    }
  }
  const size_t bj = (size_t)b * NSL + j;
  float* const slot = p.part + (bj * 16) * D;
#pragma unroll
  for (int t = 0; t < NT2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) st_agent_f(slot + (4 * kq + r) * D + 64 * t + 16 * w + l15, pm[t][r]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) s_last = __hip_atomic_fetch_add(p.cnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == NSL - 1;
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) __hip_atomic_store(p.cnt + b, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- finisher: z = x + drop(sum_j share_j + b2); y = LN(z) (sa_small_fwd_kernel's, plus the bias) ----
  constexpr int NV = D / 256;
  constexpr int RG = D == 256 ? 4 : 2;
  const bool norm = p.flags & MMNAS_F_NORM, resid = p.flags & MMNAS_F_RESIDUAL;
  const __amdgpu_buffer_rsrc_t prsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(p.part + (size_t)b * NSL * 16 * D), 0, (unsigned)(NSL * 16 * D * 4), 0x00020000);
#pragma unroll 1
  for (int g = 0; g < 4 / RG; ++g) {
    u32x4s t[RG][NSL][NV];
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
      const int s = min(w + 4 * (g * RG + rr), 15);
#pragma unroll
      for (int hh = 0; hh < NSL; ++hh)
#pragma unroll
        for (int i = 0; i < NV; ++i)
          t[rr][hh][i] = __builtin_amdgcn_raw_buffer_load_b128(prsrc, (unsigned)(((hh * 16 + s) * D + 4 * (lane + 64 * i)) * 4), 0, 16 /* sc1 */);
    }
#pragma unroll
    for (int rr = 0; rr < RG; ++rr) {
      const int s = w + 4 * (g * RG + rr);
      if (s >= S) continue;     // wave-uniform
      float v[4 * NV];
#pragma unroll
      for (int i = 0; i < 4 * NV; ++i) v[i] = 0.f;
#pragma unroll
      for (int hh = 0; hh < NSL; ++hh)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          v[4 * i] += __uint_as_float(t[rr][hh][i].x); v[4 * i + 1] += __uint_as_float(t[rr][hh][i].y);
          v[4 * i + 2] += __uint_as_float(t[rr][hh][i].z); v[4 * i + 3] += __uint_as_float(t[rr][hh][i].w);
        }
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 b2v = p.b2 ? *reinterpret_cast<const float4*>(p.b2 + 4 * (lane + 64 * i)) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float bb[4] = {b2v.x, b2v.y, b2v.z, b2v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int col = 4 * (lane + 64 * i) + e;
          float x = v[4 * i + e] + bb[e];
          if (p.drop_out.thresh) x *= drop_mult(p.drop_out, (uint32_t)(row0 + s) * (uint32_t)D + (uint32_t)col);
          if (resid) x += xs[s * LDX + col];
          v[4 * i + e] = x;
          sm += x;
        }
      }
      float* const yr = p.y + (row0 + s) * D;
      if (!norm) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
          *reinterpret_cast<float4*>(yr + 4 * (lane + 64 * i)) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
        continue;
      }
      float* const zr = p.z + (row0 + s) * D;
#pragma unroll
      for (int i = 0; i < NV; ++i)
        *reinterpret_cast<float4*>(zr + 4 * (lane + 64 * i)) = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
      const float mean = wave_sum(sm) / (float)D;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 4 * NV; ++i) { v[i] -= mean; ss += v[i] * v[i]; }
      const float sd = sqrtf(wave_sum(ss) / (float)(D - 1));
      const float invs = 1.0f / (sd + p.eps);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int col = 4 * (lane + 64 * i);
        const float4 av = *reinterpret_cast<const float4*>(p.ln_a + col);
        const float4 bv = *reinterpret_cast<const float4*>(p.ln_b + col);
        *reinterpret_cast<float4*>(yr + col) = make_float4(av.x * v[4 * i] * invs + bv.x, av.y * v[4 * i + 1] * invs + bv.y,
                                                           av.z * v[4 * i + 2] * invs + bv.z, av.w * v[4 * i + 3] * invs + bv.w);
      }
    }
  }
}


static int env_on(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && e[0] ? atoi(e) : dflt;
}
static int g_small_ops = -1;   // -1: not read yet
static bool small_ops_on() {
  if (g_small_ops < 0) g_small_ops = env_on("MMNAS_SMALL_OPS", 1) ? 1 : 0;
  return g_small_ops != 0;
}

// Does the short-sequence kernel take this operator?  (self-attention without relation bias, <= 16 rows per sample,
// heads of 64, model width 256 or 512, at most 256 (sample, head) pairs; MMNAS_SMALL_OPS=0 switches the family off)
bool sa_small_applies(const mmnas_att_op* op) {
  if (!small_ops_on() || op->q_off || op->k_off) return false;   // (packed rows: the general path)
  const int fl = op->flags;
  return (fl & MMNAS_F_SELF) && !(fl & MMNAS_F_REL) && op->Sq == op->Sk && op->Sq <= 16 && op->dh == 64 &&
         op->di == op->d && (op->d == 256 || op->d == 512) && op->xq == op->xkv &&
         op->B * op->H <= 256;   // one round of workgroups (one per CU): with two rounds (B = 64, d = 512: 512 workgroups)
                                 // the general path is faster -- training step 11.61 vs 11.71 ms
}

int sa_small_fwd(const mmnas_att_op* op, float* Q, float* K, float* V, float* att, float* stats, float* z,
                 hipStream_t st) {
  const int fl = op->flags;
  const bool drop = (fl & MMNAS_F_TRAIN) && op->drop_p > 0.f;
  SaSmallK k;
  memset(&k, 0, sizeof(k));
  k.B = op->B; k.S = op->Sq; k.H = op->H; k.flags = fl;
  k.x = op->xq; k.mask = (fl & MMNAS_F_MASK) ? op->mask : nullptr;
  k.Wq = op->Wq; k.Wk = op->Wk; k.Wv = op->Wv; k.Wm = op->Wm; k.ln_a = op->ln_a; k.ln_b = op->ln_b;
  k.Q = Q; k.K = K; k.V = V; k.att = att; k.stats = stats; k.z = z; k.y = op->y;
  k.drop_att = make_drop(drop ? op->drop_p : 0.f, op->seed, 0);
  k.drop_out = make_drop(drop ? op->drop_p : 0.f, op->seed, 1);
  k.eps = op->eps;
  size_t wsf = 0; int ncnt = 0;
  int rc = sk_workspace(st, &k.part, &wsf, &k.cnt, &ncnt);
  if (rc) return rc;
  MMNAS_REQUIRE((size_t)op->B * op->H * 16 * op->d <= wsf && op->B <= ncnt, MMNAS_E_SHAPE,
                "sa_small_fwd: B=%d H=%d d=%d exceeds the hand-off workspace", op->B, op->H, op->d);
  const double M = (double)op->B * op->Sq, d = op->d;
  ProfScope ps(MMNAS_K_SMALL, 2.0 * M * d * d * 4.0 + 4.0 * M * op->Sq * d, 4.0 * (4.0 * d * d + 7.0 * M * d), st, "sa_small_fwd");
  dim3 grid(op->H, op->B), block(256);
  if (op->d == 256) MMNAS_LAUNCH(sa_small_fwd_kernel<256>, grid, block, 0, st, k);
  else MMNAS_LAUNCH(sa_small_fwd_kernel<512>, grid, block, 0, st, k);
  return check_launch("sa_small_fwd");
}

static int g_small_bwd = -1;
static bool small_bwd_on() {
  if (g_small_bwd < 0) g_small_bwd = env_on("MMNAS_SMALL_BWD", 1) ? 1 : 0;
  return g_small_bwd != 0;
}

// Does the one-launch backward take this operator?  (What the forward kernel takes -- either backward may follow either
// forward, the saved block is the same -- and a LayerNorm partial row per sample must fit the operator's LayerNorm scratch.)
bool sa_small_bwd_applies(const mmnas_att_op* op) {
  if (!small_bwd_on() || !sa_small_applies(op)) return false;
  if ((op->flags & MMNAS_F_NORM) && (size_t)op->B * 3 * op->d > mmnas_layernorm_bwd_ws_floats(op->B * op->Sq, op->d)) return false;
  return true;
}

// dt_out: where the gradient wrt the core's output goes (NULL: it IS dy -- no LayerNorm, no dropout); lnpart: [B][3][d]
int sa_small_bwd(const mmnas_att_op* op, const float* Q, const float* K, const float* V, const float* stats, const float* z,
                 float* dt_out, float* dQ, float* dK, float* dV, float* lnpart, hipStream_t st) {
  const int fl = op->flags;
  const bool drop = (fl & MMNAS_F_TRAIN) && op->drop_p > 0.f;
  SaSmallBwdK k;
  memset(&k, 0, sizeof(k));
  k.B = op->B; k.S = op->Sq; k.H = op->H; k.flags = fl;
  k.dy = op->dy; k.z = z; k.ln_a = op->ln_a; k.mask = (fl & MMNAS_F_MASK) ? op->mask : nullptr;
  k.Wq = op->Wq; k.Wk = op->Wk; k.Wv = op->Wv; k.Wm = op->Wm;
  k.Q = Q; k.K = K; k.V = V; k.stats = stats;
  k.dt = dt_out; k.dQ = dQ; k.dK = dK; k.dV = dV; k.dx = op->dxq; k.lnpart = lnpart;
  k.drop_att = make_drop(drop ? op->drop_p : 0.f, op->seed, 0);
  k.drop_out = make_drop(drop ? op->drop_p : 0.f, op->seed, 1);
  k.eps = op->eps;
  size_t wsf = 0; int ncnt = 0;
  int rc = sk_workspace(st, &k.part, &wsf, &k.cnt, &ncnt);
  if (rc) return rc;
  MMNAS_REQUIRE((size_t)op->B * op->H * 16 * op->d <= wsf && op->B <= ncnt, MMNAS_E_SHAPE,
                "sa_small_bwd: B=%d H=%d d=%d exceeds the hand-off workspace", op->B, op->H, op->d);
  const double M = (double)op->B * op->Sq, d = op->d;
  ProfScope ps(MMNAS_K_SMALL, 2.0 * M * d * d * 4.0 + 10.0 * M * op->Sq * d, 4.0 * (4.0 * d * d + 9.0 * M * d), st, "sa_small_bwd");
  dim3 grid(op->H, op->B), block(256);
  if (op->d == 256) MMNAS_LAUNCH(sa_small_bwd_kernel<256>, grid, block, 0, st, k);
  else MMNAS_LAUNCH(sa_small_bwd_kernel<512>, grid, block, 0, st, k);
  return check_launch("sa_small_bwd");
}

// Does the one-launch FeedForward forward take this operator?  (two layers d -> 4d -> d with d = 256 and at most 1024 rows:
// 4 workgroups per group of 16 rows, one round of <= 256)
static int g_small_ffn = -1;
static bool small_ffn_on() {     // default OFF: measured neutral against the three launches it replaces (see the header comment)
  if (g_small_ffn < 0) g_small_ffn = env_on("MMNAS_SMALL_FFN", 0);      // 1: four slices of 256 hidden units; 2: eight of 128
  return g_small_ffn != 0;
}
bool ffn_small_applies(const mmnas_mlp_op* op) {
  if (!small_ops_on() || !small_ffn_on()) return false;
  return op->nl == 2 && op->dims[0] == 256 && op->dims[1] == 1024 && op->dims[2] == 256 && op->M <= 1024 && op->W[0] && op->W[1];
}

int ffn_small_fwd(const mmnas_mlp_op* op, float* h, float* z, hipStream_t st) {
  const int fl = op->flags;
  const bool drop = (fl & MMNAS_F_TRAIN) && op->drop_p > 0.f;
  FfnSmallK k;
  memset(&k, 0, sizeof(k));
  k.B = (op->M + 15) / 16; k.M = op->M; k.flags = fl;
  k.x = op->x; k.W1 = op->W[0]; k.b1 = op->b[0]; k.W2 = op->W[1]; k.b2 = op->b[1];
  k.ln_a = op->ln_a; k.ln_b = op->ln_b; k.h = h; k.z = z; k.y = op->y;
  k.drop_h = make_drop(drop ? op->drop_p : 0.f, op->seed, 0);
  k.drop_out = make_drop(drop ? op->drop_p : 0.f, op->seed, 1);
  k.eps = op->eps;
  size_t wsf = 0; int ncnt = 0;
  int rc = sk_workspace(st, &k.part, &wsf, &k.cnt, &ncnt);
  if (rc) return rc;
  const int nsl = g_small_ffn == 2 ? 8 : 4;
  MMNAS_REQUIRE((size_t)k.B * nsl * 16 * 256 <= wsf && k.B <= ncnt, MMNAS_E_SHAPE, "ffn_small_fwd: M=%d exceeds the hand-off workspace", op->M);
  const double M = op->M;
  ProfScope ps(MMNAS_K_SMALL, 2.0 * M * 256.0 * 1024.0 * 2.0, 4.0 * (2.0 * 256.0 * 1024.0 + M * (3.0 * 256.0 + 1024.0)), st, "ffn_small_fwd");
  if (nsl == 8) MMNAS_LAUNCH((ffn_small_fwd_kernel<256, 128>), dim3(8, k.B), dim3(256), 0, st, k);
  else MMNAS_LAUNCH((ffn_small_fwd_kernel<256, 256>), dim3(4, k.B), dim3(256), 0, st, k);
  return check_launch("ffn_small_fwd");
}

}  // namespace mmnas

extern "C" int mmnas_set_small_ffn(int on) {
  const int prev = mmnas::small_ffn_on() ? 1 : 0;
  mmnas::g_small_ffn = on < 0 ? 0 : (on > 2 ? 2 : on);
  return prev;
}

extern "C" int mmnas_set_small_bwd(int on) {
  const int prev = mmnas::small_bwd_on() ? 1 : 0;
  mmnas::g_small_bwd = on ? 1 : 0;
  return prev;
}

extern "C" int mmnas_set_small_ops(int on) {
  const int prev = mmnas::small_ops_on() ? 1 : 0;
  mmnas::g_small_ops = on ? 1 : 0;
  return prev;
}

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

}  // namespace mmnas

extern "C" int mmnas_set_small_ops(int on) {
  const int prev = mmnas::small_ops_on() ? 1 : 0;
  mmnas::g_small_ops = on ? 1 : 0;
  return prev;
}

// Single-layer LSTM over a short sequence as ONE persistent launch per direction (forward pass / backward pass).
//
// Replaces the MIOpen path behind nn.LSTM in the nets' language stem (hygr_vqa.py:86-92 construction, :106-107 call):
// for 14 time steps of a 64-row problem MIOpen issues a GEMM + a pointwise kernel per step and direction plus
// weight-buffer copies -- ~110 dependent launches of 3-8 us each, ~0.6 ms of a 7 ms supernet step.
//
// Decomposition (MI355X): the recurrent matrix is split over workgroups by HIDDEN UNIT, so that every workgroup keeps
// its slice of W_hh in REGISTERS (MFMA A-operand fragments) for the whole sequence and only the 64 x H state vector
// travels between workgroups once per step:
//   forward : workgroup = 8 units = 32 gate rows (native nn.LSTM order: row g*H + u, g in i,f,g,o); 4 waves split the
//             reduction over H; per step 2 x (H/8) v_mfma_f32_32x32x2_f32 per wave, a 32 KB LDS reduction across the
//             waves, then the gate arithmetic with the cell state held in registers.
//   backward: workgroup = 16 units; dh_t = dout_t + dG_{t+1} W_hh needs W_hh[:, units] (reduction over the 4H gate
//             rows): 8 waves split it, v_mfma_f32_16x16x4_f32 (16-unit tiles: no padding), running dc in registers.
// Step hand-off between workgroups (h_t forward, dG_t backward): the payload is stored write-through (sc1), every
// storing wave drains its stores, one lane adds to a per-sample-block arrival counter (agent scope); one lane polls
// the counter (relaxed, bounded), workgroup barrier, then EVERY load of the handed-off state is an sc1 (L1-bypassing)
// 16-byte buffer load -- the R1 recipe of the CDNA programming guide with its all-sc1-loads form in place of the
// acquire fence, placement-independent.  Samples are independent: blocks of 64 samples have their own
// counter and never wait for each other.  Counters are zeroed by a memset node in front of every launch.
#include <string.h>
#include <map>
#include <mutex>
#include <utility>
#include "common.h"

namespace mmnas {

constexpr int LSTM_MAX_SB = 60;

struct LstmSeqK {
  const float* xp;      // [B,T,4H] input projection x W_ih^T + b_ih (native gate order)
  const float* bhh;     // [4H] b_hh (added in the gate arithmetic)
  const float* Whh;     // [4H,H]
  float* Hprev;         // [B,T,H]  Hprev[b][t] = h_{t-1} (slice t = 0 zero): exchange buffer + saved for dW_hh
  float* Cs;            // [B,T,H]  c_t
  float* Gall;          // [B,T,4H] activated gates
  float* out;           // [B,T,H]  h_t
  const float* dout;    // [B,T,H]
  float* DG;            // [B,T,4H] pre-activation gradients: exchange buffer + operand of the weight gradients
  unsigned* cnt;        // [LSTM_MAX_SB] arrival counters (one per block of 64 samples), then one timeout word
  int T, B, H, nsb;
};

__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16-byte load that bypasses this CU's L1 (sc1): with EVERY load of the handed-off state done this way (and every
// store of it write-through, drained before the arrival), the consumer needs no acquire fence after the poll -- the
// guide's hand-off table, first row -- which takes ~1.7 us off every time step.
typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_sc1_f4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
  const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16 /* sc1 */);
  float4 f;
  f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
  return f;
}

// Arrive at / wait for step barrier `target` arrivals on counter c.  Called by every thread of the workgroup.
__device__ __forceinline__ void step_barrier(unsigned* c, unsigned target, unsigned* tmo) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its write-through stores have landed
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 22)) {   // a workgroup of this launch is not resident: give up loudly instead of hanging
        __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (no instruction: keeps the compiler from hoisting loads above the poll)
  }
  __syncthreads();
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// ---------------------------------------------------------------------------------------------- forward
template <int H, int SB>   // SB: samples per workgroup (32 or 64)
__global__ void __launch_bounds__(256, 1) lstm_seq_fwd_kernel(const LstmSeqK p) {
  constexpr int U = 8, R = 4 * U;          // units / gate rows per workgroup
  constexpr int NJ = SB / 32;              // 32-sample MFMA tiles
  constexpr int NP = U * SB / 256;         // (unit, sample) pairs per thread in the gate arithmetic
  constexpr int KW = H / 4;                // reduction range of one wave
  constexpr int NS = KW / 8;               // 8-wide k groups per wave (one float4 per lane and group)
  static_assert(H % 32 == 0, "hidden size must be a multiple of 32");
  __shared__ float part[4][R][SB];         // per-wave partial pre-activations [gate row][sample]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int u0 = blockIdx.x * U, n0 = blockIdx.y * SB;
  const int T = p.T, B = p.B;
  const unsigned nub = gridDim.x;
  unsigned* const cnt = p.cnt + blockIdx.y;
  unsigned* const tmo = p.cnt + LSTM_MAX_SB;

  // W_hh slice as MFMA A fragments: local row l31 = gate (l31 >> 3), unit (l31 & 7); k = wave*KW + 8s + 4hh + 0..3
  float4 wa[NS];
  {
    const float* w = p.Whh + (size_t)((l31 >> 3) * H + u0 + (l31 & 7)) * H + wave * KW + 4 * hh;
#pragma unroll
    for (int s = 0; s < NS; ++s) wa[s] = *reinterpret_cast<const float4*>(w + 8 * s);
  }
  const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.Hprev, 0, (unsigned)((size_t)B * T * H * 4), 0x00020000);
  // gate arithmetic: thread owns NP (unit, sample) pairs: pair index tid + 256 i -> unit = idx / SB, sample = idx % SB
  float cst[NP];
  float bh[NP][4];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    cst[i] = 0.f;
    const int u = (tid + 256 * i) / SB;
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[i][g] = p.bhh ? p.bhh[g * H + u0 + u] : 0.f;
  }

  for (int t = 0; t < T; ++t) {
    // input-projection terms of this step (independent of the recurrence: issued first)
    float xv[NP][4];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + 256 * i, u = idx / SB, n = n0 + (idx % SB);
      const bool ok = n < B;
      const float* x = p.xp + ((size_t)(ok ? n : 0) * T + t) * (4 * H) + u0 + u;
#pragma unroll
      for (int g = 0; g < 4; ++g) xv[i][g] = ok ? x[g * H] : 0.f;
    }
    f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    if (t > 0) {   // h_{-1} = 0
      float4 hb[NJ][NS];   // every load of the step in flight before the first MFMA
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = n0 + 32 * j + l31;
        const unsigned off = n < B ? (unsigned)((((size_t)n * T + t) * H + wave * KW + 4 * hh) * 4) : ~0u;   // out of range: zeros
#pragma unroll
        for (int s = 0; s < NS; ++s) hb[j][s] = ld_sc1_f4(hrsrc, off == ~0u ? ~0u : off + 32u * s);
      }
      __builtin_amdgcn_sched_barrier(0);   // (the scheduler would otherwise sink the loads between the MFMAs to save registers)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          acc[j] = mfma32(wa[s].x, hb[j][s].x, acc[j]);
          acc[j] = mfma32(wa[s].y, hb[j][s].y, acc[j]);
          acc[j] = mfma32(wa[s].z, hb[j][s].z, acc[j]);
          acc[j] = mfma32(wa[s].w, hb[j][s].w, acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) part[wave][acc_row(r, hh)][32 * j + l31] = acc[j][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + 256 * i, u = idx / SB, nl = idx % SB, n = n0 + nl;
      float pre[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        pre[g] = ((part[0][g * U + u][nl] + part[1][g * U + u][nl]) + (part[2][g * U + u][nl] + part[3][g * U + u][nl])) + (xv[i][g] + bh[i][g]);
      const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = tanhf(pre[2]), go = sigmoidf_(pre[3]);
      const float c = gf * cst[i] + gi * gg;
      const float hv = go * tanhf(c);
      cst[i] = c;
      if (n < B) {
        const size_t row = (size_t)n * T + t;
        float* ga = p.Gall + row * (4 * H) + u0 + u;
        ga[0] = gi; ga[H] = gf; ga[2 * H] = gg; ga[3 * H] = go;
        p.Cs[row * H + u0 + u] = c;
        p.out[row * H + u0 + u] = hv;
        if (t + 1 < T) st_sc1(p.Hprev + (row + 1) * H + u0 + u, hv);   // the next step's input, read by every workgroup
        if (t == 0) p.Hprev[row * H + u0 + u] = 0.f;
      }
    }
    if (t + 1 < T) step_barrier(cnt, (unsigned)(t + 1) * nub, tmo);
  }
  // A step barrier of this launch gave up (a workgroup was not resident): the states behind it are garbage.  Make that
  // visible where nobody polls the timeout word -- a NaN in the output sequence reaches the loss.
  if (tid == 0 && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
    p.out[((size_t)n0 * T + (T - 1)) * H + u0] = __builtin_nanf("");
}

// ---------------------------------------------------------------------------------------------- backward
template <int H, int SB>   // SB: samples per workgroup (16, 32 or 64)
__global__ void __launch_bounds__(512, 1) lstm_seq_bwd_kernel(const LstmSeqK p) {
  constexpr int U = 16;
  constexpr int NJ = SB / 16;                          // 16-sample MFMA tiles
  constexpr int NP = (U * SB + 511) / 512;             // (unit, sample) pairs per thread (SB = 16: threads >= 256 have none)
  constexpr int K = 4 * H, KW = K / 8, NS = KW / 16;   // 8 waves split the reduction over the 4H gate rows
  static_assert(H % 32 == 0, "hidden size must be a multiple of 32");
  __shared__ float part[8][U][SB];                     // per-wave partial dh [unit][sample]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  const int u0 = blockIdx.x * U, n0 = blockIdx.y * SB;
  const int T = p.T, B = p.B;
  const unsigned nub = gridDim.x;
  unsigned* const cnt = p.cnt + blockIdx.y;
  unsigned* const tmo = p.cnt + LSTM_MAX_SB;

  // W_hh[:, units] as A fragments of the 16x16x4 MFMA: row = unit l15, k = wave*KW + 16 s + 4 kq + 0..3
  float wa[NS][4];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int w = 0; w < 4; ++w) wa[s][w] = p.Whh[(size_t)(wave * KW + 16 * s + 4 * kq + w) * H + u0 + l15];
  const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.DG, 0, (unsigned)((size_t)B * T * K * 4), 0x00020000);
  float dcr[NP];   // running cell-state gradient of the thread's (unit, sample) pairs
#pragma unroll
  for (int i = 0; i < NP; ++i) dcr[i] = 0.f;

  for (int t = T - 1; t >= 0; --t) {
    // saved activations of this step (independent of the recurrence: issued first)
    float gate[NP][4], cc[NP], cp[NP], dov[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + 512 * i, u = (idx / SB) % U, n = n0 + (idx % SB);
      const bool ok = n < B && idx < U * SB;
      const size_t row = (size_t)(ok ? n : 0) * T + t;
      const float* ga = p.Gall + row * (4 * H) + u0 + u;
#pragma unroll
      for (int g = 0; g < 4; ++g) gate[i][g] = ok ? ga[g * H] : 0.f;
      cc[i] = ok ? p.Cs[row * H + u0 + u] : 0.f;
      cp[i] = (ok && t > 0) ? p.Cs[(row - 1) * H + u0 + u] : 0.f;
      dov[i] = ok ? p.dout[row * H + u0 + u] : 0.f;
    }
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t + 1 < T) {   // dG_T = 0
      float4 bf[NJ][NS];   // every load of the step in flight before the first MFMA
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int n = n0 + 16 * j + l15;
        const unsigned off = n < B ? (unsigned)((((size_t)n * T + t + 1) * K + wave * KW + 4 * kq) * 4) : ~0u;
#pragma unroll
        for (int s = 0; s < NS; ++s) bf[j][s] = ld_sc1_f4(grsrc, off == ~0u ? ~0u : off + 64u * s);
      }
      __builtin_amdgcn_sched_barrier(0);   // (see the forward kernel)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s][0], bf[j][s].x, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s][1], bf[j][s].y, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s][2], bf[j][s].z, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s][3], bf[j][s].w, acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[wave][4 * kq + r][16 * j + l15] = acc[j][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + 512 * i, u = (idx / SB) % U, nl = idx % SB, n = n0 + nl;
      float dh = dov[i];
#pragma unroll
      for (int w = 0; w < 8; ++w) dh += part[w][u][nl];
      const float gi = gate[i][0], gf = gate[i][1], gg = gate[i][2], go = gate[i][3];
      const float tc = tanhf(cc[i]);
      const float dc = dcr[i] + dh * go * (1.0f - tc * tc);
      dcr[i] = dc * gf;
      if (n < B && idx < U * SB) {
        float* dg = p.DG + ((size_t)n * T + t) * K + u0 + u;
        st_sc1(dg, dc * gg * gi * (1.0f - gi));             // d pre_i
        st_sc1(dg + H, dc * cp[i] * gf * (1.0f - gf));      // d pre_f
        st_sc1(dg + 2 * H, dc * gi * (1.0f - gg * gg));     // d pre_g
        st_sc1(dg + 3 * H, dh * tc * go * (1.0f - go));     // d pre_o
      }
    }
    if (t > 0) step_barrier(cnt, (unsigned)(T - t) * nub, tmo);
  }
  // (as in the forward kernel: a timed-out launch poisons the gate gradients, i.e. every LSTM / embedding gradient)
  if (tid == 0 && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
    st_sc1(p.DG + (size_t)n0 * T * K + u0, __builtin_nanf(""));
}

// ---- per-(device, stream) sync words: LSTM_MAX_SB counters + a timeout word (64 words) ----
struct LstmSync { unsigned* words; };
static std::mutex g_ls_mu;
static std::map<std::pair<int, hipStream_t>, LstmSync> g_ls;

static int lstm_sync_words(hipStream_t st, unsigned** out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("lstm: hipGetDevice failed"); return MMNAS_E_LAUNCH; }
  std::lock_guard<std::mutex> lk(g_ls_mu);
  auto key = std::make_pair(dev, st);
  auto it = g_ls.find(key);
  if (it == g_ls.end()) {
    LstmSync s{nullptr};
    if (hipMalloc((void**)&s.words, 256) != hipSuccess) { set_error("lstm: cannot allocate the step counters"); return MMNAS_E_LAUNCH; }
    it = g_ls.emplace(key, s).first;
  }
  *out = it->second.words;
  return MMNAS_OK;
}

// Samples per workgroup: the state every workgroup reads per step is (its samples) x (all units / all gate columns), so
// narrower sample blocks divide the per-CU hand-off volume (256 KB per step and workgroup in the backward pass at 64
// samples) and multiply the CUs that share the work; the blocks are independent, each has its own arrival counter.
constexpr int SB_FWD = 32, SB_BWD = 16;
template <int H>
static void launch_fwd(const LstmSeqK& k, dim3 grid, hipStream_t st) {
  MMNAS_LAUNCH((lstm_seq_fwd_kernel<H, SB_FWD>), grid, dim3(256), 0, st, k);
}
template <int H>
static void launch_bwd(const LstmSeqK& k, dim3 grid, hipStream_t st) {
  MMNAS_LAUNCH((lstm_seq_bwd_kernel<H, SB_BWD>), grid, dim3(512), 0, st, k);
}

}  // namespace mmnas

using namespace mmnas;

namespace mmnas {
// Workgroups of the two kernels the device holds at once, per hidden size (0: no device / query failed).  The step
// barriers spin on the other unit workgroups of a sample block, so a grid beyond this number would rely on dispatch
// order: the entry points cut the batch into launches of whole sample blocks that fit (the blocks are independent; the
// ITM batch of 160 at H = 512 is 320 workgroups per pass).  One block per CU is taken off
// the occupancy query when it reports more than one (ROCm 7.2 over-reports by one for SGPR-heavy kernels,
// MI355X_MICROARCH.md "Residency and cooperative launch").
template <int H>
static void resident_query(int* fwd, int* bwd) {
  static int cache[2] = {-1, -1};
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (cache[0] < 0) {
    cache[0] = cache[1] = 0;
    int dev = 0, cus = 0, nf = 0, nb = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, lstm_seq_fwd_kernel<H, SB_FWD>, 256, 0) == hipSuccess &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm_seq_bwd_kernel<H, SB_BWD>, 512, 0) == hipSuccess) {
      cache[0] = (nf > 1 ? nf - 1 : nf) * cus;
      cache[1] = (nb > 1 ? nb - 1 : nb) * cus;
    } else {
      (void)hipGetLastError();
    }
  }
  *fwd = cache[0]; *bwd = cache[1];
}
}  // namespace mmnas

extern "C" int mmnas_lstm_seq_supported(int H, int B) {
  if (!((H == 64 || H == 128 || H == 256 || H == 512) && B >= 1 && (B + SB_BWD - 1) / SB_BWD <= LSTM_MAX_SB)) return 0;
  int rf = 0, rb = 0;
  switch (H) {
    case 64: resident_query<64>(&rf, &rb); break;
    case 128: resident_query<128>(&rf, &rb); break;
    case 256: resident_query<256>(&rf, &rb); break;
    default: resident_query<512>(&rf, &rb); break;
  }
  // (no device: nothing can be launched anyway.)  One sample block's workgroups must be resident together; more samples
  // than fit one launch are run as several (lstm_chunk)
  return rf >= H / 8 && rb >= H / 16;
}

namespace mmnas {
// samples per launch: whole sample blocks of SB whose (units x blocks) grid is resident at once
static int lstm_chunk(int H, int B, bool bwd) {
  int rf = 0, rb = 0;
  switch (H) {
    case 64: resident_query<64>(&rf, &rb); break;
    case 128: resident_query<128>(&rf, &rb); break;
    case 256: resident_query<256>(&rf, &rb); break;
    default: resident_query<512>(&rf, &rb); break;
  }
  const int per_block = bwd ? H / 16 : H / 8, sb = bwd ? SB_BWD : SB_FWD;
  int blocks = (bwd ? rb : rf) / per_block;
  if (blocks < 1) blocks = 1;
  if (blocks > LSTM_MAX_SB) blocks = LSTM_MAX_SB;
  const int n = blocks * sb;
  return n < B ? n : B;
}
}  // namespace mmnas

extern "C" int mmnas_lstm_seq_fwd(const float* xp, const float* bhh, const float* Whh, float* Hprev, float* Cs, float* Gall,
                                  float* out, int T, int B, int H, void* stream) {
  MMNAS_REQUIRE(xp && Whh && Hprev && Cs && Gall && out, MMNAS_E_ARG, "lstm_seq_fwd: null pointer");
  MMNAS_REQUIRE(T >= 1 && mmnas_lstm_seq_supported(H, B), MMNAS_E_SHAPE, "lstm_seq_fwd: T=%d B=%d H=%d unsupported", T, B, H);
  MMNAS_REQUIRE((((uintptr_t)Whh | (uintptr_t)Hprev) & 15) == 0, MMNAS_E_ARG, "lstm_seq_fwd: W_hh / state buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  unsigned* words = nullptr;
  int rc = lstm_sync_words(st, &words);
  if (rc) return rc;
  const int chunk = lstm_chunk(H, B, false);
  for (int b0 = 0; b0 < B; b0 += chunk) {   // (one launch for every batch of the VQA / VGD configurations)
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    const size_t r0 = (size_t)b0 * T;
    LstmSeqK k;
    memset(&k, 0, sizeof(k));
    k.xp = xp + r0 * 4 * H; k.bhh = bhh; k.Whh = Whh; k.Hprev = Hprev + r0 * H; k.Cs = Cs + r0 * H; k.Gall = Gall + r0 * 4 * H;
    k.out = out + r0 * H;
    k.T = T; k.B = nb; k.H = H; k.nsb = (nb + SB_FWD - 1) / SB_FWD;
    k.cnt = words;
    // counters and, on the first launch only, the timeout word (a later launch must not clear an earlier one's report)
    if (hipMemsetAsync(k.cnt, 0, b0 == 0 ? 256 : LSTM_MAX_SB * sizeof(unsigned), st) != hipSuccess) { set_error("lstm_seq_fwd: memset failed"); return MMNAS_E_LAUNCH; }
    const dim3 grid(H / 8, k.nsb);
    ProfScope ps(MMNAS_K_LSTM, 2.0 * T * nb * 4.0 * H * H, 4.0 * (4.0 * H * H + 7.0 * T * nb * H), st, "lstm_seq_fwd");
    switch (H) {
      case 64: launch_fwd<64>(k, grid, st); break;
      case 128: launch_fwd<128>(k, grid, st); break;
      case 256: launch_fwd<256>(k, grid, st); break;
      default: launch_fwd<512>(k, grid, st); break;
    }
  }
  return check_launch("lstm_seq_fwd");
}

extern "C" int mmnas_lstm_seq_bwd(const float* dout, const float* Whh, const float* Cs, const float* Gall, float* DG, int T,
                                  int B, int H, void* stream) {
  MMNAS_REQUIRE(dout && Whh && Cs && Gall && DG, MMNAS_E_ARG, "lstm_seq_bwd: null pointer");
  MMNAS_REQUIRE(T >= 1 && mmnas_lstm_seq_supported(H, B), MMNAS_E_SHAPE, "lstm_seq_bwd: T=%d B=%d H=%d unsupported", T, B, H);
  MMNAS_REQUIRE(((uintptr_t)DG & 15) == 0, MMNAS_E_ARG, "lstm_seq_bwd: DG must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  unsigned* words = nullptr;
  int rc = lstm_sync_words(st, &words);
  if (rc) return rc;
  const int chunk = lstm_chunk(H, B, true);
  for (int b0 = 0; b0 < B; b0 += chunk) {
    const int nb = B - b0 < chunk ? B - b0 : chunk;
    const size_t r0 = (size_t)b0 * T;
    LstmSeqK k;
    memset(&k, 0, sizeof(k));
    k.Whh = Whh; k.Cs = const_cast<float*>(Cs) + r0 * H; k.Gall = const_cast<float*>(Gall) + r0 * 4 * H; k.dout = dout + r0 * H;
    k.DG = DG + r0 * 4 * H;
    k.T = T; k.B = nb; k.H = H; k.nsb = (nb + SB_BWD - 1) / SB_BWD;
    k.cnt = words;
    if (hipMemsetAsync(k.cnt, 0, b0 == 0 ? 256 : LSTM_MAX_SB * sizeof(unsigned), st) != hipSuccess) { set_error("lstm_seq_bwd: memset failed"); return MMNAS_E_LAUNCH; }
    const dim3 grid(H / 16, k.nsb);
    ProfScope ps(MMNAS_K_LSTM, 2.0 * T * nb * 4.0 * H * H, 4.0 * (4.0 * H * H + 12.0 * T * nb * H), st, "lstm_seq_bwd");
    switch (H) {
      case 64: launch_bwd<64>(k, grid, st); break;
      case 128: launch_bwd<128>(k, grid, st); break;
      case 256: launch_bwd<256>(k, grid, st); break;
      default: launch_bwd<512>(k, grid, st); break;
    }
  }
  return check_launch("lstm_seq_bwd");
}

// 1 when a step barrier of the last mmnas_lstm_seq_* launch on `stream` gave up waiting (a workgroup was not resident);
// synchronises the stream.  Tests and debugging: results of such a launch are garbage.
extern "C" int mmnas_lstm_seq_timed_out(void* stream) {
  unsigned* w = nullptr;
  if (lstm_sync_words((hipStream_t)stream, &w)) return -1;
  unsigned host[64];
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return -1;
  if (hipMemcpy(host, w, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return host[LSTM_MAX_SB] != 0 ? 1 : 0;   // (the timeout word sits behind nsb <= LSTM_MAX_SB counters)
}

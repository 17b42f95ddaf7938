// Attention core of MHAtt.att / RelMHAtt.forward (modules.py:191-199, 231-241) for every
// (batch, head) pair:  Z = Q K^T / sqrt(dh) (+ rel bias);  Z[mask] = -1e9;  P = softmax(Z);
// A = dropout(P);  O = A V  -- and its backward, recomputing P from the saved row statistics.
//
// MI355X mapping (all products on v_mfma_f32_32x32x2_f32, exact fp32):
//   * forward / dQ kernels are QUERY-owner: a wave owns 32 queries and computes the TRANSPOSED
//     score tile S^T = K Q^T, so a lane holds one query column and its registers hold the keys:
//     the softmax row reduction is in-register plus one cross-half shuffle, and the probability
//     tile is already in MFMA A-operand form for O = A V (accumulator-as-operand, no LDS trip).
//   * the dK/dV kernel is KEY-owner (a wave owns 32 keys, sweeps the queries): S = Q K^T puts the
//     key on the lane, so A^T dO and dZ^T Q again take the accumulator tile as the A operand.
//     Recomputing S in both kernels (7 products instead of 5) avoids any transpose through LDS or
//     cross-workgroup reduction; the core is ~10 % of an attention operator's FLOPs.
//   * Q/K/V/dO head slices are staged through LDS as [row][dh_chunk + 4] (ds_read_b128 fragment
//     reads conflict-free), head dim processed in chunks of <= 64 so dh = 16..256 share one code.
//   * the relation bias is stored key-major [B,H,Sk,Sq]: query-owner lanes read it coalesced.
// Row statistics: stats[b,h,q] = (row max, 1 / row sum) -- two floats, because a fully masked row
// has max = -1e9 where a single log-sum-exp float would lose log(sum).
#include <string.h>
#include "common.h"

namespace mmnas {

struct MhaK {
  int B, H, Sq, Sk, dh, nch;
  int ldq, ldk, ldv, ldo;
  const float* Q; const float* K; const float* V;
  const uint8_t* mask; const float* biasT;
  float* O; float* stats;
  DropCfg drop; float scale;
  const float* dO; float* dQ; float* dK; float* dV; float* dbiasT; float* delta;
  // PACKED rows (sequences of different lengths stored back to back, no padding rows): batch b owns rows
  // qoff[b] .. qoff[b+1] of Q / O / dO / dQ (koff: of K / V / dK / dV); Sq / Sk are then the MAXIMUM lengths -- the grid and
  // the strides of stats / delta / bias / the dropout index.  NULL: row b * Sq + q as ever.
  const int* qoff; const int* koff;
};

// rows x DHC floats from global (row stride ld) into LDS [rows][DHC+4]; rows >= nvalid are zero
// Two phases so that every 16-byte load of a tile (or of several tiles) is in flight before the first
// LDS write waits for one: the one-loop form compiled to a load -> vmcnt(0) -> ds_write chain per
// iteration, i.e. 16-24 serialized memory round trips per workgroup.  Rows beyond nvalid read row 0
// (always valid) and are zeroed by a select, so there is no branch around a load.
template <int ROWS, int DHC, int NT>
struct TileRegs { float4 v[(ROWS * (DHC / 4) + NT - 1) / NT]; };

template <int ROWS, int DHC, int NT>
__device__ __forceinline__ void tile_fetch(TileRegs<ROWS, DHC, NT>& t, const float* __restrict__ src, int nvalid,
                                           int ld, int tid) {
  constexpr int F4 = DHC / 4, N = (ROWS * F4 + NT - 1) / NT;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int f = tid + i * NT;
    const int r = f / F4, c4 = f - r * F4;
    const bool ok = (ROWS * F4 % NT == 0 || f < ROWS * F4) && r < nvalid;
    const float4 x = *reinterpret_cast<const float4*>(src + (size_t)(ok ? r : 0) * ld + 4 * c4);
    t.v[i] = ok ? x : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <int ROWS, int DHC, int NT>
__device__ __forceinline__ void tile_store(const TileRegs<ROWS, DHC, NT>& t, float* __restrict__ dst, int tid) {
  constexpr int F4 = DHC / 4, LD = DHC + 4, N = (ROWS * F4 + NT - 1) / NT;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int f = tid + i * NT;
    const int r = f / F4, c4 = f - r * F4;
    if (ROWS * F4 % NT == 0 || f < ROWS * F4) *reinterpret_cast<float4*>(dst + r * LD + 4 * c4) = t.v[i];
  }
}

template <int ROWS, int DHC, int NT>
__device__ __forceinline__ void load_tile(float* __restrict__ dst, const float* __restrict__ src, int nvalid,
                                          int ld, int tid) {
  TileRegs<ROWS, DHC, NT> t;
  tile_fetch<ROWS, DHC, NT>(t, src, nvalid, ld, tid);
  tile_store<ROWS, DHC, NT>(t, dst, tid);
}

// two tiles: all loads of both in flight before any LDS write
template <int RA, int RB, int DHC, int NT>
__device__ __forceinline__ void load_tiles2(float* __restrict__ da, const float* __restrict__ sa, int na, int lda,
                                            float* __restrict__ db, const float* __restrict__ sb, int nb, int ldb,
                                            int tid) {
  TileRegs<RA, DHC, NT> ta;
  TileRegs<RB, DHC, NT> tb;
  tile_fetch<RA, DHC, NT>(ta, sa, na, lda, tid);
  tile_fetch<RB, DHC, NT>(tb, sb, nb, ldb, tid);
  tile_store<RA, DHC, NT>(ta, da, tid);
  tile_store<RB, DHC, NT>(tb, db, tid);
}

#define MFMA4(ACC, AF, BF)            \
  ACC = mfma32(AF.x, BF.x, ACC);      \
  ACC = mfma32(AF.y, BF.y, ACC);      \
  ACC = mfma32(AF.z, BF.z, ACC);      \
  ACC = mfma32(AF.w, BF.w, ACC);

// ------------------------------------------------------------------------------------------
// forward: grid (ceil(Sq / (32 NW)), H, B), block 64 NW.  NKC = key chunks of 32 (all keys).
// ------------------------------------------------------------------------------------------
#ifndef MMNAS_DBG_FWD
#define MMNAS_DBG_FWD 0   // timing experiments only (wrong results): bit mask of forward phases left out (tools/mha_fwd_phases.sh)
#endif
template <int DHC, int NKC, int NW>
__device__ __forceinline__ void mha_fwd_body(const MhaK& p, const int qblk) {
  constexpr int LD = DHC + 4, NT = 64 * NW, JC = DHC >= 32 ? DHC / 32 : 1;
  __shared__ __attribute__((aligned(16))) float Qs[32 * NW * LD];
  __shared__ __attribute__((aligned(16))) float KVs[32 * NKC * LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, q0 = qblk * 32 * NW;
  const int SqS = p.Sq, SkS = p.Sk;   // strides of the per-(batch, head) arrays; the lengths of this batch element:
  int Sq = p.Sq, Sk = p.Sk;
  size_t qrow0 = (size_t)b * p.Sq, krow0 = (size_t)b * p.Sk;
  if (p.qoff) { const int o = p.qoff[b]; Sq = p.qoff[b + 1] - o; qrow0 = (size_t)o; }
  if (p.koff) { const int o = p.koff[b]; Sk = p.koff[b + 1] - o; krow0 = (size_t)o; }
  if (q0 >= Sq || Sk <= 0) return;   // (packed rows: this query block lies behind the sequence's end; workgroup-uniform)
  const bool active = q0 + 32 * w < Sq;  // wave-uniform

  // Key mask and key range as BIT MASKS in scalar registers (one ballot per 64 keys, every wave its own copy), pre-shifted
  // per lane half so that element (kc, r) tests a compile-time bit: bit 32 (kc & 1) + (r & 3) + 8 (r >> 2) of word kc >> 1.
  // (Round 5 kept the mask as floats in LDS: one dependent ds_read per score element inside the softmax -- the forward phase
  //  timing of tools/mha_fwd_phases.sh charged the softmax 7.7 us of the kernel's 21.)
  constexpr int NMW = (NKC + 1) / 2;
  unsigned long long mbits[NMW], vbits[NMW];
#pragma unroll
  for (int j = 0; j < NMW; ++j) {
    const int key = 64 * j + lane;
    const bool inr = key < Sk;
    const bool mk = p.mask && inr && p.mask[(size_t)b * SkS + (inr ? key : 0)];
    mbits[j] = __ballot(mk) >> (4 * hh);
    vbits[j] = __ballot(inr) >> (4 * hh);
  }

  f32x16 acc[NKC];
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kc][r] = 0.f;

  // One head-dim chunk (d_h <= 64: every attention of the workloads) and a V tile of <= 8 16-byte loads per thread: the V
  // loads are issued HERE, behind the Q / K loads and in front of everything that waits -- round 5 issued them behind the
  // softmax, so every workgroup (one per CU, one wave per SIMD: nothing else to run) sat out a second full memory round trip
  // between its two MFMA phases.  The relation bias (NKC <= 4: 64 values per lane) is fetched in the same burst.
  constexpr bool VPF = (32 * NKC * (DHC / 4) + NT - 1) / NT <= 8;
  constexpr bool BPF = false;   // (bias prefetch: 64 more live registers spill the 2-workgroups-per-CU build; measured 30 us instead of 21)
  const bool vpf = VPF && p.nch == 1;
  const int qi = q0 + 32 * w + l31;
  const bool qok = qi < Sq;
  const size_t bh = (size_t)b * p.H + h;
  const bool has_bias = p.biasT != nullptr;  // wave-uniform: hoisted so the bias loads issue as a batch
  const int qic = qok ? qi : Sq - 1;
  TileRegs<VPF ? 32 * NKC : 1, DHC, NT> tv;
  float biasp[BPF ? NKC : 1][16];

  // ---- S^T = K Q^T over the head-dim chunks ----
  for (int c = 0; c < p.nch; ++c) {
    if (c) __syncthreads();
    {
      TileRegs<32 * NW, DHC, NT> ta;
      TileRegs<32 * NKC, DHC, NT> tb;
      tile_fetch<32 * NW, DHC, NT>(ta, p.Q + (qrow0 + q0) * p.ldq + h * p.dh + c * DHC, Sq - q0, p.ldq, tid);
      tile_fetch<32 * NKC, DHC, NT>(tb, p.K + krow0 * p.ldk + h * p.dh + c * DHC, Sk, p.ldk, tid);
      if (c == 0) {
        if (VPF && vpf) tile_fetch<VPF ? 32 * NKC : 1, DHC, NT>(tv, p.V + krow0 * p.ldv + h * p.dh, Sk, p.ldv, tid);
        if (BPF && has_bias && active) {
#pragma unroll
          for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = min(32 * kc + acc_row(r, hh), Sk - 1);
              biasp[BPF ? kc : 0][r] = p.biasT[(bh * SkS + key) * SqS + qic];
            }
        }
      }
      tile_store<32 * NW, DHC, NT>(ta, Qs, tid);
      tile_store<32 * NKC, DHC, NT>(tb, KVs, tid);
    }
    __syncthreads();
    if (active) {
#if MMNAS_DBG_FWD & 1
      for (int kc = 0; kc < NKC; ++kc) acc[kc][0] = Qs[(32 * w + l31) * LD + kc] + KVs[l31 * LD + kc];
#else
#pragma unroll
      for (int s = 0; s < DHC / 8; ++s) {
        const float4 qf = *reinterpret_cast<const float4*>(Qs + (32 * w + l31) * LD + 8 * s + 4 * hh);
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
          const float4 kf = *reinterpret_cast<const float4*>(KVs + (32 * kc + l31) * LD + 8 * s + 4 * hh);
          MFMA4(acc[kc], kf, qf)
        }
      }
#endif
    }
  }

  // ---- softmax over the keys of this lane's query (registers + the other half-wave) ----
  float m = -INFINITY;
#if MMNAS_DBG_FWD & 2
  float sum = 1.f, inv = 1.f;
  if (qok && hh == 0) { p.stats[(bh * SqS + qi) * 2] = acc[0][0]; p.stats[(bh * SqS + qi) * 2 + 1] = inv; }
#else
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) {
    float bias[16];
    if (has_bias) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (BPF) { bias[r] = biasp[BPF ? kc : 0][r]; continue; }
        const int key = min(32 * kc + acc_row(r, hh), Sk - 1);
        bias[r] = p.biasT[(bh * SkS + key) * SqS + qic];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      constexpr unsigned long long one = 1ull;
      const unsigned long long bit = one << (32 * (kc & 1) + (r & 3) + 8 * (r >> 2));
      float v = acc[kc][r] * p.scale;
      if (has_bias) v += bias[r];
      v = (mbits[kc >> 1] & bit) ? -1e9f : v;
      v = (vbits[kc >> 1] & bit) ? v : -INFINITY;
      acc[kc][r] = v;
      m = fmaxf(m, v);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(acc[kc][r] - m);   // (v_exp_f32 on x*log2e: ~1e-6 relative; the backward kernels use the same form)
      acc[kc][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  if (qok && hh == 0) {
    p.stats[(bh * SqS + qi) * 2] = m;
    p.stats[(bh * SqS + qi) * 2 + 1] = inv;
  }
  // (dropout index (bh Sq + q) Sk + key, key = 32 kc + (r & 3) + 8 (r >> 2) + 4 hh: pre-multiplied per lane, a constant per element)
  const uint32_t dpre = drop_pre(p.drop, (uint32_t)((bh * SqS + qi) * SkS + 4 * hh));
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a = acc[kc][r] * inv;
      if (p.drop.thresh) a *= drop_mult_pre(p.drop, dpre + (uint32_t)(32 * kc + (r & 3) + 8 * (r >> 2)) * DROP_G);
      acc[kc][r] = a;
    }
#endif

  // ---- O = A V, head-dim chunk by chunk (A tile = the accumulators, as MFMA A operand) ----
  for (int c = 0; c < p.nch; ++c) {
    __syncthreads();
    if (VPF && vpf) tile_store<VPF ? 32 * NKC : 1, DHC, NT>(tv, KVs, tid);
    else load_tile<32 * NKC, DHC, NT>(KVs, p.V + krow0 * p.ldv + h * p.dh + c * DHC, Sk, p.ldv, tid);
    __syncthreads();
    if (!active) continue;
    f32x16 o[JC];
#pragma unroll
    for (int jc = 0; jc < JC; ++jc)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[jc][r] = 0.f;
#if MMNAS_DBG_FWD & 4
    for (int kc = 0; kc < NKC; ++kc) o[kc & (JC - 1)][kc] += acc[kc][3] + KVs[(32 * kc + l31) * LD + hh];
#else
    // (the B operands of a key chunk are read in ONE batch in front of its MFMAs: read -> wait -> MFMA pairs left every
    //  product waiting out an LDS round trip -- the P V phase took 7.2 us against 3.4 us of MFMA time)
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      float bvv[16][JC];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = 32 * kc + acc_row(r, hh);
#pragma unroll
        for (int jc = 0; jc < JC; ++jc) bvv[r][jc] = (DHC >= 32 || l31 < DHC) ? KVs[key * LD + 32 * jc + l31] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int jc = 0; jc < JC; ++jc) o[jc] = mfma32(acc[kc][r], bvv[r][jc], o[jc]);
    }
#endif
#if MMNAS_DBG_FWD & 8
    if (o[0][0] + o[JC - 1][5] == 12345.678f) p.O[(qrow0 + q0) * p.ldo] = o[0][1];
#else
#pragma unroll
    for (int jc = 0; jc < JC; ++jc) {
      const int col = 32 * jc + l31;
      if (col < DHC) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = q0 + 32 * w + acc_row(r, hh);
          if (q < Sq) p.O[(qrow0 + q) * p.ldo + h * p.dh + c * DHC + col] = o[jc][r];
        }
      }
    }
#endif
  }
}

template <int DHC, int NKC, int NW>
__global__ void __launch_bounds__(64 * NW, (NKC <= 4 && NW == 4) ? 2 : 1) mha_fwd_kernel(const MhaK p) {
  mha_fwd_body<DHC, NKC, NW>(p, blockIdx.x);
}

// Two attention cores of one geometry in ONE launch (the self / relation-self candidates of a supernet decoder node in the
// architecture step): a core of B*H = 256 workgroups is a single latency-bound round at one workgroup per CU although its
// LDS image admits two -- the second problem's workgroups ride beside the first's.  blockIdx.x = problem * nqb + query block.
template <int DHC, int NKC, int NW>
__global__ void __launch_bounds__(64 * NW, (NKC <= 4 && NW == 4) ? 2 : 1) mha_fwd_pair_kernel(const MhaK p0, const MhaK p1, const int nqb) {
  const int second = (int)blockIdx.x >= nqb;
  if (second) mha_fwd_body<DHC, NKC, NW>(p1, (int)blockIdx.x - nqb);
  else mha_fwd_body<DHC, NKC, NW>(p0, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// delta[b,h,q] = sum_j dO[b,q,h,j] * O[b,q,h,j]  ( = sum_k dA[q,k] A[q,k])
// ------------------------------------------------------------------------------------------
__global__ void mha_delta_kernel(const float* __restrict__ dO, const float* __restrict__ O, float* __restrict__ delta,
                                 int B, int H, int Sq, int dh, int ldo) {
  const long n = (long)B * Sq * H;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int h = (int)(i % H);
    const long row = i / H;  // b*Sq + q
    const float4* a = reinterpret_cast<const float4*>(dO + row * ldo + h * dh);
    const float4* o = reinterpret_cast<const float4*>(O + row * ldo + h * dh);
    float s = 0.f;
    for (int j = 0; j < dh / 4; ++j) {
      const float4 x = a[j], y = o[j];
      s += (x.x * y.x + x.y * y.y) + (x.z * y.z + x.w * y.w);
    }
    const int b = (int)(row / Sq), q = (int)(row - (long)b * Sq);
    delta[((size_t)b * H + h) * Sq + q] = s;
  }
}

// ------------------------------------------------------------------------------------------
// backward, query-owner: dQ (head-dim chunk oc) and dZ -> dbiasT.
// grid (ceil(Sq / (32 NW)), H * nch, B); keys swept in blocks of 32 NKC.
// ------------------------------------------------------------------------------------------
template <int DHC, int NKC, int NW>
__global__ void __launch_bounds__(64 * NW, (NKC <= 2 && NW == 4) ? 2 : 1) mha_bwd_q_kernel(const MhaK p) {
  constexpr int LD = DHC + 4, NT = 64 * NW, JC = DHC >= 32 ? DHC / 32 : 1, KB = 32 * NKC, NS = DHC / 8;
  // Only the key / value blocks go through LDS (shared by the waves).  The Q and dO operands of a wave are the
  // rows of its own 32 queries: each lane reads its fragments (4 consecutive floats of its row per k-step)
  // straight from global into registers, once for all key blocks.  Staging them through LDS as well cost
  // 70 KB per workgroup and, with the registers of the staging copies, held the kernel at one wave per SIMD.
  __shared__ __attribute__((aligned(16))) float Ks[KB * LD];
  __shared__ __attribute__((aligned(16))) float Vs[KB * LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y % p.H, oc = blockIdx.y / p.H, q0 = blockIdx.x * 32 * NW;
  const int Sq = p.Sq, Sk = p.Sk;
  const bool active = q0 + 32 * w < Sq;
  const int qi = q0 + 32 * w + l31;
  const bool qok = qi < Sq;
  const size_t bh = (size_t)b * p.H + h;
  float m = 0.f, inv = 0.f, del = 0.f;
  if (qok) {
    m = p.stats[(bh * Sq + qi) * 2];
    inv = p.stats[(bh * Sq + qi) * 2 + 1];
    if (p.nch > 1) del = p.delta[bh * Sq + qi];   // (one head-dim chunk: computed below from the fragments)
  }
  float4 qfr[NS], gfr[NS];   // B operands of S^T = K Q^T and dA^T = V dO^T for head-dim chunk c
  auto load_frags = [&](int c) {
    const int co = h * p.dh + c * DHC + 4 * hh;
    const size_t row = (size_t)b * Sq + (qok ? qi : Sq - 1);   // clamped: no branch around the loads
    const float* qrow = p.Q + row * p.ldq + co;
    const float* grow = p.dO + row * p.ldo + co;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      qfr[s] = *reinterpret_cast<const float4*>(qrow + 8 * s);
      gfr[s] = *reinterpret_cast<const float4*>(grow + 8 * s);
    }
    if (!qok) {
#pragma unroll
      for (int s = 0; s < NS; ++s) { qfr[s] = make_float4(0.f, 0.f, 0.f, 0.f); gfr[s] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
  };
  if (p.nch == 1) {
    // delta[b,h,q] = sum_j dO[q,j] O[q,j] (= sum_k dA[q,k] A[q,k]) from this lane's half row, the other half-wave
    // holds the other half; written for the dK/dV kernel that follows on the stream (replaces a separate launch)
    load_frags(0);
    const float* orow = p.O + ((size_t)b * Sq + (qok ? qi : Sq - 1)) * p.ldo + h * p.dh + 4 * hh;
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float4 o4 = *reinterpret_cast<const float4*>(orow + 8 * s);
      part += (gfr[s].x * o4.x + gfr[s].y * o4.y) + (gfr[s].z * o4.z + gfr[s].w * o4.w);
    }
    part += __shfl_xor(part, 32, 64);
    del = qok ? part : 0.f;
    if (qok && hh == 0) p.delta[bh * Sq + qi] = del;
  }
  f32x16 dq[JC];
#pragma unroll
  for (int jc = 0; jc < JC; ++jc)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[jc][r] = 0.f;

  for (int kb = 0; kb < Sk; kb += KB) {
    f32x16 acc[NKC], dacc[NKC];
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[kc][r] = 0.f; dacc[kc][r] = 0.f; }
    for (int c = 0; c < p.nch; ++c) {
      __syncthreads();
      const int co = h * p.dh + c * DHC;
      if (p.nch > 1) load_frags(c);
      load_tiles2<KB, KB, DHC, NT>(Ks, p.K + (size_t)(b * Sk + kb) * p.ldk + co, Sk - kb, p.ldk,
                                   Vs, p.V + (size_t)(b * Sk + kb) * p.ldv + co, Sk - kb, p.ldv, tid);
      __syncthreads();
      if (active) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int fo = 8 * s + 4 * hh;
          const float4 qf = qfr[s];
          const float4 gf = gfr[s];
#pragma unroll
          for (int kc = 0; kc < NKC; ++kc) {
            const float4 kf = *reinterpret_cast<const float4*>(Ks + (32 * kc + l31) * LD + fo);
            const float4 vf = *reinterpret_cast<const float4*>(Vs + (32 * kc + l31) * LD + fo);
            MFMA4(acc[kc], kf, qf)
            MFMA4(dacc[kc], vf, gf)
          }
        }
      }
    }
    // dZ^T for this key block (in place of acc)
    const bool has_bias = p.biasT != nullptr, has_mask = p.mask != nullptr, has_drop = p.drop.thresh != 0;
    const bool put_dbias = p.dbiasT != nullptr && oc == 0;
    const int qic = qok ? qi : Sq - 1;
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      float bias[16], mk[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {  // batched, branch-free operand fetch (clamped indices)
        const int key = min(kb + 32 * kc + acc_row(r, hh), Sk - 1);
        bias[r] = has_bias ? p.biasT[(bh * Sk + key) * Sq + qic] : 0.f;
        mk[r] = has_mask ? (float)p.mask[(size_t)b * Sk + key] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb + 32 * kc + acc_row(r, hh);
        const bool ok = key < Sk && qok;
        const bool masked = mk[r] != 0.f;
        float v = acc[kc][r] * p.scale + bias[r];
        if (masked) v = -1e9f;
        const float pr = __expf(v - m) * inv;
        const float dm = has_drop ? drop_mult(p.drop, (uint32_t)((bh * Sq + qi) * Sk + key)) : 1.f;
        const float dz = (ok && !masked) ? pr * (dacc[kc][r] * dm - del) : 0.f;
        if (put_dbias && ok) p.dbiasT[(bh * Sk + key) * Sq + qi] = dz;
        acc[kc][r] = dz * p.scale;
      }
    }
    if (p.nch > 1) {
      __syncthreads();
      load_tile<KB, DHC, NT>(Ks, p.K + (size_t)(b * Sk + kb) * p.ldk + h * p.dh + oc * DHC, Sk - kb, p.ldk, tid);
      __syncthreads();
    }
    if (active) {
#pragma unroll
      for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kl = 32 * kc + acc_row(r, hh);
#pragma unroll
          for (int jc = 0; jc < JC; ++jc) {
            const float bv = (DHC >= 32 || l31 < DHC) ? Ks[kl * LD + 32 * jc + l31] : 0.f;
            dq[jc] = mfma32(acc[kc][r], bv, dq[jc]);
          }
        }
    }
  }
  if (active) {
#pragma unroll
    for (int jc = 0; jc < JC; ++jc) {
      const int col = 32 * jc + l31;
      if (col < DHC) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = q0 + 32 * w + acc_row(r, hh);
          if (q < Sq) p.dQ[(size_t)(b * Sq + q) * p.ldq + h * p.dh + oc * DHC + col] = dq[jc][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, key-owner: dK, dV (head-dim chunk oc).  grid (ceil(Sk / (32 NW)), H * nch, B);
// queries swept 32 at a time.
// ------------------------------------------------------------------------------------------
// QS > 1 (few keys, many queries -- guided attention over 14 words): QS groups of NW waves own the SAME keys and
// split the query blocks among them (the single-wave form walked the 4 query blocks one after the other,
// 41 us of pure latency); their dK / dV partials are added in group order through LDS at the end.
template <int DHC, int NW, int QS>
__global__ void __launch_bounds__(64 * NW * QS, (NW == 4 && QS == 1) ? 2 : 1) mha_bwd_kv_kernel(const MhaK p) {
  constexpr int LD = DHC + 4, NT = 64 * NW, JC = DHC >= 32 ? DHC / 32 : 1, NS = DHC / 8;
  // Q / dO blocks (shared by the waves) go through LDS; a wave's own K and V rows are read straight from global
  // into register fragments (see mha_bwd_q_kernel).
  constexpr int PER = NW * JC * 2 * 16 * 64;   // floats of one group's dK / dV partials (QS > 1)
  constexpr int TILES = 2 * QS * 32 * LD, SMEM = TILES > (QS - 1) * PER ? TILES : (QS - 1) * PER;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* QsAll = smem;
  float* GsAll = smem + QS * 32 * LD;
  __shared__ float sMAll[QS * 32], sInvAll[QS * 32], sDelAll[QS * 32];
  const int qs = threadIdx.x / NT;            // query-split group of this wave
  const int tid = threadIdx.x - qs * NT;      // thread index inside the group
  float* Qs = QsAll + qs * 32 * LD;
  float* Gs = GsAll + qs * 32 * LD;
  float* sM = sMAll + qs * 32;
  float* sInv = sInvAll + qs * 32;
  float* sDel = sDelAll + qs * 32;
  const int lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y % p.H, oc = blockIdx.y / p.H, k0 = blockIdx.x * 32 * NW;
  const int Sq = p.Sq, Sk = p.Sk;
  const bool active = k0 + 32 * w < Sk;
  const int key = k0 + 32 * w + l31;
  const bool kok = key < Sk;
  const bool masked = kok && p.mask && p.mask[(size_t)b * Sk + key];
  const size_t bh = (size_t)b * p.H + h;
  f32x16 dk[JC], dv[JC];
#pragma unroll
  for (int jc = 0; jc < JC; ++jc)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[jc][r] = 0.f; dv[jc][r] = 0.f; }
  float4 kfr[NS], vfr[NS];
  auto load_frags = [&](int c) {
    const int co = h * p.dh + c * DHC + 4 * hh;
    const size_t row = (size_t)b * Sk + (kok ? key : Sk - 1);   // clamped: no branch around the loads
    const float* krow = p.K + row * p.ldk + co;
    const float* vrow = p.V + row * p.ldv + co;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      kfr[s] = *reinterpret_cast<const float4*>(krow + 8 * s);
      vfr[s] = *reinterpret_cast<const float4*>(vrow + 8 * s);
    }
    if (!kok) {
#pragma unroll
      for (int s = 0; s < NS; ++s) { kfr[s] = make_float4(0.f, 0.f, 0.f, 0.f); vfr[s] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
  };

  // every group runs the same number of rounds (the barriers are workgroup-wide); a group whose block lies beyond
  // Sq works on zero tiles
  for (int qc0 = 0; qc0 < Sq; qc0 += 32 * QS) {
    const int qc = qc0 + 32 * qs;
    f32x16 acc, dacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; dacc[r] = 0.f; }
    for (int c = 0; c < p.nch; ++c) {
      __syncthreads();
      const int co = h * p.dh + c * DHC;
      const int qcl = qc < Sq ? qc : 0;   // (a block beyond Sq: valid addresses, zero rows)
      load_tiles2<32, 32, DHC, NT>(Qs, p.Q + (size_t)(b * Sq + qcl) * p.ldq + co, Sq - qc, p.ldq,
                                   Gs, p.dO + (size_t)(b * Sq + qcl) * p.ldo + co, Sq - qc, p.ldo, tid);
      if (p.nch > 1 || qc0 == 0) load_frags(c);
      if (c == 0 && tid < 32) {
        const int q = qc + tid;
        const bool ok = q < Sq;
        sM[tid] = ok ? p.stats[(bh * Sq + q) * 2] : 0.f;
        sInv[tid] = ok ? p.stats[(bh * Sq + q) * 2 + 1] : 0.f;
        sDel[tid] = ok ? p.delta[bh * Sq + q] : 0.f;
      }
      __syncthreads();
      if (active) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int fo = 8 * s + 4 * hh;
          const float4 qf = *reinterpret_cast<const float4*>(Qs + l31 * LD + fo);
          const float4 gf = *reinterpret_cast<const float4*>(Gs + l31 * LD + fo);
          const float4 kf = kfr[s];
          const float4 vf = vfr[s];
          MFMA4(acc, qf, kf)    // S[query][key]
          MFMA4(dacc, gf, vf)   // dA[query][key]
        }
      }
    }
    // A and dZ for (query = register row, key = lane)
    {
      const bool has_bias = p.biasT != nullptr, has_drop = p.drop.thresh != 0;
      const int keyc = kok ? key : Sk - 1;
      float bias[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {  // batched, branch-free bias fetch (clamped indices)
        const int q = min(qc + acc_row(r, hh), Sq - 1);
        bias[r] = has_bias ? p.biasT[(bh * Sk + keyc) * Sq + q] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = acc_row(r, hh), q = qc + ql;
        const bool ok = q < Sq && kok;
        float v = acc[r] * p.scale + bias[r];
        if (masked) v = -1e9f;
        const float pr = __expf(v - sM[ql]) * sInv[ql];
        const float dm = has_drop ? drop_mult(p.drop, (uint32_t)((bh * Sq + q) * Sk + key)) : 1.f;
        acc[r] = ok ? pr * dm : 0.f;
        dacc[r] = (ok && !masked) ? pr * (dacc[r] * dm - sDel[ql]) * p.scale : 0.f;
      }
    }
    if (p.nch > 1) {
      __syncthreads();
      const int co = h * p.dh + oc * DHC;
      const int qcl = qc < Sq ? qc : 0;
      load_tiles2<32, 32, DHC, NT>(Qs, p.Q + (size_t)(b * Sq + qcl) * p.ldq + co, Sq - qc, p.ldq,
                                   Gs, p.dO + (size_t)(b * Sq + qcl) * p.ldo + co, Sq - qc, p.ldo, tid);
      __syncthreads();
    }
    if (active) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = acc_row(r, hh);
#pragma unroll
        for (int jc = 0; jc < JC; ++jc) {
          const bool cok = (DHC >= 32 || l31 < DHC);
          const float gv = cok ? Gs[ql * LD + 32 * jc + l31] : 0.f;
          const float qv = cok ? Qs[ql * LD + 32 * jc + l31] : 0.f;
          dv[jc] = mfma32(acc[r], gv, dv[jc]);    // dV[key][j] += A[q][key] dO[q][j]
          dk[jc] = mfma32(dacc[r], qv, dk[jc]);   // dK[key][j] += dZ[q][key] Q[q][j] / sqrt(dh)
        }
      }
    }
  }
  if (QS > 1) {   // groups 1.. leave their partials in LDS (the tile images are free now); group 0 adds them in order
    __syncthreads();
    float* red = smem;
    float* base = red + (qs > 0 ? (qs - 1) * PER : 0) + w * (JC * 2 * 16 * 64);
    if (qs > 0) {
#pragma unroll
      for (int jc = 0; jc < JC; ++jc)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          base[((jc * 2 + 0) * 16 + r) * 64 + lane] = dk[jc][r];
          base[((jc * 2 + 1) * 16 + r) * 64 + lane] = dv[jc][r];
        }
    }
    __syncthreads();
    if (qs > 0) return;
    for (int g = 1; g < QS; ++g) {
      const float* src = red + (g - 1) * PER + w * (JC * 2 * 16 * 64);
#pragma unroll
      for (int jc = 0; jc < JC; ++jc)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dk[jc][r] += src[((jc * 2 + 0) * 16 + r) * 64 + lane];
          dv[jc][r] += src[((jc * 2 + 1) * 16 + r) * 64 + lane];
        }
    }
  }
  if (active) {
#pragma unroll
    for (int jc = 0; jc < JC; ++jc) {
      const int col = 32 * jc + l31;
      if (col < DHC) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kr = k0 + 32 * w + acc_row(r, hh);
          if (kr < Sk) {
            p.dK[(size_t)(b * Sk + kr) * p.ldk + h * p.dh + oc * DHC + col] = dk[jc][r];
            p.dV[(size_t)(b * Sk + kr) * p.ldv + h * p.dh + oc * DHC + col] = dv[jc][r];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, FUSED: dQ, dK, dV (and dZ -> dbiasT) of one (batch, head) pair in ONE workgroup, five products instead
// of the seven of the query-owner + key-owner pair above.  For dh = 64, Sq <= 128, Sk <= 128 -- every attention of the
// VQA / VGD / ITM workloads (100 regions, 14 / 50 tokens).  grid (H, B), 4 waves = the 4 query blocks of 32.
//   * K and V of the head sit in LDS once; a wave keeps the Q / dO rows of its query block in registers, both as the
//     B fragments of S^T = K Q^T / dA^T = V dO^T and as the B operands of the dK / dV products.
//   * per (key block, query block) tile the wave computes S^T and dA^T, the softmax backward in registers
//     (key = register row, query = lane), dQ += dZ K with the tile as MFMA A operand -- as the query-owner kernel does.
//   * dK += dZ^T Q and dV += A^T dO need the tile with key on the lane: instead of recomputing S in that orientation
//     (two extra products) the two 32x32 tiles are TRANSPOSED through a wave-private 4 KB LDS image (write rows, read
//     columns; stride 33: conflict-free) and fed as A operands again.
//   * the dK / dV contributions of the four query blocks meet in an LDS accumulator [key][64]: at step s wave w works
//     on key block (w + s) mod NKB, so the waves of one step touch different key blocks; a barrier separates the steps.
//     With fewer key blocks than waves (guided attention: 14 keys) the waves sharing a block have their own copies,
//     summed at write-out.
// LDS: NKB = 4: K, V 68 KB + accumulators 64 KB + transposition 16.5 KB = 148.5 KB (one workgroup per CU).
// ------------------------------------------------------------------------------------------
#ifndef MMNAS_DBG_MHA
#define MMNAS_DBG_MHA 0   // timing experiments only (wrong results): bit mask of kernel phases left out (tools/mha_phases.sh)
#endif
#ifndef MMNAS_MHA_VPM
#define MMNAS_MHA_VPM 8    // vector instructions scheduled behind each MFMA of the overlapped phase (tuning)
#endif
template <int NKB, bool DB>   // DB: the bias gradient dZ is written (relation attention)
__global__ void __launch_bounds__(256, 1) mha_bwd_fused_kernel(const MhaK p) {
  constexpr int DHC = 64, LD = DHC + 4, NS = DHC / 8, JC = 2, NW = 4, KB = 32 * NKB;
  constexpr int CP = NW / NKB;   // accumulator copies (waves that meet in one key block at a step)
  __shared__ __attribute__((aligned(16))) float Ks[KB * LD];
  __shared__ __attribute__((aligned(16))) float Vs[KB * LD];
  __shared__ __attribute__((aligned(16))) float dKs[CP * KB * DHC];
  __shared__ __attribute__((aligned(16))) float dVs[CP * KB * DHC];
  __shared__ float TrAll[NW][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, h = blockIdx.x;
  const int SqS = p.Sq, SkS = p.Sk;   // strides of the per-(batch, head) arrays; the lengths of this batch element:
  int Sq = p.Sq, Sk = p.Sk;
  size_t qrow0 = (size_t)b * p.Sq, krow0 = (size_t)b * p.Sk;
  if (p.qoff) { const int o = p.qoff[b]; Sq = p.qoff[b + 1] - o; qrow0 = (size_t)o; }
  if (p.koff) { const int o = p.koff[b]; Sk = p.koff[b + 1] - o; krow0 = (size_t)o; }
  if (Sq <= 0 || Sk <= 0) return;   // (an empty packed sequence: nothing to read, nothing to write)
  const int q0 = 32 * w;
  const bool active = q0 < Sq;   // wave-uniform
  const int qi = q0 + l31;
  const bool qok = qi < Sq;
  const size_t bh = (size_t)b * p.H + h;
  float* Tr = TrAll[w];

  load_tiles2<KB, KB, DHC, 256>(Ks, p.K + krow0 * p.ldk + h * p.dh, Sk, p.ldk,
                                Vs, p.V + krow0 * p.ldv + h * p.dh, Sk, p.ldv, tid);
  for (int i = tid; i < CP * KB * DHC / 4; i += 256) {
    reinterpret_cast<float4*>(dKs)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(dVs)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float m = 0.f, inv = 0.f, del = 0.f;
  float4 qfr[NS], gfr[NS];       // B fragments of S^T = K Q^T and dA^T = V dO^T: this lane's query row
  float qB[JC][16], gB[JC][16];  // B operands of dK += dZ^T Q, dV += A^T dO: B[k = query acc_row(r, hh)][j = 32 jc + l31]
  if (active) {
    if (qok) {
      m = p.stats[(bh * SqS + qi) * 2];
      inv = p.stats[(bh * SqS + qi) * 2 + 1];
    }
    const size_t row = qrow0 + (qok ? qi : Sq - 1);   // clamped: no branch around the loads
    const float* qrow = p.Q + row * p.ldq + h * p.dh + 4 * hh;
    const float* grow = p.dO + row * p.ldo + h * p.dh + 4 * hh;
    const float* orow = p.O + row * p.ldo + h * p.dh + 4 * hh;
    float part = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      qfr[s] = *reinterpret_cast<const float4*>(qrow + 8 * s);
      gfr[s] = *reinterpret_cast<const float4*>(grow + 8 * s);
      const float4 o4 = *reinterpret_cast<const float4*>(orow + 8 * s);
      part += (gfr[s].x * o4.x + gfr[s].y * o4.y) + (gfr[s].z * o4.z + gfr[s].w * o4.w);
    }
    part += __shfl_xor(part, 32, 64);   // delta[q] = sum_j dO[q,j] O[q,j]: the other half-wave holds the other half row
    del = qok ? part : 0.f;
    if (!qok) {
#pragma unroll
      for (int s = 0; s < NS; ++s) { qfr[s] = make_float4(0.f, 0.f, 0.f, 0.f); gfr[s] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = q0 + acc_row(r, hh);
      const bool ok = q < Sq;
      const size_t rr = qrow0 + (ok ? q : Sq - 1);
#pragma unroll
      for (int jc = 0; jc < JC; ++jc) {
#if MMNAS_DBG_MHA & 32
        const float qv = (float)r, gv = (float)jc;
#else
        const float qv = p.Q[rr * p.ldq + h * p.dh + 32 * jc + l31];
        const float gv = p.dO[rr * p.ldo + h * p.dh + 32 * jc + l31];
#endif
        qB[jc][r] = ok ? qv : 0.f;
        gB[jc][r] = ok ? gv : 0.f;
      }
    }
  }
  f32x16 dq[JC];
#pragma unroll
  for (int jc = 0; jc < JC; ++jc)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[jc][r] = 0.f;
  __syncthreads();

  // Bias / mask operands and the bias-gradient stores go through buffer resources whose range check does the
  // predication: an absent operand is a zero-length buffer (loads return 0), a lane outside the problem an out-of-range
  // offset (loads return 0, stores are dropped).  No branch anywhere in a tile: its whole body is ONE scheduling region,
  // which lets the S^T / dA^T products of the NEXT tile be interleaved with the softmax-backward arithmetic of this one.
  const size_t bho = bh * (size_t)SkS * SqS;
  const unsigned plane = (unsigned)SkS * (unsigned)SqS * 4u;
  const __amdgpu_buffer_rsrc_t bias_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.biasT ? p.biasT + bho : p.Q), 0, p.biasT ? plane : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t mask_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.mask ? p.mask + (size_t)b * SkS : (const uint8_t*)p.Q), 0, p.mask ? (unsigned)Sk : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t dbias_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(DB ? p.dbiasT + bho : p.dQ), 0, DB ? plane : 0u, 0x00020000);
  const uint32_t drop_pre0 = drop_pre(p.drop, (uint32_t)((bh * SqS + qi) * SkS + 4 * hh));   // pre-multiplied dropout index of key 4 hh

  // phase 1 of a tile: S^T = K Q^T and dA^T = V dO^T (64 MFMAs) + the tile's bias / mask operands
  auto P1 = [&](int k0, f32x16& acc, f32x16& dacc, float (&bias)[16], float (&mk)[16]) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + acc_row(r, hh);
      const bool okk = key < Sk && qok;
      bias[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bias_rs, okk ? (unsigned)(key * SqS + qi) * 4u : ~0u, 0, 0));
      mk[r] = (float)__builtin_amdgcn_raw_buffer_load_b8(mask_rs, key < Sk ? (unsigned)key : ~0u, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; dacc[r] = 0.f; }
#if !(MMNAS_DBG_MHA & 1)
#pragma unroll
    for (int s8 = 0; s8 < NS; ++s8) {
      const int fo = 8 * s8 + 4 * hh;
      // (f32x4, not float4: the HIP struct is taken apart into four scalar loads before instruction selection, and inside the
      //  software-pipelined loop they came back as ds_read2_b32 pairs -- 32-bank accesses at a row stride of 68 words, 4-way
      //  conflicts: the 0.285 SQ_LDS_BANK_CONFLICT fraction of rounds 3-5.  A first-class vector load stays one ds_read_b128.)
      const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (k0 + l31) * LD + fo);
      const f32x4 vf = *reinterpret_cast<const f32x4*>(Vs + (k0 + l31) * LD + fo);
      MFMA4(acc, kf, qfr[s8])
      MFMA4(dacc, vf, gfr[s8])
    }
#else
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = qfr[r & 7].x + bias[r]; dacc[r] = gfr[r & 7].y + mk[r]; }
#endif
  };
  // phase 2: softmax backward in registers; acc becomes dZ^T / sqrt(dh), dacc becomes A^T = (P o D)^T   [key][query]
  auto P2 = [&](int k0, f32x16& acc, f32x16& dacc, const float (&bias)[16], const float (&mk)[16]) __attribute__((always_inline)) {
#if !(MMNAS_DBG_MHA & 2)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + acc_row(r, hh);
      const bool ok = key < Sk && qok;
      const bool masked = mk[r] != 0.f;
      float v = acc[r] * p.scale + bias[r];
      v = masked ? -1e9f : v;
      const float pr = __expf(v - m) * inv;
      // (no dropout: thresh 0, scale 1 -> multiplier 1)
      const float dm = drop_mult_pre(p.drop, drop_pre0 + (uint32_t)k0 * DROP_G + (uint32_t)((r & 3) + 8 * (r >> 2)) * DROP_G);
      const float dz = (ok && !masked) ? pr * (dacc[r] * dm - del) : 0.f;
      if (DB) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dz), dbias_rs, ok ? (unsigned)(key * SqS + qi) * 4u : ~0u, 0, 0);
      acc[r] = dz * p.scale;
      dacc[r] = ok ? pr * dm : 0.f;
    }
#endif
  };
  // phases 3 + 4: dQ += dZ K; the two tiles transposed through the wave's LDS image; dK / dV contributions into the
  // LDS accumulators of this step's key block
  auto P34 = [&](int k0, int cp, f32x16& acc, f32x16& dacc) __attribute__((always_inline)) {
#if !(MMNAS_DBG_MHA & 4)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kl = k0 + acc_row(r, hh);
#pragma unroll
      for (int jc = 0; jc < JC; ++jc) dq[jc] = mfma32(acc[r], Ks[kl * LD + 32 * jc + l31], dq[jc]);
    }
#else
    dq[0][0] += acc[3]; dq[1][1] += dacc[5];
#endif
#if MMNAS_DBG_MHA & 8
    Tr[lane] = acc[0] + dacc[1];
#else
    // transpose the two tiles: [key][query] -> registers = query, lane = key (same-wave LDS accesses stay in order)
    f32x16 tz, tp;
#pragma unroll
    for (int r = 0; r < 16; ++r) Tr[acc_row(r, hh) * 33 + l31] = acc[r];
#pragma unroll
    for (int r = 0; r < 16; ++r) tz[r] = Tr[l31 * 33 + acc_row(r, hh)];
#pragma unroll
    for (int r = 0; r < 16; ++r) Tr[acc_row(r, hh) * 33 + l31] = dacc[r];
#pragma unroll
    for (int r = 0; r < 16; ++r) tp[r] = Tr[l31 * 33 + acc_row(r, hh)];
    // dK[key][j] += dZ[q][key] Q[q][j] / sqrt(dh),  dV[key][j] += A[q][key] dO[q][j]
    f32x16 dkp[JC], dvp[JC];
#pragma unroll
    for (int jc = 0; jc < JC; ++jc)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dkp[jc][r] = 0.f; dvp[jc][r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int jc = 0; jc < JC; ++jc) {
        dkp[jc] = mfma32(tz[r], qB[jc][r], dkp[jc]);
        dvp[jc] = mfma32(tp[r], gB[jc][r], dvp[jc]);
      }
    float* dka = dKs + (size_t)(cp * KB + k0) * DHC;
    float* dva = dVs + (size_t)(cp * KB + k0) * DHC;
#if MMNAS_DBG_MHA & 16
    dka[lane] = dkp[0][0] + dkp[1][3]; dva[lane] = dvp[0][1] + dvp[1][2];
#else
#pragma unroll
    for (int jc = 0; jc < JC; ++jc)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = acc_row(r, hh) * DHC + 32 * jc + l31;
        dka[o] += dkp[jc][r];
        dva[o] += dvp[jc][r];
      }
#endif
#endif
  };

  // Software pipeline over the key blocks (fully unrolled: the two tile register sets alternate statically).  With one
  // wave per SIMD nothing else hides the ~480 vector instructions of a tile's softmax backward (index hash of the dropout
  // replay, exp, selects): issued between the 64 independent MFMAs of the next tile's first phase they cost nothing --
  // measured 15 us of this kernel's 51 (tools/mha_phases.sh) before.
  f32x16 tA[2], tD[2];
  float tb[2][16], tm[2][16];
  constexpr int NSTEP = (MMNAS_DBG_MHA & 64) ? 0 : NKB;
  if (active && NSTEP > 0) P1(32 * (w % NKB), tA[0], tD[0], tb[0], tm[0]);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    const int kbi = (w + s) % NKB, k0 = 32 * kbi, cp = w / NKB;
    if (active) {
      if (s + 1 < NSTEP) P1(32 * ((w + s + 1) % NKB), tA[nxt], tD[nxt], tb[nxt], tm[nxt]);
      P2(k0, tA[cur], tD[cur], tb[cur], tm[cur]);
      if (s + 1 < NSTEP) {
        // one MFMA of the next tile, then a slice of this tile's vector work, 64 times
#pragma unroll
        for (int i = 0; i < 64; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, MMNAS_MHA_VPM, 0);
        }
      }
      P34(k0, cp, tA[cur], tD[cur]);
    }
    __syncthreads();
  }
  if (active) {
#pragma unroll
    for (int jc = 0; jc < JC; ++jc) {
      const int col = 32 * jc + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = q0 + acc_row(r, hh);
        if (q < Sq) p.dQ[(qrow0 + q) * p.ldq + h * p.dh + col] = dq[jc][r];
      }
    }
  }
  for (int i = tid; i < Sk * (DHC / 4); i += 256) {
    const int key = i / (DHC / 4), c4 = i - key * (DHC / 4);
    float4 a = reinterpret_cast<const float4*>(dKs)[key * (DHC / 4) + c4];
    float4 v = reinterpret_cast<const float4*>(dVs)[key * (DHC / 4) + c4];
#pragma unroll
    for (int c = 1; c < CP; ++c) {
      const float4 a2 = reinterpret_cast<const float4*>(dKs)[(c * KB + key) * (DHC / 4) + c4];
      const float4 v2 = reinterpret_cast<const float4*>(dVs)[(c * KB + key) * (DHC / 4) + c4];
      a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
      v.x += v2.x; v.y += v2.y; v.z += v2.z; v.w += v2.w;
    }
    *reinterpret_cast<float4*>(p.dK + (krow0 + key) * p.ldk + h * p.dh + 4 * c4) = a;
    *reinterpret_cast<float4*>(p.dV + (krow0 + key) * p.ldv + h * p.dh + 4 * c4) = v;
  }
}

// (Measured and removed in round 2: the same kernel on EIGHT waves -- wave = (query block, key half), two tiles per wave,
//  two waves per SIMD sharing one K / V image, 16-column half transpositions to fit 152 KB of LDS.  With LDS float
//  atomics for the dK / dV accumulators 118 us against this kernel's 51 us (B = 64, H = 4, S = 100); with two-turn
//  barrier-separated read-modify-writes 60 us.  The second wave per SIMD buys less than its duplicate operand loads,
//  half-lane LDS stores and extra barriers cost.  tools/mha_bench.py times the attention kernels alone.)

// attention_bwd16.hip
bool mha_bwd_b16_launch(int B, int H, int Sq, int Sk, int ldq, int ldk, int ldv, int ldo, const float* Q, const float* K, const float* V,
                        const float* O, const float* dO, const uint8_t* mask, const float* biasT, const float* stats, float* dQ, float* dK,
                        float* dV, float* dbiasT, DropCfg drop, float scale, const int* qoff, const int* koff, hipStream_t st);

bool mha_fwd_b16_launch(int B, int H, int Sq, int Sk, int ldq, int ldk, int ldv, int ldo, const float* Q, const float* K, const float* V,
                        const uint8_t* mask, const float* biasT, float* O, float* stats, DropCfg drop, float scale, const int* qoff,
                        const int* koff, hipStream_t st);

bool mha_fwd_b16_pair(int B, int H, int Sq, int Sk, int ld, const float* const* Q, const float* const* K, const float* const* V,
                      const uint8_t* const* mask, const float* const* biasT, float* const* O, float* const* stats, const DropCfg* drop,
                      float scale, const int* qoff, const int* koff, hipStream_t st);

static int fill(const mmnas_mha_desc* d, MhaK& k, bool bwd) {
  MMNAS_REQUIRE(d, MMNAS_E_ARG, "mha: null descriptor");
  MMNAS_REQUIRE(d->B > 0 && d->H > 0 && d->Sq > 0 && d->Sk > 0, MMNAS_E_SHAPE, "mha: B=%d H=%d Sq=%d Sk=%d", d->B,
                d->H, d->Sq, d->Sk);
  MMNAS_REQUIRE(d->dh == 16 || d->dh == 32 || d->dh == 64 || d->dh == 128 || d->dh == 256, MMNAS_E_SHAPE,
                "mha: head dim %d not in {16,32,64,128,256}", d->dh);
  MMNAS_REQUIRE(d->Sk <= 256, MMNAS_E_SHAPE, "mha: Sk=%d > 256 keys not supported", d->Sk);
  MMNAS_REQUIRE(d->Q && d->K && d->V && d->lse, MMNAS_E_ARG, "mha: null Q/K/V/stats");
  MMNAS_REQUIRE(d->ldq % 4 == 0 && d->ldk % 4 == 0 && d->ldv % 4 == 0 && d->ldo % 4 == 0, MMNAS_E_SHAPE,
                "mha: row strides must be multiples of 4 floats");
  MMNAS_REQUIRE((((uintptr_t)d->Q | (uintptr_t)d->K | (uintptr_t)d->V) & 15) == 0, MMNAS_E_ARG,
                "mha: Q/K/V must be 16-byte aligned");
  MMNAS_REQUIRE((double)d->B * d->H * d->Sq * d->Sk < 4294967296.0, MMNAS_E_SHAPE, "mha: score tensor too large");
  k.B = d->B; k.H = d->H; k.Sq = d->Sq; k.Sk = d->Sk; k.dh = d->dh;
  k.ldq = d->ldq; k.ldk = d->ldk; k.ldv = d->ldv; k.ldo = d->ldo;
  k.Q = d->Q; k.K = d->K; k.V = d->V; k.mask = d->mask; k.biasT = d->biasT; k.O = d->O; k.stats = d->lse;
  k.drop = make_drop(d->drop_p, d->drop_seed, d->drop_site);
  k.scale = 1.0f / sqrtf((float)d->dh);
  k.dO = d->dO; k.dQ = d->dQ; k.dK = d->dK; k.dV = d->dV; k.dbiasT = d->dbiasT; k.delta = d->delta;
  k.qoff = d->q_off; k.koff = d->k_off;
  if (k.qoff || k.koff) {
    MMNAS_REQUIRE(d->dh == 64 && d->Sq <= 128 && d->Sk <= 128, MMNAS_E_SHAPE, "mha: packed rows (q_off / k_off) need d_h = 64 and at most 128 queries / keys per sequence (Sq=%d Sk=%d dh=%d)", d->Sq, d->Sk, d->dh);
    MMNAS_REQUIRE(!(k.koff && d->mask), MMNAS_E_ARG, "mha: packed keys carry no padding: no mask with k_off");
  }
  if (bwd) {
    MMNAS_REQUIRE(d->dO && d->dQ && d->dK && d->dV && d->delta && d->O, MMNAS_E_ARG, "mha_bwd: null gradient buffer");
    MMNAS_REQUIRE((((uintptr_t)d->dO | (uintptr_t)d->O) & 15) == 0, MMNAS_E_ARG, "mha_bwd: dO/O alignment");
  } else {
    MMNAS_REQUIRE(d->O, MMNAS_E_ARG, "mha_fwd: null output");
  }
  return MMNAS_OK;
}

static int mha_nw() {
  const char* e = getenv("MMNAS_MHA_NW");
  return (e && atoi(e) == 2) ? 2 : 4;   // measured: 4-wave groups are 7-15 % faster (K/V tiles loaded half as often)
}

template <int DHC>
static void launch_fwd(const MhaK& k, hipStream_t st) {
  const int nkc = cdiv(k.Sk, 32);
#define FWD(NKC, NW) MMNAS_LAUNCH((mha_fwd_kernel<DHC, NKC, NW>), dim3(cdiv(k.Sq, 32 * NW), k.H, k.B), \
                                        dim3(64 * NW), 0, st, k)
  // waves per workgroup: 2 (64 queries) keeps the LDS image small enough for 2-3 workgroups per CU,
  // which hides the tile loads of one behind the MFMAs of another (MMNAS_MHA_NW=4 restores 128-query groups)
  const int nwmax = mha_nw();
  if (k.Sq <= 32) {
    if (nkc <= 1) FWD(1, 1); else if (nkc <= 2) FWD(2, 1); else if (nkc <= 4) FWD(4, 1); else FWD(8, 1);
  } else if (k.Sq <= 64 || nwmax == 2) {
    if (nkc <= 1) FWD(1, 2); else if (nkc <= 2) FWD(2, 2); else if (nkc <= 4) FWD(4, 2); else FWD(8, 2);
  } else {
    if (nkc <= 1) FWD(1, 4); else if (nkc <= 2) FWD(2, 4); else if (nkc <= 4) FWD(4, 4); else FWD(8, 4);
  }
#undef FWD
}

template <int DHC>
static void launch_bwd(const MhaK& k, hipStream_t st) {
  const int nkc = cdiv(k.Sk, 32);
#define BQ(NKC, NW) MMNAS_LAUNCH((mha_bwd_q_kernel<DHC, NKC, NW>), dim3(cdiv(k.Sq, 32 * NW), k.H * k.nch, k.B), \
                                       dim3(64 * NW), 0, st, k)
  const int nwmax = mha_nw();
  if (k.Sq <= 32) { if (nkc <= 1) BQ(1, 1); else BQ(2, 1); }
  else if (k.Sq <= 64 || nwmax == 2) { if (nkc <= 1) BQ(1, 2); else BQ(2, 2); }
  else {
    // 65..128 keys: one 128-key block (Q/dO/K/V tiles loaded once, 139 KB LDS) instead of two 64-key
    // blocks that reload Q and dO; MMNAS_MHA_BQ4=0 restores the two-block form
    if (nkc <= 1) BQ(1, 4); else BQ(2, 4);   // 64-key blocks: 35 KB LDS, <= 256 VGPRs -> two workgroups per CU
  }
#undef BQ
#define BKV(NW, QS) MMNAS_LAUNCH((mha_bwd_kv_kernel<DHC, NW, QS>), dim3(cdiv(k.Sk, 32 * NW), k.H * k.nch, k.B), \
                                       dim3(64 * NW * QS), 0, st, k)
  if (nkc <= 1) { if (k.Sq > 32) BKV(1, 4); else BKV(1, 1); }
  else if (nkc <= 2 || nwmax == 2) BKV(2, 1); else BKV(4, 1);
#undef BKV
}

// Two cores in one launch when their geometry (B, H, Sq, Sk, d_h = 64, 33..128 keys, more than 64 queries) is the same;
// otherwise -- or with MMNAS_MHA_PAIR=0 -- two launches.  Internal: the mixed chain of ops.hip.
int mha_core_fwd_pair(const mmnas_mha_desc* d0, const mmnas_mha_desc* d1, hipStream_t st) {
  MhaK k0, k1;
  int rc = fill(d0, k0, false);
  if (rc) return rc;
  if ((rc = fill(d1, k1, false))) return rc;
  static const int on = [] { const char* e = getenv("MMNAS_MHA_PAIR"); return !(e && e[0] == '0'); }();
  const int nkc = cdiv(k0.Sk, 32);
  const bool same = on && k0.B == k1.B && k0.H == k1.H && k0.Sq == k1.Sq && k0.Sk == k1.Sk && k0.dh == 64 && k1.dh == 64 &&
                    nkc > 2 && nkc <= 4 && k0.Sq > 64 && mha_nw() == 4 && !k0.qoff == !k1.qoff && !k0.koff == !k1.koff;
  if (!same) {
    if ((rc = mmnas_mha_core_fwd(d0, st))) return rc;
    return mmnas_mha_core_fwd(d1, st);
  }
  k0.nch = k1.nch = 1;
  const double bhqk = (double)k0.B * k0.H * k0.Sq * k0.Sk;
  if (k0.ldq == k0.ldk && k0.ldq == k0.ldv && k0.ldq == k0.ldo && k1.ldq == k0.ldq && k1.ldk == k0.ldq && k1.ldv == k0.ldq && k1.ldo == k0.ldq &&
      k0.qoff == k1.qoff && k0.koff == k1.koff) {   // round 6: both cores on the bf16 pipe, two workgroups per CU
    const float* Qs[2] = {k0.Q, k1.Q}; const float* Ks[2] = {k0.K, k1.K}; const float* Vs[2] = {k0.V, k1.V};
    const uint8_t* Ms[2] = {k0.mask, k1.mask}; const float* Bs[2] = {k0.biasT, k1.biasT};
    float* Os[2] = {k0.O, k1.O}; float* Ss[2] = {k0.stats, k1.stats};
    const DropCfg Ds[2] = {k0.drop, k1.drop};
    ProfScope ps16(MMNAS_K_MHA_FWD, 8.0 * bhqk * k0.dh,
                   4.0 * (2.0 * (double)k0.B * k0.H * k0.dh * (2.0 * k0.Sq + 2.0 * k0.Sk) + (k0.biasT ? bhqk : 0.0) + (k1.biasT ? bhqk : 0.0)), st);
    if (mha_fwd_b16_pair(k0.B, k0.H, k0.Sq, k0.Sk, k0.ldq, Qs, Ks, Vs, Ms, Bs, Os, Ss, Ds, k0.scale, k0.qoff, k0.koff, st))
      return check_launch("mha_core_fwd_pair");
  }
  ProfScope ps(MMNAS_K_MHA_FWD, 8.0 * bhqk * k0.dh,
               4.0 * (2.0 * (double)k0.B * k0.H * k0.dh * (2.0 * k0.Sq + 2.0 * k0.Sk) + (k0.biasT ? bhqk : 0.0) + (k1.biasT ? bhqk : 0.0)), st);
  const int nqb = cdiv(k0.Sq, 128);
  MMNAS_LAUNCH((mha_fwd_pair_kernel<64, 4, 4>), dim3(2 * nqb, k0.H, k0.B), dim3(256), 0, st, k0, k1, nqb);
  return check_launch("mha_core_fwd_pair");
}

}  // namespace mmnas

using namespace mmnas;

extern "C" int mmnas_mha_core_fwd(const mmnas_mha_desc* d, void* stream) {
  MhaK k;
  int rc = fill(d, k, false);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const double bhqk = (double)k.B * k.H * k.Sq * k.Sk;
  ProfScope ps(MMNAS_K_MHA_FWD, 4.0 * bhqk * k.dh,
               4.0 * ((double)k.B * k.H * k.dh * (2.0 * k.Sq + 2.0 * k.Sk) + (k.biasT ? bhqk : 0.0)), st);
  // d_h = 64, 65..128 keys (the image stream): both products on the bf16 pipe as exactly split operands (attention_bwd16.hip)
  if (k.dh == 64 && mha_fwd_b16_launch(k.B, k.H, k.Sq, k.Sk, k.ldq, k.ldk, k.ldv, k.ldo, k.Q, k.K, k.V, k.mask, k.biasT, k.O, k.stats,
                                       k.drop, k.scale, k.qoff, k.koff, st))
    return check_launch("mha_core_fwd");
  if (k.dh >= 64) { k.nch = k.dh / 64; launch_fwd<64>(k, st); }
  else if (k.dh == 32) { k.nch = 1; launch_fwd<32>(k, st); }
  else { k.nch = 1; launch_fwd<16>(k, st); }
  return check_launch("mha_core_fwd");
}

extern "C" int mmnas_mha_core_bwd(const mmnas_mha_desc* d, void* stream) {
  MhaK k;
  int rc = fill(d, k, true);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const double bhqk = (double)k.B * k.H * k.Sq * k.Sk;
  ProfScope ps(MMNAS_K_MHA_BWD, 10.0 * bhqk * k.dh,
               4.0 * ((double)k.B * k.H * k.dh * (4.0 * k.Sq + 4.0 * k.Sk) + (k.biasT ? 2.0 * bhqk : 0.0)), st);
  static const bool fused_on = !(getenv("MMNAS_MHA_BWD_FUSED") && getenv("MMNAS_MHA_BWD_FUSED")[0] == '0');
  const bool packed = k.qoff || k.koff;
  if (packed) MMNAS_REQUIRE((((uintptr_t)k.dK | (uintptr_t)k.dV) & 15) == 0, MMNAS_E_ARG, "mha_bwd: dK / dV alignment (packed rows run the fused kernel only)");
  if ((fused_on || packed) && k.dh == 64 && k.Sq <= 128 && k.Sk <= 128 && (((uintptr_t)k.dK | (uintptr_t)k.dV) & 15) == 0) {
    k.nch = 1;
    // 65..128 keys (the image stream): all five products on the bf16 pipe as exactly split operands (attention_bwd16.hip)
    if (mha_bwd_b16_launch(k.B, k.H, k.Sq, k.Sk, k.ldq, k.ldk, k.ldv, k.ldo, k.Q, k.K, k.V, (const float*)d->O, k.dO, k.mask, k.biasT,
                           k.stats, k.dQ, k.dK, k.dV, k.dbiasT, k.drop, k.scale, k.qoff, k.koff, st))
      return check_launch("mha_core_bwd");
    const dim3 grid(k.H, k.B);
    const int nkb = cdiv(k.Sk, 32);
    if (k.dbiasT) {
      if (nkb <= 1) MMNAS_LAUNCH((mha_bwd_fused_kernel<1, true>), grid, dim3(256), 0, st, k);
      else if (nkb <= 2) MMNAS_LAUNCH((mha_bwd_fused_kernel<2, true>), grid, dim3(256), 0, st, k);
      else MMNAS_LAUNCH((mha_bwd_fused_kernel<4, true>), grid, dim3(256), 0, st, k);
    } else {
      if (nkb <= 1) MMNAS_LAUNCH((mha_bwd_fused_kernel<1, false>), grid, dim3(256), 0, st, k);
      else if (nkb <= 2) MMNAS_LAUNCH((mha_bwd_fused_kernel<2, false>), grid, dim3(256), 0, st, k);
      else MMNAS_LAUNCH((mha_bwd_fused_kernel<4, false>), grid, dim3(256), 0, st, k);
    }
    return check_launch("mha_core_bwd");
  }
  if (k.dh > 64) {   // several head-dim chunks: delta needs the whole row first (dh <= 64: fused into the dQ kernel)
    const long n = (long)k.B * k.Sq * k.H;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    MMNAS_LAUNCH(mha_delta_kernel, dim3(blocks), dim3(256), 0, st, k.dO, (const float*)d->O, k.delta, k.B, k.H,
                       k.Sq, k.dh, k.ldo);
  }
  if (k.dh >= 64) { k.nch = k.dh / 64; launch_bwd<64>(k, st); }
  else if (k.dh == 32) { k.nch = 1; launch_bwd<32>(k, st); }
  else { k.nch = 1; launch_bwd<16>(k, st); }
  return check_launch("mha_core_bwd");
}

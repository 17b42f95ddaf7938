// Relation bias of ALL relation operators of a step in one launch per direction (round 5).
//
// Every RelSelfAtt of a network reads the SAME relation embedding: `rel = relu(linear_y_rel(raw))` is ONE stem layer
// (hygr_vqa.py:111, full_vqa.py:103) and an operator's bias  log(max(relu(linear_r(rel)), 1e-6))  (modules.py:231-235)
// depends on nothing the backbone computes.  relfused.hip recomputes the hidden layer `hid = relu(Wy raw + by)` once per
// operator and direction (320 of its ~1400 multiply-adds per element) and reduces its own dWy / dby partials; a supernet
// weight step ran it 5-6 times forward (21 us each) and backward (70 + 6 us each), an arch step 18 times forward.
//
// Here the operators of a step form the ROWS of one small matrix: row (operator n, head h) holds Wr_n[h, :].  Per tile of
// 32 (key, query) elements a wave computes, all on the fp32 MFMA (32x32x2) and chained through the accumulator layout:
//   forward   1. hid^T[j, e]  = relu(Wy_ext[j, :] . raw_ext[e, :])              8 MFMAs   (once, whatever the operator count)
//             2. r[row, e]    = Wr_all[row, :] . hid[:, e]  per 32-row tile      32 MFMAs  (8 operators of 4 heads / 4 of 8)
//                biasT_n[b, h, k, q] = log(max(r + br, 1e-6))                    128-byte coalesced stores per (row, tile)
//   backward  1. + 2. as above (one 32-row tile per launch)
//             3. dpre[row, e] = dbias_n[b, h, k, q] / r   (r >= 1e-6)
//             4. dhid^T[j, e] = relu'(hid) * sum_row Wr_all[row, j] dpre[row, e]  32 MFMAs  -- summed over the operators
//             5. dWr_all[row, j] += sum_e dpre[row, e] hid[j, e]                  32 MFMAs  (operands through a wave-private LDS image)
//                dWy_ext[j, c]   += sum_e dhid[j, e] raw_ext[e, c]                32 MFMAs 16x16x4 -- ONCE for all operators
//   then one reduction launch scatters the workgroups' partial rows into every operator's dWr / dbr and the shared dWy / dby.
// Algorithmic work per element (fwd): 2 R (C + 1) + 2 R rows flop; bytes 4 (C + rows).  Bound: MFMA (fp32, 157 TF/s); the
// [B, S, S, 64] relation tensor never exists (as in relfused.hip).  Ragged batches (valid n_b x n_b corner per sample)
// walk the same tile table as mmnas_rel_fused_bwd_ragged.
#include <stdlib.h>
#include <string.h>
#include "common.h"

#ifndef MMNAS_DBG_REL
#define MMNAS_DBG_REL 0     // 1: the timing switches of docs/LAB_NOTES.md "Round 5 notebook" (wrong results by design)
#endif
#define RM_DBG(bit) (MMNAS_DBG_REL && (p.dbg & (bit)))

namespace mmnas {

constexpr int RM_R = 64;        // REL_SIZE
constexpr int RM_CP = 8;        // raw channels padded (C + 1 <= 8: the column of ones carries by / yields dby)
constexpr int RM_ROWS = 32;     // head rows per row tile = one MFMA tile
constexpr int RM_NT_MAX = 3;    // row tiles per forward launch
constexpr int RM_LDH = RM_R + 4;        // LDS row stride of the hid / dhid image
constexpr int RM_LDP = RM_ROWS + 4;     // LDS row stride of the dpre image
constexpr int RM_ROW = RM_ROWS * RM_R + RM_R * RM_CP + RM_ROWS;   // partial row: [dWr_all 32 x 64 | dWy_ext 64 x 8 | dbr_all 32]

struct RelMultiK {
  const float* raw;
  int B, S, C, H, nrows, nops;
  int dbg;                               // timing experiments only, in a -DMMNAS_DBG_REL=1 build (MMNAS_REL_MULTI_DBG): 1 no stores, 2 no raw reloads, 4 no head-projection MFMAs
  const int* off; const int* toff;       // ragged batches (see relfused.hip); NULL = all S x S elements
  float* part;                           // backward: partial rows [grid][RM_ROW]
  const float* Wr[MMNAS_REL_MULTI_MAX];  // per operator of this launch: linear_r.weight [H, 64], .bias [H]
  const float* br[MMNAS_REL_MULTI_MAX];
  float* io[RM_NT_MAX * RM_ROWS];        // per ROW: forward biasT_n + h S^2 (written) / backward dbiasT_n + h S^2 (read)
};

// tile T of the whole batch -> sample and tile inside it (ragged); sample = B when T lies behind the last tile
__device__ __forceinline__ void rm_locate(const RelMultiK& p, int T, int ntiles, int& b, int& tb) {
  if (T >= ntiles) { b = p.B; tb = 0; return; }
  int lo = 0, hi = p.B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (p.toff[mid] <= T) lo = mid; else hi = mid;
  }
  b = lo; tb = T - p.toff[lo];
}

// Walks the tiles of one wave: tile t covers the flattened elements f = 32 tb + lane&31 of sample b, f = k * S + q (dense)
// or k * n_b + q over the valid corner (ragged).  elem() gives this lane's element of the current tile.
struct RmWalk {
  int tile, ntiles, nwaves, tpb, cb, ct, adv_b, adv_t;
  __device__ __forceinline__ void init(const RelMultiK& p, int ntiles_, int tpb_, int w) {
    ntiles = ntiles_; tpb = tpb_;
    nwaves = (int)gridDim.x * 4;
    adv_b = nwaves / tpb; adv_t = nwaves - adv_b * tpb;
    tile = (int)blockIdx.x * 4 + w;
    cb = tile / tpb; ct = tile - cb * tpb;
    if (p.toff) rm_locate(p, tile, ntiles, cb, ct);
  }
  __device__ __forceinline__ void next(const RelMultiK& p) {
    tile += nwaves;
    cb += adv_b; ct += adv_t;
    if (ct >= tpb) { ct -= tpb; ++cb; }
    if (p.toff) rm_locate(p, tile, ntiles, cb, ct);
  }
};

struct RmElem { bool ok; int b; unsigned k, q, fc; };

__device__ __forceinline__ RmElem rm_elem(const RelMultiK& p, int b, int tb, int l31) {
  RmElem e;
  const unsigned S = (unsigned)p.S, SS = S * S;
  const unsigned f = (unsigned)tb * 32u + (unsigned)l31;
  bool ok = b < p.B && f < SS;
  unsigned wq = S;
  if (p.off) {
    const int bb = b < p.B ? b : 0;
    const int n = p.off[bb + 1] - p.off[bb];
    ok = b < p.B && n > 0 && f < (unsigned)n * (unsigned)n;
    wq = (unsigned)max(n, 1);
  }
  const unsigned f0 = ok ? f : 0u;
  e.ok = ok; e.b = ok ? b : 0;
  e.k = f0 / wq; e.q = f0 - e.k * wq;
  e.fc = e.k * S + e.q;          // position in the padded [S_k, S_q] plane of biasT / dbiasT
  return e;
}

// the lane's raw row, extended by a one (-> by / dby) and zero padding.  A VECTOR value, not an array: an array that is
// re-filled inside the tile loop and read as `hh ? ex[2 s + 1] : ex[2 s]` became a scratch buffer indexed by hh.
typedef float f32x8 __attribute__((ext_vector_type(8)));
template <int C>
__device__ __forceinline__ f32x8 rm_raw(const RelMultiK& p, const RmElem& e) {
  const float* src = p.raw + (((size_t)e.b * p.S + e.q) * p.S + e.k) * C;
  f32x8 ex = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < C; ++c) { const float v = src[c]; ex[c] = e.ok ? v : 0.f; }   // (address clamped: the load is unconditional)
  ex[C] = e.ok ? 1.f : 0.f;
  return ex;
}

// step-2 A operands of all row tiles: sWrA[rt][step = 16 t + r][lane] = Wr_all[32 rt + (lane & 31)][32 t + acc_row(r, lane >> 5)]
// (zero for rows behind nrows); sBr[row] likewise
// sIo[row]: the row's plane of biasT (forward, written) / dbiasT (backward, read), NULL behind nrows -- per-lane row
// pointers come from LDS (the kernel-argument copy would cost 2 SGPRs per row: 64 of ~100)
template <bool WITH_T>
__device__ __forceinline__ void rm_stage_weights(const RelMultiK& p, int nt, float* sWrA, float* sWrB, float* sBr, float** sIo) {
  const int tid = threadIdx.x;
  for (int i = tid; i < nt * RM_ROWS * RM_R; i += 256) { sWrA[i] = 0.f; if (WITH_T) sWrB[i] = 0.f; }
  for (int i = tid; i < nt * RM_ROWS; i += 256) { sBr[i] = 0.f; sIo[i] = i < p.nrows ? p.io[i] : nullptr; }
  __syncthreads();
  const int H = p.H;
  for (int n = 0; n < p.nops; ++n) {
    const float* __restrict__ W = p.Wr[n];
    const float* __restrict__ bb = p.br[n];
    for (int i = tid; i < H * RM_R; i += 256) {
      const int h = i >> 6, j = i & 63;
      const int row = n * H + h, rt = row >> 5, rl = row & 31;
      const float v = W[i];
      {   // A[i = row][k = j]: j = 32 t + acc_row(r, hh)
        const int t = j >> 5, jj = j & 31, hh = (jj >> 2) & 1, r = (jj & 3) + 4 * (jj >> 3);
        sWrA[(rt * 32 + 16 * t + r) * 64 + hh * 32 + rl] = v;
      }
      if (WITH_T) {   // step 4, A[i = j][k = row]: row = acc_row(r, hh) (one row tile only)
        const int t = j >> 5, hh = (rl >> 2) & 1, r = (rl & 3) + 4 * (rl >> 3);
        sWrB[(16 * t + r) * 64 + hh * 32 + (j & 31)] = v;
      }
    }
    for (int h = tid; h < H; h += 256) sBr[n * H + h] = bb[h];
  }
  __syncthreads();
}

template <int C>
__device__ __forceinline__ void rm_wy_operand(const float* __restrict__ Wy, const float* __restrict__ by, int l31, int hh, float (*wyA)[4]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int j = 32 * t + l31, c = 2 * s4 + hh;
      wyA[t][s4] = c < C ? Wy[j * C + c] : (c == C ? by[j] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------------------ forward
// Y (tuning): s_nop behind every MFMA of the long dependent chains.  A wave whose next instruction is another MFMA holds the
// SIMD's issue port until the matrix pipe takes it (docs/LAB_NOTES.md, the round-3 overlap probe), so the other waves' vector
// work -- here a third of a tile: relu, log, addresses, stores -- cannot run beside the chain; the nop lets them in.
template <int Y> __device__ __forceinline__ void rm_yield() {
  if constexpr (Y == 1) asm volatile("s_nop 1");
  if constexpr (Y == 3) asm volatile("s_nop 3");
}

template <int C, int NT, int Y>
__global__ void __launch_bounds__(256) rel_multi_fwd_kernel(const RelMultiK p, int ntiles, int tpb, const float* __restrict__ Wy,
                                                            const float* __restrict__ by) {
  __shared__ float sWrA[NT * RM_ROWS * RM_R];
  __shared__ __attribute__((aligned(16))) float sBr[NT * RM_ROWS];
  __shared__ __attribute__((aligned(16))) float* sIo[NT * RM_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  rm_stage_weights<false>(p, NT, sWrA, nullptr, sBr, sIo);
  float wyA[2][4];
  rm_wy_operand<C>(Wy, by, l31, hh, wyA);
  const unsigned SS = (unsigned)p.S * (unsigned)p.S;
  RmWalk wk;
  wk.init(p, ntiles, tpb, w);
  RmElem cur = rm_elem(p, wk.cb, wk.ct, l31);
  f32x8 ext = rm_raw<C>(p, cur);
  while (wk.tile < ntiles) {
    f32x16 hid[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) hid[t] = mfma32(wyA[t][s4], hh ? ext[2 * s4 + 1] : ext[2 * s4], hid[t]);
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = fmaxf(hid[t][r], 0.f);
    }
    const bool ok = cur.ok;
    const unsigned eoff = (unsigned)cur.b * (unsigned)p.H * SS + cur.fc;   // (host: B H S^2 < 2^31)
    wk.next(p);
    cur = rm_elem(p, wk.cb, wk.ct, l31);
    if (!RM_DBG(2)) ext = rm_raw<C>(p, cur);         // the next tile's raw row: in flight during this tile's MFMA chain
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) {
      if (rt * RM_ROWS >= p.nrows) break;
      f32x16 rr;
#pragma unroll
      for (int r = 0; r < 16; ++r) rr[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { if (!RM_DBG(4) || r == 0) { rr = mfma32(sWrA[(rt * 32 + 16 * t + r) * 64 + lane], hid[t][r], rr); rm_yield<Y>(); } }
#pragma unroll
      for (int u = 0; u < 4; ++u) {        // registers 4u .. 4u + 3 hold rows 8u + 4 hh + 0..3: contiguous table entries
        const int row = rt * RM_ROWS + 8 * u + 4 * hh;
        const float4 bq = *reinterpret_cast<const float4*>(sBr + row);
        const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float* const o = sIo[row + i];
          if (ok && o && !(RM_DBG(1) && rr[4 * u + i] != 12345.f)) o[eoff] = __logf(fmaxf(rr[4 * u + i] + bv[i], 1e-6f));   // max(relu(r), 1e-6) == max(r, 1e-6)
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------------- backward
template <int C, int Y>
__global__ void __launch_bounds__(256, 2) rel_multi_bwd_kernel(const RelMultiK p, int ntiles, int tpb, const float* __restrict__ Wy,
                                                               const float* __restrict__ by) {
  __shared__ __attribute__((aligned(16))) float sHidAll[4][32 * RM_LDH];
  __shared__ __attribute__((aligned(16))) float sDpreAll[4][32 * RM_LDP];
  __shared__ __attribute__((aligned(16))) float sRawAll[4][32 * RM_CP];
  __shared__ float sWrA[RM_ROWS * RM_R];
  __shared__ float sWrB[RM_ROWS * RM_R];
  __shared__ __attribute__((aligned(16))) float sBr[RM_ROWS];
  __shared__ __attribute__((aligned(16))) float* sIo[RM_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int l15 = lane & 15, q4 = lane >> 4;
  float* sHid = sHidAll[w];
  float* sDpre = sDpreAll[w];
  float* sRaw = sRawAll[w];
  rm_stage_weights<true>(p, 1, sWrA, sWrB, sBr, sIo);
  float wyA[2][4];
  rm_wy_operand<C>(Wy, by, l31, hh, wyA);

  f32x16 accWr[2];
  f32x4 accWy16[4];
  float accbr = 0.f;      // lane (row = l31, half): sum over the elements e = half (mod 2) of dpre[row, e]
#pragma unroll
  for (int r = 0; r < 16; ++r) { accWr[0][r] = 0.f; accWr[1][r] = 0.f; }
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) accWy16[n][r] = 0.f;

  const unsigned SS = (unsigned)p.S * (unsigned)p.S;
  // dbias of this lane's 16 rows for one tile (row pointers from LDS; rows behind nrows are NULL and read as zero)
  auto load_db = [&](const RmElem& e, float* db) __attribute__((always_inline)) {
    const unsigned eoff = (unsigned)e.b * (unsigned)p.H * SS + e.fc;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* const d = sIo[8 * u + 4 * hh + i];
        const bool live = e.ok && d != nullptr;
        const float v = (live ? d : p.raw)[live ? eoff : 0u];     // (the load itself is unconditional: a valid address either way)
        db[4 * u + i] = live ? v : 0.f;
      }
  };
  RmWalk wk;
  wk.init(p, ntiles, tpb, w);
  RmElem cur = rm_elem(p, wk.cb, wk.ct, l31);
  float db[16];
  f32x8 ext = rm_raw<C>(p, cur);
  load_db(cur, db);
  while (wk.tile < ntiles) {
    // 1. hidden layer (transposed: rows j, columns e); relu and its gate as one bit per accumulator register
    f32x16 hid[2];
    unsigned gm = 0u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) hid[t] = mfma32(wyA[t][s4], hh ? ext[2 * s4 + 1] : ext[2 * s4], hid[t]);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        gm |= hid[t][r] > 0.f ? (1u << (16 * t + r)) : 0u;
        hid[t][r] = fmaxf(hid[t][r], 0.f);
      }
    }
    // 2. r[row, e]
    f32x16 rr;
#pragma unroll
    for (int r = 0; r < 16; ++r) rr[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { rr = mfma32(sWrA[(16 * t + r) * 64 + lane], hid[t][r], rr); rm_yield<Y>(); }
    // 3. d(log max(r, 1e-6)) / dr
    float dpre[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 bq = *reinterpret_cast<const float4*>(sBr + 8 * u + 4 * hh);
      const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float rv = rr[4 * u + i] + bv[i];
        dpre[4 * u + i] = rv >= 1e-6f ? db[4 * u + i] / rv : 0.f;
      }
    }
    // the wave's LDS image: hid as [e][j], dpre as [e][row], the raw rows (LDS operations of one wave execute in order; the
    // waits only keep the compiler from moving a read above the write it depends on)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        *reinterpret_cast<float4*>(sHid + l31 * RM_LDH + 32 * t + 8 * u + 4 * hh) =
            make_float4(hid[t][4 * u], hid[t][4 * u + 1], hid[t][4 * u + 2], hid[t][4 * u + 3]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      *reinterpret_cast<float4*>(sDpre + l31 * RM_LDP + 8 * u + 4 * hh) = make_float4(dpre[4 * u], dpre[4 * u + 1], dpre[4 * u + 2], dpre[4 * u + 3]);
    if (hh == 0) {
      *reinterpret_cast<float4*>(sRaw + l31 * RM_CP) = make_float4(ext[0], ext[1], ext[2], ext[3]);
      *reinterpret_cast<float4*>(sRaw + l31 * RM_CP + 4) = make_float4(ext[4], ext[5], ext[6], ext[7]);
    }
    // the next tile's raw row and bias gradients: into the registers this tile is done with, in flight during steps 4-5
    wk.next(p);
    cur = rm_elem(p, wk.cb, wk.ct, l31);
    ext = rm_raw<C>(p, cur);
    load_db(cur, db);
    // 4. gradient of the hidden layer, summed over every operator's heads, gated by relu'
    f32x16 dh[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[t][r] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dh[t] = mfma32(sWrB[(16 * t + r) * 64 + lane], dpre[r], dh[t]); rm_yield<Y>(); }
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[t][r] = (gm >> (16 * t + r)) & 1u ? dh[t][r] : 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // 5a. dWr_all[row, j] += sum_e dpre[row, e] hid[j, e]          (k index = e: two elements per MFMA)
#pragma unroll
    for (int s0 = 0; s0 < 16; s0 += 4) {
      float av[4], b0[4], b1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = 2 * (s0 + u) + hh;
        av[u] = sDpre[e * RM_LDP + l31];
        b0[u] = sHid[e * RM_LDH + l31];
        b1[u] = sHid[e * RM_LDH + 32 + l31];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        accbr += av[u];                     // dbr_all[row] = sum_e dpre[row, e]
        accWr[0] = mfma32(av[u], b0[u], accWr[0]);
        rm_yield<Y>();
        accWr[1] = mfma32(av[u], b1[u], accWr[1]);
        rm_yield<Y>();
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        *reinterpret_cast<float4*>(sHid + l31 * RM_LDH + 32 * t + 8 * u + 4 * hh) =
            make_float4(dh[t][4 * u], dh[t][4 * u + 1], dh[t][4 * u + 2], dh[t][4 * u + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // 5b. dWy_ext[j, c] += sum_e dhid[j, e] raw_ext[e, c]   (column C of raw_ext is 1: dby); 16-wide tiles halve the padding
    {
      const int ccl = l15 < RM_CP ? l15 : 0;
#pragma unroll
      for (int s0 = 0; s0 < 8; s0 += 2) {
        float av[2][4], bv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = 4 * (s0 + u) + q4;
          bv[u] = sRaw[e * RM_CP + ccl];
#pragma unroll
          for (int m = 0; m < 4; ++m) av[u][m] = sHid[e * RM_LDH + 16 * m + l15];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float bb = l15 < RM_CP ? bv[u] : 0.f;
#pragma unroll
          for (int m = 0; m < 4; ++m) accWy16[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][m], bb, accWy16[m], 0, 0, 0);
        }
      }
    }
  }

  // ---- the workgroup's partial row: the 4 waves add their accumulators in wave order through LDS (fixed order:
  //      reproducible), the last one writes the row ----
  __syncthreads();
  float* srow = &sHidAll[0][0];
  static_assert(4 * 32 * RM_LDH >= RM_ROW, "partial row does not fit the hid images");
  float* grow = p.part + (size_t)blockIdx.x * RM_ROW;
  const float brsum = accbr + __shfl_xor(accbr, 32, 64);   // even + odd elements: lane l31 = row holds dbr_all[row] of this wave
  for (int turn = 0; turn < 4; ++turn) {
    if (w == turn) {
      const bool first = turn == 0, last = turn == 3;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o1 = acc_row(r, hh) * RM_R + 32 * t + l31;                       // dWr_all[row][j]
          const float v1 = accWr[t][r] + (first ? 0.f : srow[o1]);
          if (last) grow[o1] = v1; else srow[o1] = v1;
        }
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (l15 < RM_CP) {
            const int o2 = RM_ROWS * RM_R + (16 * n + 4 * q4 + r) * RM_CP + l15;     // dWy_ext[j][c]
            const float v2 = accWy16[n][r] + (first ? 0.f : srow[o2]);
            if (last) grow[o2] = v2; else srow[o2] = v2;
          }
      if (hh == 0) {
        const int o3 = RM_ROWS * RM_R + RM_R * RM_CP + l31;                           // dbr_all[row]
        const float v3 = brsum + (first ? 0.f : srow[o3]);
        if (last) grow[o3] = v3; else srow[o3] = v3;
      }
    }
    __syncthreads();
  }
}

struct RelMultiRed {
  const float* part; int nrows_part, C, H, nrows;
  float* dWr[MMNAS_REL_MULTI_MAX]; float* dbr[MMNAS_REL_MULTI_MAX];
  float* dWy; float* dby;
};

// sum the partial rows and add into the parameter gradients (single writer per output: plain +=, fixed order)
__global__ void __launch_bounds__(1024) rel_multi_reduce_kernel(const RelMultiRed p) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  if ((int)blockIdx.x < RM_ROWS && (int)blockIdx.x >= p.nrows) return;   // a 64-column block of the dWr region = one row
  float s = 0.f;
  if (col < RM_ROW) {
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = g;
    for (; r + 48 < p.nrows_part; r += 64) {
      s += p.part[(size_t)r * RM_ROW + col];
      s1 += p.part[(size_t)(r + 16) * RM_ROW + col];
      s2 += p.part[(size_t)(r + 32) * RM_ROW + col];
      s3 += p.part[(size_t)(r + 48) * RM_ROW + col];
    }
    for (; r < p.nrows_part; r += 16) s += p.part[(size_t)r * RM_ROW + col];
    s += (s1 + s2) + s3;
  }
  __shared__ float red[16][64];
  red[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g != 0 || col >= RM_ROW) return;
  const int cl = threadIdx.x & 63;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) v += red[i][cl];
  if (col < RM_ROWS * RM_R) {
    const int row = col / RM_R, j = col - row * RM_R;     // (uniform over the block)
    if (row < p.nrows) { const int n = row / p.H, h = row - n * p.H; p.dWr[n][h * RM_R + j] += v; }
  } else if (col < RM_ROWS * RM_R + RM_R * RM_CP) {
    const int o = col - RM_ROWS * RM_R, j = o / RM_CP, c = o - j * RM_CP;
    if (c < p.C) p.dWy[j * p.C + c] += v;
    else if (c == p.C) p.dby[j] += v;
  } else {
    const int row = col - RM_ROWS * RM_R - RM_R * RM_CP;
    if (row < p.nrows) { const int n = row / p.H, h = row - n * p.H; p.dbr[n][h] += v; }
  }
}

static int rm_check(const mmnas_rel_multi* m, const char* who) {
  MMNAS_REQUIRE(m, MMNAS_E_ARG, "%s: null descriptor", who);
  MMNAS_REQUIRE(m->B > 0 && m->S > 0, MMNAS_E_SHAPE, "%s: B=%d S=%d", who, m->B, m->S);
  MMNAS_REQUIRE(m->R == RM_R, MMNAS_E_SHAPE, "%s: REL_SIZE=%d (the fused path handles 64)", who, m->R);
  MMNAS_REQUIRE(m->C == 3 || m->C == 4, MMNAS_E_SHAPE, "%s: %d raw relation channels (3 or 4)", who, m->C);
  MMNAS_REQUIRE(m->H >= 1 && m->H <= RM_ROWS && RM_ROWS % m->H == 0, MMNAS_E_SHAPE, "%s: H=%d heads (a divisor of 32)", who, m->H);
  MMNAS_REQUIRE(m->n_ops >= 1 && m->n_ops <= MMNAS_REL_MULTI_MAX, MMNAS_E_SHAPE, "%s: %d operators (1..%d)", who, m->n_ops, MMNAS_REL_MULTI_MAX);
  MMNAS_REQUIRE(m->raw && m->Wy && m->by, MMNAS_E_ARG, "%s: null pointer", who);
  MMNAS_REQUIRE((long)m->B * m->H * m->S * m->S < (1l << 31), MMNAS_E_SHAPE, "%s: B H S^2 must fit 31 bits", who);
  if (m->off) MMNAS_REQUIRE(m->tile_off && m->ntiles >= 0, MMNAS_E_ARG, "%s: ragged batches need the tile offsets", who);
  return MMNAS_OK;
}

static long rm_tiles_per_b(int S) { return ((long)S * S + 31) / 32; }
static int rm_grid(long ntiles, int per_cu) {
  const long wgs = (ntiles + 3) / 4, cap = 256l * per_cu;
  return (int)(wgs < 1 ? 1 : (wgs < cap ? wgs : cap));
}

}  // namespace mmnas

using namespace mmnas;

extern "C" int mmnas_rel_multi_supported(int C, int R, int H) { return R == RM_R && (C == 3 || C == 4) && H >= 1 && H <= RM_ROWS && RM_ROWS % H == 0; }

extern "C" size_t mmnas_rel_multi_bwd_ws_floats(int B, int S) {
  return (size_t)rm_grid((long)B * rm_tiles_per_b(S), 2) * RM_ROW;
}

extern "C" int mmnas_rel_multi_fwd(const mmnas_rel_multi* m, void* stream) {
  int rc = rm_check(m, "rel_multi_fwd");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int tpb = (int)rm_tiles_per_b(m->S);
  const int ntiles = m->off ? m->ntiles : m->B * tpb;
  if (ntiles == 0) return MMNAS_OK;
  const int ops_per_tile = RM_ROWS / m->H;                    // an operator's heads never straddle two row tiles
  const int max_ops = ops_per_tile * RM_NT_MAX;
  const unsigned SS = (unsigned)m->S * (unsigned)m->S;
  for (int o0 = 0; o0 < m->n_ops; o0 += max_ops) {
    const int n = m->n_ops - o0 < max_ops ? m->n_ops - o0 : max_ops;
    RelMultiK k;
    memset(&k, 0, sizeof(k));
    k.raw = m->raw; k.B = m->B; k.S = m->S; k.C = m->C; k.H = m->H; k.off = m->off; k.toff = m->tile_off;
    { static const int dbg = [] { const char* e = getenv("MMNAS_REL_MULTI_DBG"); return e && e[0] ? atoi(e) : 0; }(); k.dbg = dbg; }
    // rows: operator j of this launch holds rows j H .. j H + H - 1 (H divides 32: no operator straddles two row tiles)
    for (int j = 0; j < n; ++j) {
      MMNAS_REQUIRE(m->Wr[o0 + j] && m->br[o0 + j] && m->biasT[o0 + j], MMNAS_E_ARG, "rel_multi_fwd: operator %d: null pointer", o0 + j);
      for (int h = 0; h < m->H; ++h) k.io[j * m->H + h] = m->biasT[o0 + j] + (size_t)h * SS;
    }
    const int slots = n * m->H;
    for (int j = 0; j < n; ++j) { k.Wr[j] = m->Wr[o0 + j]; k.br[j] = m->br[o0 + j]; }
    k.nops = n; k.nrows = slots;
    const int nt = (slots + RM_ROWS - 1) / RM_ROWS;
    static const int fwd_per_cu = [] { const char* e = getenv("MMNAS_REL_MULTI_WGS"); return e && e[0] ? atoi(e) : 3; }();   // (tuning: workgroups per CU)
    const int grid = rm_grid(ntiles, fwd_per_cu);
    const double ne = m->off ? 32.0 * ntiles : (double)m->B * SS;
    ProfScope ps(MMNAS_K_REL_FWD, 2.0 * ne * (RM_R * (m->C + 1) + (double)n * m->H * RM_R), 4.0 * ne * (m->C + n * m->H), st);
    static const int yield = [] { const char* e = getenv("MMNAS_REL_MULTI_YIELD"); return e && e[0] ? atoi(e) : 0; }();
#define RM_FWD(CC, NTT) do { if (yield == 1) MMNAS_LAUNCH((rel_multi_fwd_kernel<CC, NTT, 1>), dim3(grid), dim3(256), 0, st, k, ntiles, tpb, m->Wy, m->by); \
      else if (yield == 3) MMNAS_LAUNCH((rel_multi_fwd_kernel<CC, NTT, 3>), dim3(grid), dim3(256), 0, st, k, ntiles, tpb, m->Wy, m->by); \
      else MMNAS_LAUNCH((rel_multi_fwd_kernel<CC, NTT, 0>), dim3(grid), dim3(256), 0, st, k, ntiles, tpb, m->Wy, m->by); } while (0)
    if (m->C == 4) { if (nt == 1) RM_FWD(4, 1); else if (nt == 2) RM_FWD(4, 2); else RM_FWD(4, 3); }
    else { if (nt == 1) RM_FWD(3, 1); else if (nt == 2) RM_FWD(3, 2); else RM_FWD(3, 3); }
#undef RM_FWD
    if ((rc = check_launch("rel_multi_fwd"))) return rc;
  }
  return MMNAS_OK;
}

extern "C" int mmnas_rel_multi_bwd(const mmnas_rel_multi* m, void* stream) {
  int rc = rm_check(m, "rel_multi_bwd");
  if (rc) return rc;
  MMNAS_REQUIRE(m->dWy && m->dby && m->ws, MMNAS_E_ARG, "rel_multi_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int tpb = (int)rm_tiles_per_b(m->S);
  const int ntiles = m->off ? m->ntiles : m->B * tpb;
  if (ntiles == 0) return MMNAS_OK;
  const int max_ops = RM_ROWS / m->H;
  const unsigned SS = (unsigned)m->S * (unsigned)m->S;
  for (int o0 = 0; o0 < m->n_ops; o0 += max_ops) {
    const int n = m->n_ops - o0 < max_ops ? m->n_ops - o0 : max_ops;
    RelMultiK k;
    RelMultiRed red;
    memset(&k, 0, sizeof(k));
    memset(&red, 0, sizeof(red));
    k.raw = m->raw; k.B = m->B; k.S = m->S; k.C = m->C; k.H = m->H; k.off = m->off; k.toff = m->tile_off;
    k.part = m->ws; k.nops = n; k.nrows = n * m->H;
    for (int j = 0; j < n; ++j) {
      MMNAS_REQUIRE(m->Wr[o0 + j] && m->br[o0 + j] && m->dbiasT[o0 + j] && m->dWr[o0 + j] && m->dbr[o0 + j], MMNAS_E_ARG,
                    "rel_multi_bwd: operator %d: null pointer", o0 + j);
      k.Wr[j] = m->Wr[o0 + j]; k.br[j] = m->br[o0 + j];
      red.dWr[j] = m->dWr[o0 + j]; red.dbr[j] = m->dbr[o0 + j];
      for (int h = 0; h < m->H; ++h) k.io[j * m->H + h] = const_cast<float*>(m->dbiasT[o0 + j]) + (size_t)h * SS;
    }
    const int grid = rm_grid(ntiles, 2);
    red.part = m->ws; red.nrows_part = grid; red.C = m->C; red.H = m->H; red.nrows = k.nrows; red.dWy = m->dWy; red.dby = m->dby;
    const double ne = m->off ? 32.0 * ntiles : (double)m->B * SS;
    const double rows = (double)n * m->H;
    ProfScope ps(MMNAS_K_REL_BWD, 2.0 * ne * (2.0 * RM_R * (m->C + 1) + 3.0 * rows * RM_R), 4.0 * ne * (m->C + rows), st);
    static const int yield = [] { const char* e = getenv("MMNAS_REL_MULTI_YIELD"); return e && e[0] ? atoi(e) : 0; }();
#define RM_BWD(CC) do { if (yield == 1) MMNAS_LAUNCH((rel_multi_bwd_kernel<CC, 1>), dim3(grid), dim3(256), 0, st, k, ntiles, tpb, m->Wy, m->by); \
      else if (yield == 3) MMNAS_LAUNCH((rel_multi_bwd_kernel<CC, 3>), dim3(grid), dim3(256), 0, st, k, ntiles, tpb, m->Wy, m->by); \
      else MMNAS_LAUNCH((rel_multi_bwd_kernel<CC, 0>), dim3(grid), dim3(256), 0, st, k, ntiles, tpb, m->Wy, m->by); } while (0)
    if (m->C == 4) RM_BWD(4); else RM_BWD(3);
#undef RM_BWD
    MMNAS_LAUNCH(rel_multi_reduce_kernel, dim3(cdiv(RM_ROW, 64)), dim3(1024), 0, st, red);
    if ((rc = check_launch("rel_multi_bwd"))) return rc;
  }
  return MMNAS_OK;
}

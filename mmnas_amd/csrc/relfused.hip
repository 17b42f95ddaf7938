// Lazy relation handle: the relation bias of RelMHAtt computed straight from the
// RAW box-geometry tensor, fusing the stem's  rel = relu(linear_y_rel(raw))  (hygr_vqa.py:111,
// full_vqa.py:103) with  bias = log(max(relu(linear_r(rel)), 1e-6))  (modules.py:231-235):
//
//     hid[j]   = relu(by[j] + sum_c Wy[j,c] * raw[b,q,k,c])            j < R = 64,  c < C (4 or 3)
//     r[h]     = relu(br[h] + sum_j Wr[h,j] * hid[j])
//     biasT[b,h,k,q] = log(max(r[h], 1e-6))
//
// The [B,S,S,64] tensor (164 MB at B=64,S=100), its gradient and their accumulation over the 4-5
// relation operators of a network never exist: forward reads 10 MB and writes the 20 MB bias, the
// hidden layer lives in registers (768 FMA per element: 0.5 GFLOP per launch, compute-trivial).
// Backward needs no input gradient (raw is data); the parameter gradients
//     dWr[h,j] = sum_e dpre[e,h] hid[e,j],   dWy[j,c] = sum_e dhid[e,j] raw[e,c],   dbr, dby
// are reductions over B*S*S = 640k elements: each workgroup stages hid / dhid of 256 elements in LDS
// and contracts them on the fp32 MFMA (accumulating over its elements in registers), then writes one
// partial row to a workspace that a second tiny kernel sums (no hot-address atomics, deterministic).
#include <stdlib.h>
#include "common.h"

namespace mmnas {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int RF_R = 64;      // REL_SIZE handled by the fused path
constexpr int RF_CP = 8;      // raw channels padded (C + 1 <= 8: the extra column of ones yields dby)
constexpr int RF_HP = 32;     // heads padded to one MFMA tile

struct RelFusedK {
  const float* raw; const float* Wy; const float* by; const float* Wr; const float* br;
  float* biasT; const float* dbiasT; float* part;
  int B, Sq, Sk, C, H, nbq, nbk;
  // RAGGED batches (self-attention over the first n_b of the S rows of sample b, n_b = off[b+1] - off[b]; the rows behind
  // are padding whose bias nobody reads and whose bias gradient is exactly zero): forward skips them; backward walks only
  // the n_b x n_b valid elements of every sample -- tile t of sample b covers valid elements 32 t .. 32 t + 31, element
  // f' = (key f' / n_b, query f' % n_b); toff[b] = first tile of sample b, toff[B] = the tile count.  NULL: all S x S.
  const int* off; const int* toff;
};

// (ragged) tile T of the whole batch -> sample and tile inside it; sample = B when T lies behind the last tile
__device__ __forceinline__ void rf_locate(const RelFusedK& p, int T, int ntiles, int& b, int& tb) {
  if (T >= ntiles) { b = p.B; tb = 0; return; }
  int lo = 0, hi = p.B;          // toff[lo] <= T < toff[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (p.toff[mid] <= T) lo = mid; else hi = mid;
  }
  b = lo; tb = T - p.toff[lo];
}

// element owned by a thread of workgroup-batch `batch`: 64 consecutive q x 4 consecutive k of one b
__device__ __forceinline__ bool rf_coords(const RelFusedK& p, long batch, int tid, int& b, int& q, int& k) {
  const int per_b = p.nbq * p.nbk;
  b = (int)(batch / per_b);
  const int rem = (int)(batch - (long)b * per_b);
  const int kb = rem / p.nbq, qb = rem - kb * p.nbq;
  q = qb * 64 + (tid & 63);
  k = kb * 4 + (tid >> 6);
  return q < p.Sq && k < p.Sk;
}

// The read-only parameter pointers travel as separate `const float* __restrict__` kernel arguments (not
// inside the struct): only then does hipcc prove them invariant and wave-uniform and fetch the weights
// with scalar loads; as struct members they became 142 per-lane global_load + s_waitcnt pairs.
template <int C>
__device__ __forceinline__ void rf_hidden(const RelFusedK& p, const float* __restrict__ Wy, const float* __restrict__ by,
                                          bool ok, int b, int q, int k, float* rawv, float* hid) {
  const float* src = p.raw + (((size_t)b * p.Sq + (ok ? q : 0)) * p.Sk + (ok ? k : 0)) * C;
#pragma unroll
  for (int c = 0; c < C; ++c) rawv[c] = src[c];
#pragma unroll
  for (int j = 0; j < RF_R; ++j) {
    float a = by[j];
#pragma unroll
    for (int c = 0; c < C; ++c) a += Wy[j * C + c] * rawv[c];
    hid[j] = fmaxf(a, 0.f);
  }
}

template <int C>
__global__ void __launch_bounds__(256) rel_fused_fwd_kernel(const RelFusedK p, const float* __restrict__ Wy,
                                                            const float* __restrict__ by, const float* __restrict__ Wr,
                                                            const float* __restrict__ br) {
  int b, q, k;
  bool ok = rf_coords(p, blockIdx.x, threadIdx.x, b, q, k);
  if (p.off) {   // ragged: elements of padding rows / columns are never read
    const int n = p.off[b + 1] - p.off[b];
    ok = ok && q < n && k < n;
    if (!ok) return;
  }
  float rawv[C], hid[RF_R];
  rf_hidden<C>(p, Wy, by, ok, b, q, k, rawv, hid);
  if (!ok) return;
  for (int h = 0; h < p.H; ++h) {
    float r = br[h];
    const float* w = Wr + h * RF_R;
#pragma unroll
    for (int j = 0; j < RF_R; ++j) r += w[j] * hid[j];
    p.biasT[(((size_t)b * p.H + h) * p.Sk + k) * p.Sq + q] = __logf(fmaxf(r, 1e-6f));   // max(relu(r),1e-6) == max(r,1e-6)
  }
}

// workspace row layout per workgroup: [ dWr: HP x R | dWy_ext: R x CP | dbr: HP ]
constexpr int RF_ROW = RF_HP * RF_R + RF_R * RF_CP + RF_HP;
constexpr int RF_LDH = RF_R + 4;   // LDS row stride of the hid / dhid image (16-byte aligned rows, odd in 16-byte units)

// Backward, one 32-element tile per wave at a time, everything on the fp32 MFMA (32x32x2), chained through the
// accumulator layout so no intermediate leaves the registers until the two parameter-gradient contractions:
//   1. hid^T[j,e]  = relu(Wy_ext[j,:] . raw_ext[e,:])        A = Wy|by (constant per lane), B = the lane's raw row
//   2. r[h,e]      = Wr[h,:] . hid[:,e]                       B = the hid accumulators themselves (k runs in
//                                                               accumulator order; A = Wr gathered in that order)
//   3. dpre[h,e]   = dbias / r  (r >= 1e-6)                   lane (e, half) owns heads 8g + 4*half + 0..3
//   4. dhid^T[j,e] = relu'(hid) * Wr[:,j] . dpre[:,e]         B = dpre registers, A = Wr^T (constant per lane)
//   5. dWr[h,j] += sum_e dpre[h,e] hid[j,e],  dWy_ext[j,c] += sum_e dhid[j,e] raw_ext[e,c]: the reduction index e
//      sits on the lanes of both operands, so hid / dhid / dpre / raw go through a wave-private LDS image
//      ([e][.] rows, no workgroup barrier anywhere) and come back with e as the MFMA k index.
// The previous version kept hid[64] and dhid[64] per thread in VGPRs (1 wave per SIMD) and streamed the
// weights through ~1300 scalar loads per element: 285 us per launch, latency-bound.
template <int C, int NG>   // NG = groups of 8 heads (H <= 8 * NG)
__global__ void __launch_bounds__(256, NG == 1 ? 2 : 1)
rel_fused_bwd_kernel(const RelFusedK p, int ntiles, int tiles_per_b, const float* __restrict__ Wy,
                     const float* __restrict__ by, const float* __restrict__ Wr, const float* __restrict__ br) {
  constexpr int DP = 8 * NG;
  __shared__ __attribute__((aligned(16))) float sHidAll[4][32 * RF_LDH];
  __shared__ __attribute__((aligned(16))) float sDpreAll[4][32 * DP];
  __shared__ __attribute__((aligned(16))) float sRawAll[4][32 * RF_CP];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  float* sHid = sHidAll[w];
  float* sDpre = sDpreAll[w];
  float* sRaw = sRawAll[w];
  const int H = p.H;

  // ---- per-lane constant MFMA operands ----
  float wyA[2][4];       // step 1, A[i = j][k = c]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int j = 32 * t + l31, c = 2 * s4 + hh;
      wyA[t][s4] = c < C ? Wy[j * C + c] : (c == C ? by[j] : 0.f);
    }
  // step 2, A[i = h][k = j in accumulator order]: 32 values per lane, identical for the 4 waves -> one LDS image
  // [step][lane] (keeps 32 VGPRs free: with them in registers the kernel spilled)
  __shared__ float sWrA[32 * 64];
  if (w == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        sWrA[(16 * t + r) * 64 + lane] = l31 < H ? Wr[l31 * RF_R + 32 * t + acc_row(r, hh)] : 0.f;
  }
  __syncthreads();
  float wrT[2][NG][4];   // step 4, A[i = j][k = h]
  float brv[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = 8 * g + r + 4 * hh;
      brv[g][r] = h < H ? br[h] : 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) wrT[t][g][r] = h < H ? Wr[h * RF_R + 32 * t + l31] : 0.f;
    }

  // Parameter-gradient accumulators.  H <= 8 (NG == 1, every shipped configuration): the two contractions of
  // step 5 run on v_mfma_f32_16x16x4_f32 -- their operands come from LDS, so the layout is free, and 16-wide tiles
  // halve the zero padding (8 heads / 5 raw channels) of the 32x32x2 form: 64 instead of 128 MFMA-cycles per element.
  constexpr bool T16 = NG == 1;
  f32x16 accWr[T16 ? 1 : 2], accWy[T16 ? 1 : 2];   // (32x32x2 form; one dummy entry when unused)
  f32x4 accWr16[4], accWy16[4];                     // dWr[h = 4*q4 + r][j = 16 n + l15], dWy_ext[j = 16 m + 4*q4 + r][c = l15]
#pragma unroll
  for (int t = 0; t < (T16 ? 1 : 2); ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { accWr[t][r] = 0.f; accWy[t][r] = 0.f; }
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) { accWr16[n][r] = 0.f; accWy16[n][r] = 0.f; }
  const int l15 = lane & 15, q4 = lane >> 4;
  float accbr[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) accbr[g][r] = 0.f;

  // 32-bit index arithmetic throughout (host: B * tiles_per_b < 2^31, Sq * Sk < 2^31): the 64-bit divisions this
  // used to do per tile and lane are emulated in hundreds of VALU cycles each.  (batch, tile-in-batch) advance
  // incrementally by the grid stride.
  const unsigned SS = (unsigned)p.Sq * (unsigned)p.Sk;
  const int nwaves = (int)gridDim.x * 4;
  const int adv_b = nwaves / tiles_per_b, adv_t = nwaves - adv_b * tiles_per_b;
  // loads of one tile: the lane's raw row (gathered, 16 B) and its heads' dbias; issued one tile ahead
  float ext[RF_CP], db[NG][4];
  auto tile_load = [&](int b, int tb, float* ex, float (*dbv)[4]) {
    // element of this lane: b, flattened f = k * Sq + q (dbiasT is contiguous in f)
    const unsigned f = (unsigned)tb * 32u + (unsigned)l31;
    bool ok = b < p.B && f < SS;
    unsigned wq = (unsigned)p.Sq;            // row length of the flattened (key, query) index f
    if (p.off) {                             // ragged: f runs over the n x n valid elements of the sample
      const int bb = b < p.B ? b : 0;
      const unsigned n = (unsigned)max(p.off[bb + 1] - p.off[bb], 1);
      ok = b < p.B && f < n * n && p.off[bb + 1] > p.off[bb];
      wq = n;
    }
    const unsigned fc0 = ok ? f : 0u;
    const int bc = ok ? b : 0;
    const unsigned k = fc0 / wq, q = fc0 - k * wq;
    const unsigned fc = k * (unsigned)p.Sq + q;   // position in the padded [Sk, Sq] plane of dbiasT
    const float* src = p.raw + (((size_t)bc * p.Sq + q) * p.Sk + k) * C;
#pragma unroll
    for (int c = 0; c < RF_CP; ++c) ex[c] = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) ex[c] = ok ? src[c] : 0.f;   // (address clamped above: the load itself is unconditional)
    ex[C] = ok ? 1.f : 0.f;
    const float* dbp = p.dbiasT + (size_t)bc * H * SS + fc;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int h = 8 * g + r + 4 * hh;
        const float v = dbp[(size_t)(h < H ? h : 0) * SS];
        dbv[g][r] = (ok && h < H) ? v : 0.f;
      }
  };
  int tile = (int)blockIdx.x * 4 + w;
  int cb = tile / tiles_per_b, ct = tile - cb * tiles_per_b;           // this tile
  int nb_ = cb + adv_b, nt_ = ct + adv_t;                              // the next one of this wave
  if (nt_ >= tiles_per_b) { nt_ -= tiles_per_b; ++nb_; }
  if (p.toff) { rf_locate(p, tile, ntiles, cb, ct); rf_locate(p, tile + nwaves, ntiles, nb_, nt_); }
  tile_load(cb, ct, ext, db);
  for (; tile < ntiles; tile += nwaves) {
    float ext_n[RF_CP], db_n[NG][4];
    tile_load(nb_, nt_, ext_n, db_n);   // in flight during this tile's MFMA chain
    nb_ += adv_b; nt_ += adv_t;
    if (nt_ >= tiles_per_b) { nt_ -= tiles_per_b; ++nb_; }
    if (p.toff) rf_locate(p, tile + 2 * nwaves, ntiles, nb_, nt_);
    // 1. hidden layer (transposed: rows j, columns e)
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
    // 2. r[h, e]
    f32x16 rr;
#pragma unroll
    for (int r = 0; r < 16; ++r) rr[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) rr = mfma32(sWrA[(16 * t + r) * 64 + lane], hid[t][r], rr);
    // 3. d(log max(r, 1e-6)) / dr
    float dpre[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float rv = rr[4 * g + r] + brv[g][r];
        dpre[g][r] = rv >= 1e-6f ? db[g][r] / rv : 0.f;
        accbr[g][r] += dpre[g][r];
      }
    // 4. gradient of the hidden layer, gated by relu'
    f32x16 dh[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[t][r] = 0.f;
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[t] = mfma32(wrT[t][g][r], dpre[g][r], dh[t]);
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[t][r] = hid[t][r] > 0.f ? dh[t][r] : 0.f;
    }
    // 5. transposes through the wave's LDS image.  LDS operations of one wave execute in order; the waits only
    //    keep the compiler from moving a read above the write it depends on.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        *reinterpret_cast<float4*>(sHid + l31 * RF_LDH + 32 * t + 8 * u + 4 * hh) =
            make_float4(hid[t][4 * u], hid[t][4 * u + 1], hid[t][4 * u + 2], hid[t][4 * u + 3]);
#pragma unroll
    for (int g = 0; g < NG; ++g)
      *reinterpret_cast<float4*>(sDpre + l31 * DP + 8 * g + 4 * hh) = make_float4(dpre[g][0], dpre[g][1], dpre[g][2], dpre[g][3]);
    if (hh == 0) {
      *reinterpret_cast<float4*>(sRaw + l31 * RF_CP) = make_float4(ext[0], ext[1], ext[2], ext[3]);
      *reinterpret_cast<float4*>(sRaw + l31 * RF_CP + 4) = make_float4(ext[4], ext[5], ext[6], ext[7]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // 5a. dWr[h, j] += sum_e dpre[h, e] hid[j, e]
    if constexpr (T16) {
      const int hcl = l15 < DP ? l15 : 0;
#pragma unroll
      for (int s0 = 0; s0 < 8; s0 += 2) {
        float av[2], bv[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = 4 * (s0 + u) + q4;
          av[u] = sDpre[e * DP + hcl];
#pragma unroll
          for (int n = 0; n < 4; ++n) bv[u][n] = sHid[e * RF_LDH + 16 * n + l15];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float a = l15 < DP ? av[u] : 0.f;
#pragma unroll
          for (int n = 0; n < 4; ++n) accWr16[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[u][n], accWr16[n], 0, 0, 0);
        }
      }
    } else {
    const int hcl = l31 < DP ? l31 : 0;
#pragma unroll
    for (int s0 = 0; s0 < 16; s0 += 4) {
      float av[4], b0[4], b1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = 2 * (s0 + u) + hh;
        av[u] = sDpre[e * DP + hcl];
        b0[u] = sHid[e * RF_LDH + l31];
        b1[u] = sHid[e * RF_LDH + 32 + l31];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float a = l31 < DP ? av[u] : 0.f;
        accWr[0] = mfma32(a, b0[u], accWr[0]);
        accWr[T16 ? 0 : 1] = mfma32(a, b1[u], accWr[T16 ? 0 : 1]);
      }
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        *reinterpret_cast<float4*>(sHid + l31 * RF_LDH + 32 * t + 8 * u + 4 * hh) =
            make_float4(dh[t][4 * u], dh[t][4 * u + 1], dh[t][4 * u + 2], dh[t][4 * u + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // 5b. dWy_ext[j, c] += sum_e dhid[j, e] raw_ext[e, c]   (column C of raw_ext is 1: dby)
    if constexpr (T16) {
      const int ccl = l15 < RF_CP ? l15 : 0;
#pragma unroll
      for (int s0 = 0; s0 < 8; s0 += 2) {
        float av[2][4], bv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = 4 * (s0 + u) + q4;
          bv[u] = sRaw[e * RF_CP + ccl];
#pragma unroll
          for (int m = 0; m < 4; ++m) av[u][m] = sHid[e * RF_LDH + 16 * m + l15];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float bb = l15 < RF_CP ? bv[u] : 0.f;
#pragma unroll
          for (int m = 0; m < 4; ++m) accWy16[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][m], bb, accWy16[m], 0, 0, 0);
        }
      }
    } else {
    const int ccl = l31 < RF_CP ? l31 : 0;
#pragma unroll
    for (int s0 = 0; s0 < 16; s0 += 4) {
      float a0[4], a1[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = 2 * (s0 + u) + hh;
        a0[u] = sHid[e * RF_LDH + l31];
        a1[u] = sHid[e * RF_LDH + 32 + l31];
        bv[u] = sRaw[e * RF_CP + ccl];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float bb = l31 < RF_CP ? bv[u] : 0.f;
        accWy[0] = mfma32(a0[u], bb, accWy[0]);
        accWy[T16 ? 0 : 1] = mfma32(a1[u], bb, accWy[T16 ? 0 : 1]);
      }
    }
    }
#pragma unroll
    for (int c = 0; c < RF_CP; ++c) ext[c] = ext_n[c];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) db[g][r] = db_n[g][r];
  }

  // ---- the workgroup's partial row: the 4 waves add their accumulators in wave order through LDS (fixed order:
  //      reproducible), the last one writes the row ----
  __syncthreads();                    // every wave is done with its LDS image
  float* srow = &sHidAll[0][0];       // 4 * 32 * RF_LDH floats >= RF_ROW
  static_assert(4 * 32 * RF_LDH >= RF_ROW, "partial row does not fit the hid images");
  float* grow = p.part + (size_t)blockIdx.x * RF_ROW;
  float brsum[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = accbr[g][r];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      brsum[g][r] = v;
    }
  for (int turn = 0; turn < 4; ++turn) {
    if (w == turn) {
      const bool first = turn == 0, last = turn == 3;
      if constexpr (T16) {   // (rows h >= 16 of the partial row stay unwritten: the reduction only uses h < H <= 8)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o1 = (4 * q4 + r) * RF_R + 16 * n + l15;                  // dWr[h][j]
            const float v1 = accWr16[n][r] + (first ? 0.f : srow[o1]);
            if (last) grow[o1] = v1; else srow[o1] = v1;
            if (l15 < RF_CP) {
              const int o2 = RF_HP * RF_R + (16 * n + 4 * q4 + r) * RF_CP + l15;  // dWy_ext[j][c]
              const float v2 = accWy16[n][r] + (first ? 0.f : srow[o2]);
              if (last) grow[o2] = v2; else srow[o2] = v2;
            }
          }
      } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = acc_row(r, hh);
          const int o1 = i * RF_R + 32 * t + l31;                               // dWr[h = i][j]
          const float v1 = accWr[T16 ? 0 : t][r] + (first ? 0.f : srow[o1]);
          if (last) grow[o1] = v1; else srow[o1] = v1;
          if (l31 < RF_CP) {
            const int o2 = RF_HP * RF_R + (32 * t + i) * RF_CP + l31;           // dWy_ext[j][c]
            const float v2 = accWy[T16 ? 0 : t][r] + (first ? 0.f : srow[o2]);
            if (last) grow[o2] = v2; else srow[o2] = v2;
          }
        }
      }
      // dbr: lane (l31 == 0, half) carries heads 8g + 4*half + r; the other slots of the 32-head block are zero
      if (hh == 0) {
        const int o3 = RF_HP * RF_R + RF_R * RF_CP + l31;
        const float v3 = first ? 0.f : srow[o3];
        if (last) grow[o3] = v3; else srow[o3] = v3;
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (l31 == 0) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int o3 = RF_HP * RF_R + RF_R * RF_CP + 8 * g + r + 4 * hh;
            const float v3 = brsum[g][r] + (first ? 0.f : srow[o3]);
            if (last) grow[o3] = v3; else srow[o3] = v3;
          }
      }
    }
    __syncthreads();
  }
}

// Backward for H <= 4 heads (HSIZE 256: the supernet of search_*.py), round 3.  The MFMA chain
// above multiplies mostly padding once the head count is small: the head projection (step 2) and the two parameter-
// gradient contractions (step 5) put 4-8 heads / 5 raw channels on 16- or 32-wide tiles -- 4352 of its 5120
// MFMA-cycles per 32 elements.  The fp32 vector pipe has the same peak as the fp32 MFMA and no padding, so here:
//   1. hid^T[j,e]  on the MFMA as before (8 MFMAs; A = Wy|by, B = the lane's raw row);
//   2. r[h,e]      on the VALU: lane (e, half) holds 32 of the 64 hidden values of its element; HH partial dot products
//                  with Wr^T read as broadcast float4 from LDS, one cross-half shuffle each;
//   3. dpre[h,e]   every lane of an element holds all HH heads;
//   4. dhid^T[j,e] on the MFMA with the heads as the k index (HH MFMAs instead of 8-16);
//   5. dWr, dWy    on the VALU with lane = hidden unit j: hid / dhid pass through the wave's LDS image (written [e][j],
//                  read back row by row: conflict-free), dpre[e,:] and raw_ext[e,:] are broadcast float4 reads;
//                  4 + 6 (HH = 4) fused multiply-adds per element and lane, HH + C + 1 accumulators per lane.
// Per 32 elements: 768 MFMA-cycles + ~2400 VALU-cycles at two waves per SIMD (52 KB LDS), against 5120 MFMA-cycles.
// Measured (tools/rel_bench.py, B = 64, 100 x 100, 4 heads): 91 -> 76 us per backward (the reduction launch included).
template <int C, int HH>
__global__ void __launch_bounds__(256, 2)
rel_fused_bwd_v_kernel(const RelFusedK p, int ntiles, int tiles_per_b, const float* __restrict__ Wy,
                       const float* __restrict__ by, const float* __restrict__ Wr, const float* __restrict__ br) {
  static_assert(HH == 4 || HH == 8, "heads padded to 4 or 8");
  __shared__ __attribute__((aligned(16))) float sHidAll[4][32 * RF_LDH];
  __shared__ __attribute__((aligned(16))) float sDpreAll[4][32 * 8];
  __shared__ __attribute__((aligned(16))) float sRawAll[4][32 * RF_CP];
  __shared__ __attribute__((aligned(16))) float sWrT[RF_R * 8];      // Wr^T: [j][h], heads padded to 8
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  float* sHid = sHidAll[w];
  float* sDpre = sDpreAll[w];
  float* sRaw = sRawAll[w];
  const int H = p.H;
  for (int i = tid; i < RF_R * 8; i += 256) {
    const int j = i >> 3, h = i & 7;
    sWrT[i] = h < H ? Wr[h * RF_R + j] : 0.f;
  }
  // ---- per-lane constant MFMA operands ----
  float wyA[2][4];       // step 1, A[i = j][k = c]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int j = 32 * t + l31, c = 2 * s4 + hh;
      wyA[t][s4] = c < C ? Wy[j * C + c] : (c == C ? by[j] : 0.f);
    }
  float wrK[2][HH / 2];  // step 4, A[i = j][k = h]: MFMA m covers heads 2m (lower half-wave) and 2m + 1 (upper)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int m = 0; m < HH / 2; ++m) {
      const int h = 2 * m + hh;
      wrK[t][m] = h < H ? Wr[h * RF_R + 32 * t + l31] : 0.f;
    }
  float brv[HH];
#pragma unroll
  for (int h = 0; h < HH; ++h) brv[h] = h < H ? br[h] : 0.f;
  __syncthreads();

  float accWr[HH], accWy[RF_CP], accbr[HH];   // lane = hidden unit j (step 5); accbr: lane = element
#pragma unroll
  for (int h = 0; h < HH; ++h) { accWr[h] = 0.f; accbr[h] = 0.f; }
#pragma unroll
  for (int c = 0; c < RF_CP; ++c) accWy[c] = 0.f;

  const unsigned SS = (unsigned)p.Sq * (unsigned)p.Sk;
  const int nwaves = (int)gridDim.x * 4;
  const int adv_b = nwaves / tiles_per_b, adv_t = nwaves - adv_b * tiles_per_b;
  struct TileIn { float ex[RF_CP]; float db[HH]; };
  // loads of one tile: the lane's raw row (gathered, 16 B) and its element's dbias of every head; issued one tile ahead
  auto tile_load = [&](int b, int tb) __attribute__((always_inline)) {
    TileIn t;
    const unsigned f = (unsigned)tb * 32u + (unsigned)l31;
    bool ok = b < p.B && f < SS;
    unsigned wq = (unsigned)p.Sq;            // row length of the flattened (key, query) index f
    if (p.off) {                             // ragged: f runs over the n x n valid elements of the sample
      const int bb = b < p.B ? b : 0;
      const unsigned n = (unsigned)max(p.off[bb + 1] - p.off[bb], 1);
      ok = b < p.B && f < n * n && p.off[bb + 1] > p.off[bb];
      wq = n;
    }
    const unsigned fc0 = ok ? f : 0u;
    const int bc = ok ? b : 0;
    const unsigned k = fc0 / wq, q = fc0 - k * wq;
    const unsigned fc = k * (unsigned)p.Sq + q;   // position in the padded [Sk, Sq] plane of dbiasT
    const float* src = p.raw + (((size_t)bc * p.Sq + q) * p.Sk + k) * C;
#pragma unroll
    for (int c = 0; c < RF_CP; ++c) t.ex[c] = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { const float v = src[c]; t.ex[c] = ok ? v : 0.f; }   // (address clamped: the load is unconditional)
    t.ex[C] = ok ? 1.f : 0.f;
    const float* dbp = p.dbiasT + (size_t)bc * H * SS + fc;
#pragma unroll
    for (int h = 0; h < HH; ++h) {
      const float v = dbp[(size_t)(h < H ? h : 0) * SS];
      t.db[h] = (ok && h < H) ? v : 0.f;
    }
    return t;
  };
  int tile = (int)blockIdx.x * 4 + w;
  int cb = tile / tiles_per_b, ct = tile - cb * tiles_per_b;
  int nb_ = cb + adv_b, nt_ = ct + adv_t;
  if (nt_ >= tiles_per_b) { nt_ -= tiles_per_b; ++nb_; }
  if (p.toff) { rf_locate(p, tile, ntiles, cb, ct); rf_locate(p, tile + nwaves, ntiles, nb_, nt_); }
  TileIn cur = tile_load(cb, ct);
  for (; tile < ntiles; tile += nwaves) {
    const TileIn nxt = tile_load(nb_, nt_);   // in flight during this tile's arithmetic
    nb_ += adv_b; nt_ += adv_t;
    if (nt_ >= tiles_per_b) { nt_ -= tiles_per_b; ++nb_; }
    if (p.toff) rf_locate(p, tile + 2 * nwaves, ntiles, nb_, nt_);
    // 1. hidden layer (transposed: rows j, columns e); relu, and its gate as one bit per accumulator register
    f32x16 hid[2];
    unsigned gm = 0u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) hid[t] = mfma32(wyA[t][s4], hh ? cur.ex[2 * s4 + 1] : cur.ex[2 * s4], hid[t]);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        gm |= hid[t][r] > 0.f ? (1u << (16 * t + r)) : 0u;
        hid[t][r] = fmaxf(hid[t][r], 0.f);
      }
    }
    // 2. r[h, e]: this lane's 32 hidden values against the matching rows of Wr^T (broadcast reads: the 16-lane read
    //    groups of ds_read_b128 sit inside one half-wave, i.e. on one address)
    //    (The Wr^T rows are loop invariant: for 4 heads the compiler keeps all 128 of a lane's values in registers, so the
    //    tile loop reads no weights at all -- at the price of two waves per SIMD.  Tried and dropped: volatile reads +
    //    three waves per SIMD (168 VGPRs: >200 spill instructions in the loop); two-wide vector types so the products
    //    issue as v_pk_fma_f32 (97 packed instead of ~190 scalar FMAs: 79.6 vs 76.2 us -- the chain of dependent MFMA /
    //    shuffle / LDS round trips, not the issue rate, sets the pace).)
    float rr[HH];
#pragma unroll
    for (int h = 0; h < HH; ++h) rr[h] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float* wp = sWrT + (32 * t + acc_row(r, hh)) * 8;
        const float4 w0 = *reinterpret_cast<const float4*>(wp);
        rr[0] += w0.x * hid[t][r]; rr[1] += w0.y * hid[t][r]; rr[2] += w0.z * hid[t][r]; rr[3] += w0.w * hid[t][r];
        if (HH == 8) {
          const float4 w1 = *reinterpret_cast<const float4*>(wp + 4);
          rr[4] += w1.x * hid[t][r]; rr[5] += w1.y * hid[t][r]; rr[6] += w1.z * hid[t][r]; rr[7] += w1.w * hid[t][r];
        }
      }
    // 3. d(log max(r, 1e-6)) / dr -- both half-waves hold every head of their element
    float dpre[HH];
#pragma unroll
    for (int h = 0; h < HH; ++h) {
      const float rv = rr[h] + __shfl_xor(rr[h], 32, 64) + brv[h];
      dpre[h] = rv >= 1e-6f ? cur.db[h] / rv : 0.f;
      accbr[h] += dpre[h];
    }
    // 5a. through the wave's LDS image: hid written [e][j] from the accumulator layout, read back with lane = j
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        *reinterpret_cast<float4*>(sHid + l31 * RF_LDH + 32 * t + 8 * u + 4 * hh) =
            make_float4(hid[t][4 * u], hid[t][4 * u + 1], hid[t][4 * u + 2], hid[t][4 * u + 3]);
    if (hh == 0) {
      *reinterpret_cast<float4*>(sDpre + l31 * 8) = make_float4(dpre[0], dpre[1], dpre[2], dpre[3]);
      if (HH == 8) *reinterpret_cast<float4*>(sDpre + l31 * 8 + 4) = make_float4(dpre[4], dpre[5], dpre[6], dpre[7]);
      *reinterpret_cast<float4*>(sRaw + l31 * RF_CP) = make_float4(cur.ex[0], cur.ex[1], cur.ex[2], cur.ex[3]);
      *reinterpret_cast<float4*>(sRaw + l31 * RF_CP + 4) = make_float4(cur.ex[4], cur.ex[5], cur.ex[6], cur.ex[7]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);       // (the hidden values are dead from here on: 32 registers)
    //     dWr[h, j = lane] += sum_e dpre[h, e] hid[j, e]
#pragma unroll 4
    for (int e = 0; e < 32; ++e) {
      const float hv = sHid[e * RF_LDH + lane];
      const float4 d0 = *reinterpret_cast<const float4*>(sDpre + e * 8);
      accWr[0] += d0.x * hv; accWr[1] += d0.y * hv; accWr[2] += d0.z * hv; accWr[3] += d0.w * hv;
      if (HH == 8) {
        const float4 d1 = *reinterpret_cast<const float4*>(sDpre + e * 8 + 4);
        accWr[4] += d1.x * hv; accWr[5] += d1.y * hv; accWr[6] += d1.z * hv; accWr[7] += d1.w * hv;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // 4. gradient of the hidden layer, gated by relu' (the saved bits)
    f32x16 dh[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[t][r] = 0.f;
#pragma unroll
      for (int m = 0; m < HH / 2; ++m) dh[t] = mfma32(wrK[t][m], hh ? dpre[2 * m + 1] : dpre[2 * m], dh[t]);
#pragma unroll
      for (int r = 0; r < 16; ++r) dh[t][r] = (gm >> (16 * t + r)) & 1u ? dh[t][r] : 0.f;
    }
    // 5b. the same image again, now with dhid
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u)
        *reinterpret_cast<float4*>(sHid + l31 * RF_LDH + 32 * t + 8 * u + 4 * hh) =
            make_float4(dh[t][4 * u], dh[t][4 * u + 1], dh[t][4 * u + 2], dh[t][4 * u + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    //     dWy_ext[j = lane, c] += sum_e dhid[j, e] raw_ext[e, c]   (column C of raw_ext is 1: dby)
#pragma unroll 4
    for (int e = 0; e < 32; ++e) {
      const float dv = sHid[e * RF_LDH + lane];
      const float4 r0 = *reinterpret_cast<const float4*>(sRaw + e * RF_CP);
      const float4 r1 = *reinterpret_cast<const float4*>(sRaw + e * RF_CP + 4);
      accWy[0] += dv * r0.x; accWy[1] += dv * r0.y; accWy[2] += dv * r0.z; accWy[3] += dv * r0.w;
      accWy[4] += dv * r1.x;
      if (C + 1 > 5) accWy[5] += dv * r1.y;
    }
    cur = nxt;
  }

  // ---- the workgroup's partial row: the 4 waves add their accumulators in wave order through LDS (fixed order:
  //      reproducible), the last one writes the row.  Row layout as for the MFMA kernel: [dWr: HP x R | dWy_ext: R x CP | dbr: HP]
  __syncthreads();
  float* srow = &sHidAll[0][0];
  float* grow = p.part + (size_t)blockIdx.x * RF_ROW;
  float brsum[HH];
#pragma unroll
  for (int h = 0; h < HH; ++h) {
    float v = hh == 0 ? accbr[h] : 0.f;     // (both half-waves carried the same dpre: count one)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    brsum[h] = v;
  }
  for (int turn = 0; turn < 4; ++turn) {
    if (w == turn) {
      const bool first = turn == 0, last = turn == 3;
#pragma unroll
      for (int h = 0; h < HH; ++h) {
        const int o1 = h * RF_R + lane;                                  // dWr[h][j]
        const float v1 = accWr[h] + (first ? 0.f : srow[o1]);
        if (last) grow[o1] = v1; else srow[o1] = v1;
      }
#pragma unroll
      for (int c = 0; c < RF_CP; ++c) {
        const int o2 = RF_HP * RF_R + lane * RF_CP + c;                  // dWy_ext[j][c]
        const float v2 = (c <= C ? accWy[c] : 0.f) + (first ? 0.f : srow[o2]);
        if (last) grow[o2] = v2; else srow[o2] = v2;
      }
      if (lane < RF_HP) {
        const int o3 = RF_HP * RF_R + RF_R * RF_CP + lane;               // dbr[h]
        float mine = 0.f;
#pragma unroll
        for (int h = 0; h < HH; ++h) mine = lane == h ? brsum[h] : mine;
        const float v3 = mine + (first ? 0.f : srow[o3]);
        if (last) grow[o3] = v3; else srow[o3] = v3;
      }
    }
    __syncthreads();
  }
}

// sum the partial rows and add into the parameter gradients (single writer per output: plain +=)
__global__ void __launch_bounds__(1024) rel_fused_reduce_kernel(const float* __restrict__ part, int nrows, int C, int H,
                                                                float* dWr, float* dbr, float* dWy, float* dby) {
  // 16 row groups of 64 columns per workgroup (the 4-group form walked ~130 dependent loads per thread: 12.7 us for 20 KB)
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  // a block of 64 columns inside the dWr region is one head's row: heads >= H are padding (and, in the 16-wide form
  // of the backward kernel, not even written) -- 75 % of the partial rows for 8 heads
  if (blockIdx.x < RF_HP && (int)blockIdx.x >= H) return;
  float s = 0.f;
  if (col < RF_ROW) {
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = g;
    for (; r + 48 < nrows; r += 64) {
      s += part[(size_t)r * RF_ROW + col];
      s1 += part[(size_t)(r + 16) * RF_ROW + col];
      s2 += part[(size_t)(r + 32) * RF_ROW + col];
      s3 += part[(size_t)(r + 48) * RF_ROW + col];
    }
    for (; r < nrows; r += 16) s += part[(size_t)r * RF_ROW + col];
    s += (s1 + s2) + s3;
  }
  __shared__ float red[16][64];
  red[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g != 0 || col >= RF_ROW) return;
  const int cl = threadIdx.x & 63;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) v += red[i][cl];
  if (col < RF_HP * RF_R) {
    const int h = col / RF_R, j = col - h * RF_R;
    if (h < H) dWr[h * RF_R + j] += v;
  } else if (col < RF_HP * RF_R + RF_R * RF_CP) {
    const int o = col - RF_HP * RF_R, j = o / RF_CP, c = o - j * RF_CP;
    if (c < C) dWy[j * C + c] += v;
    else if (c == C) dby[j] += v;
  } else {
    const int h = col - RF_HP * RF_R - RF_R * RF_CP;
    if (h < H) dbr[h] += v;
  }
}

static int rf_check(const char* who, int B, int Sq, int Sk, int C, int R, int H) {
  MMNAS_REQUIRE(B > 0 && Sq > 0 && Sk > 0, MMNAS_E_SHAPE, "%s: B=%d Sq=%d Sk=%d", who, B, Sq, Sk);
  MMNAS_REQUIRE(R == RF_R, MMNAS_E_SHAPE, "%s: REL_SIZE=%d (the fused path handles 64)", who, R);
  MMNAS_REQUIRE(C == 3 || C == 4, MMNAS_E_SHAPE, "%s: %d raw relation channels (3 or 4)", who, C);
  MMNAS_REQUIRE(H >= 1 && H <= RF_HP, MMNAS_E_SHAPE, "%s: H=%d heads (1..32)", who, H);
  return MMNAS_OK;
}

static long rf_tiles_per_b(int Sq, int Sk) { return ((long)Sq * Sk + 31) / 32; }
static int rf_grid(long ntiles, int per_cu = 2) {   // persistent: 2 or 3 workgroups per CU, one 32-element tile per wave at a time
  const long wgs = (ntiles + 3) / 4, cap = 256l * per_cu;
  return (int)(wgs < cap ? wgs : cap);
}

}  // namespace mmnas

using namespace mmnas;

extern "C" int mmnas_rel_fused_supported(int C, int R, int H) { return R == RF_R && (C == 3 || C == 4) && H >= 1 && H <= RF_HP; }

static int rf_fwd_impl(const float* raw, const float* Wy, const float* by, const float* Wr, const float* br,
                       float* biasT, int B, int Sq, int Sk, int C, int R, int H, const int* off, void* stream) {
  MMNAS_REQUIRE(raw && Wy && by && Wr && br && biasT, MMNAS_E_ARG, "rel_fused_fwd: null pointer");
  int rc = rf_check("rel_fused_fwd", B, Sq, Sk, C, R, H);
  if (rc) return rc;
  RelFusedK k;
  k.raw = raw; k.Wy = Wy; k.by = by; k.Wr = Wr; k.br = br; k.biasT = biasT; k.dbiasT = nullptr; k.part = nullptr;
  k.B = B; k.Sq = Sq; k.Sk = Sk; k.C = C; k.H = H; k.nbq = cdiv(Sq, 64); k.nbk = cdiv(Sk, 4);
  k.off = off; k.toff = nullptr;
  if (off) MMNAS_REQUIRE(Sq == Sk, MMNAS_E_SHAPE, "rel_fused_fwd: ragged batches are self-attention (Sq == Sk)");
  const long nbatch = (long)B * k.nbq * k.nbk;
  hipStream_t st = (hipStream_t)stream;
  const double n = (double)B * Sq * Sk;
  ProfScope ps(MMNAS_K_REL_FWD, 2.0 * n * (RF_R * (C + 1) + (double)H * RF_R), 4.0 * n * (C + H), st);
  if (C == 4) MMNAS_LAUNCH(rel_fused_fwd_kernel<4>, dim3((unsigned)nbatch), dim3(256), 0, st, k, Wy, by, Wr, br);
  else MMNAS_LAUNCH(rel_fused_fwd_kernel<3>, dim3((unsigned)nbatch), dim3(256), 0, st, k, Wy, by, Wr, br);
  return check_launch("rel_fused_fwd");
}

extern "C" int mmnas_rel_fused_fwd(const float* raw, const float* Wy, const float* by, const float* Wr, const float* br,
                                   float* biasT, int B, int Sq, int Sk, int C, int R, int H, void* stream) {
  return rf_fwd_impl(raw, Wy, by, Wr, br, biasT, B, Sq, Sk, C, R, H, nullptr, stream);
}
extern "C" int mmnas_rel_fused_fwd_ragged(const float* raw, const float* Wy, const float* by, const float* Wr, const float* br,
                                          float* biasT, int B, int S, int C, int R, int H, const int* off, void* stream) {
  MMNAS_REQUIRE(off, MMNAS_E_ARG, "rel_fused_fwd_ragged: null offsets");
  return rf_fwd_impl(raw, Wy, by, Wr, br, biasT, B, S, S, C, R, H, off, stream);
}

extern "C" size_t mmnas_rel_fused_bwd_ws_floats(int B, int Sq, int Sk) {
  return (size_t)rf_grid((long)B * rf_tiles_per_b(Sq, Sk), 2) * RF_ROW;
}

static int rf_bwd_impl(const float* raw, const float* Wy, const float* by, const float* Wr, const float* br,
                       const float* dbiasT, float* dWy, float* dby, float* dWr, float* dbr, float* ws,
                       int B, int Sq, int Sk, int C, int R, int H, const int* off, const int* toff, int ntiles_ragged, void* stream) {
  MMNAS_REQUIRE(raw && Wy && by && Wr && br && dbiasT && dWy && dby && dWr && dbr && ws, MMNAS_E_ARG,
                "rel_fused_bwd: null pointer");
  int rc = rf_check("rel_fused_bwd", B, Sq, Sk, C, R, H);
  if (rc) return rc;
  RelFusedK k;
  k.raw = raw; k.Wy = Wy; k.by = by; k.Wr = Wr; k.br = br; k.biasT = nullptr; k.dbiasT = dbiasT; k.part = ws;
  k.B = B; k.Sq = Sq; k.Sk = Sk; k.C = C; k.H = H; k.nbq = cdiv(Sq, 64); k.nbk = cdiv(Sk, 4);
  const int tpb = (int)rf_tiles_per_b(Sq, Sk);
  MMNAS_REQUIRE((long)B * tpb < (1l << 30) && (long)Sq * Sk < (1l << 30), MMNAS_E_SHAPE, "rel_fused_bwd: problem too large for 32-bit tile indices");
  k.off = off; k.toff = toff;
  if (off) MMNAS_REQUIRE(toff && Sq == Sk && ntiles_ragged >= 0 && ntiles_ragged <= B * tpb, MMNAS_E_ARG,
                         "rel_fused_bwd: ragged batches need tile offsets, Sq == Sk and a tile count <= the dense one (%d vs %d)", ntiles_ragged, B * tpb);
  const int ntiles = off ? ntiles_ragged : B * tpb;
  static const bool vpath = !(getenv("MMNAS_REL_BWD_VALU") && getenv("MMNAS_REL_BWD_VALU")[0] == '0');   // 0: the all-MFMA kernel (A/B runs)
  const bool use_v = H <= 4 && vpath;
  const int grid = rf_grid(ntiles > 0 ? ntiles : 1, 2);
  hipStream_t st = (hipStream_t)stream;
  const double n = (double)B * Sq * Sk;
  ProfScope ps(MMNAS_K_REL_BWD, 2.0 * n * (RF_R * (C + 1) + 3.0 * H * RF_R + RF_R * (C + 1)), 4.0 * n * (C + H), st);
#define RF_BWD(CC, NGG) MMNAS_LAUNCH((rel_fused_bwd_kernel<CC, NGG>), dim3(grid), dim3(256), 0, st, k, ntiles, tpb, Wy, by, Wr, br)
#define RF_BWDV(CC, HHH) MMNAS_LAUNCH((rel_fused_bwd_v_kernel<CC, HHH>), dim3(grid), dim3(256), 0, st, k, ntiles, tpb, Wy, by, Wr, br)
  // H <= 4 (HSIZE 256: the supernet): the vector-pipe kernel.  At 8 heads its 256 loop-invariant Wr^T values no longer
  // stay in registers beside the tile state (the compiler keeps them there for 4 heads: no weight reads per tile at all)
  // and the head projection costs the vector pipe what the 16-wide MFMA form costs the matrix pipe: the MFMA kernel stays.
  if (use_v) { if (C == 4) RF_BWDV(4, 4); else RF_BWDV(3, 4); }
  else if (H <= 8) { if (C == 4) RF_BWD(4, 1); else RF_BWD(3, 1); }
  else if (H <= 16) { if (C == 4) RF_BWD(4, 2); else RF_BWD(3, 2); }
  else { if (C == 4) RF_BWD(4, 4); else RF_BWD(3, 4); }
#undef RF_BWD
#undef RF_BWDV
  MMNAS_LAUNCH(rel_fused_reduce_kernel, dim3(cdiv(RF_ROW, 64)), dim3(1024), 0, st, ws, grid, C, H, dWr, dbr, dWy, dby);
  return check_launch("rel_fused_bwd");
}

extern "C" int mmnas_rel_fused_bwd(const float* raw, const float* Wy, const float* by, const float* Wr, const float* br,
                                   const float* dbiasT, float* dWy, float* dby, float* dWr, float* dbr, float* ws,
                                   int B, int Sq, int Sk, int C, int R, int H, void* stream) {
  return rf_bwd_impl(raw, Wy, by, Wr, br, dbiasT, dWy, dby, dWr, dbr, ws, B, Sq, Sk, C, R, H, nullptr, nullptr, 0, stream);
}
extern "C" int mmnas_rel_fused_bwd_ragged(const float* raw, const float* Wy, const float* by, const float* Wr, const float* br,
                                          const float* dbiasT, float* dWy, float* dby, float* dWr, float* dbr, float* ws,
                                          int B, int S, int C, int R, int H, const int* off, const int* tile_off, int ntiles, void* stream) {
  MMNAS_REQUIRE(off && tile_off, MMNAS_E_ARG, "rel_fused_bwd_ragged: null offsets");
  return rf_bwd_impl(raw, Wy, by, Wr, br, dbiasT, dWy, dby, dWr, dbr, ws, B, S, S, C, R, H, off, tile_off, ntiles, stream);
}

// Relation bias of RelMHAtt (modules.py:231-235):
//   r[b,q,k,h] = relu(rel[b,q,k,:] . Wr[h,:] + br[h]);  bias = log(max(r, 1e-6))
// rel is [B,Sq,Sk,R] fp32 -- 164 MB at B=64, S=100, R=64: the one HBM-bound tensor of the path
// (4 flop/byte).  Both kernels stream it exactly once with fully coalesced 16-byte loads: a row of
// R floats is spread over R/4 consecutive lanes (16 for R=64, 4 rows per wave instruction); every
// lane forms its 4-element partial dot with all H heads, and a xor-butterfly inside the R/4-lane
// group completes the dots.  The bias is written key-major ([B,H,Sk,Sq]) for the attention core.
// Backward re-reads rel once, writes d_rel once (coalesced float4), and reduces dWr/dbr in
// registers -> LDS -> one atomic per element per workgroup.
#include "common.h"

namespace mmnas {

template <int LPR>  // lanes per row = R/4
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LPR, int HMAX>
__global__ void __launch_bounds__(256) rel_bias_fwd_kernel(const float* __restrict__ rel, const float* __restrict__ Wr,
                                                           const float* __restrict__ br, float* __restrict__ biasT,
                                                           int B, int Sq, int Sk, int H) {
  constexpr int R = LPR * 4, RPW = 64 / LPR;  // rows per wave instruction
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPR, rin = lane / LPR;
  float4 w[HMAX];
#pragma unroll
  for (int h = 0; h < HMAX; ++h)
    w[h] = h < H ? *reinterpret_cast<const float4*>(Wr + h * R + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
  const long nrows = (long)B * Sq * Sk;
  const long wave_id = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  for (long row0 = wave_id * RPW; row0 < nrows; row0 += nwaves * RPW) {
    const long row = row0 + rin;
    const bool ok = row < nrows;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) x = *reinterpret_cast<const float4*>(rel + row * R + 4 * sub);
    float mine = 0.f;  // lane `sub` keeps head `sub` (H <= LPR) -- else written by sub == 0
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
      if (h < H) {
        float d = (x.x * w[h].x + x.y * w[h].y) + (x.z * w[h].z + x.w * w[h].w);
        d = group_sum<LPR>(d);
        if (HMAX <= LPR) { if (sub == h) mine = d; }
        else if (ok && sub == 0) {
          const long bq = row / Sk; const int k = (int)(row - bq * Sk);
          const int b = (int)(bq / Sq), q = (int)(bq - (long)b * Sq);
          const float r = fmaxf(d + br[h], 0.f);
          biasT[(((size_t)b * H + h) * Sk + k) * Sq + q] = logf(fmaxf(r, 1e-6f));
        }
      }
    }
    if (HMAX <= LPR && ok && sub < H) {
      const long bq = row / Sk; const int k = (int)(row - bq * Sk);
      const int b = (int)(bq / Sq), q = (int)(bq - (long)b * Sq);
      const float r = fmaxf(mine + br[sub], 0.f);
      biasT[(((size_t)b * H + sub) * Sk + k) * Sq + q] = logf(fmaxf(r, 1e-6f));
    }
  }
}

template <int LPR, int HMAX>
__global__ void __launch_bounds__(256) rel_bias_bwd_kernel(const float* __restrict__ rel, const float* __restrict__ Wr,
                                                           const float* __restrict__ br,
                                                           const float* __restrict__ dbiasT, float* __restrict__ drel,
                                                           float* __restrict__ dWr, float* __restrict__ dbr,
                                                           int accumulate, int B, int Sq, int Sk, int H) {
  constexpr int R = LPR * 4, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPR, rin = lane / LPR;
  float4 w[HMAX], gw[HMAX];
  float gb[HMAX];
#pragma unroll
  for (int h = 0; h < HMAX; ++h) {
    w[h] = h < H ? *reinterpret_cast<const float4*>(Wr + h * R + 4 * sub) : make_float4(0.f, 0.f, 0.f, 0.f);
    gw[h] = make_float4(0.f, 0.f, 0.f, 0.f);
    gb[h] = 0.f;
  }
  const long nrows = (long)B * Sq * Sk;
  const long wave_id = (long)blockIdx.x * 4 + wave;
  const long nwaves = (long)gridDim.x * 4;
  for (long row0 = wave_id * RPW; row0 < nrows; row0 += nwaves * RPW) {
    const long row = row0 + rin;
    const bool ok = row < nrows;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    int b = 0, q = 0, k = 0;
    if (ok) {
      x = *reinterpret_cast<const float4*>(rel + row * R + 4 * sub);
      const long bq = row / Sk; k = (int)(row - bq * Sk);
      b = (int)(bq / Sq); q = (int)(bq - (long)b * Sq);
    }
    float4 dx = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
      if (h < H) {
        float d = (x.x * w[h].x + x.y * w[h].y) + (x.z * w[h].z + x.w * w[h].w);
        d = group_sum<LPR>(d) + br[h];
        float dpre = 0.f;
        if (ok && d > 1e-6f) dpre = dbiasT[(((size_t)b * H + h) * Sk + k) * Sq + q] / d;
        dx.x += dpre * w[h].x; dx.y += dpre * w[h].y; dx.z += dpre * w[h].z; dx.w += dpre * w[h].w;
        gw[h].x += dpre * x.x; gw[h].y += dpre * x.y; gw[h].z += dpre * x.z; gw[h].w += dpre * x.w;
        if (sub == 0) gb[h] += dpre;
      }
    }
    if (ok && drel) {
      float4* o = reinterpret_cast<float4*>(drel + row * R + 4 * sub);
      if (accumulate) { const float4 old = *o; dx.x += old.x; dx.y += old.y; dx.z += old.z; dx.w += old.w; }
      *o = dx;
    }
  }
  // reduce gw over the RPW row slots of the wave, then over the 4 waves, then one atomic each
  __shared__ float red[4][HMAX][R];
  __shared__ float redb[4][HMAX];
#pragma unroll
  for (int h = 0; h < HMAX; ++h) {
    float4 g = gw[h];
    float gbb = gb[h];
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
      g.x += __shfl_xor(g.x, o, 64); g.y += __shfl_xor(g.y, o, 64);
      g.z += __shfl_xor(g.z, o, 64); g.w += __shfl_xor(g.w, o, 64);
      gbb += __shfl_xor(gbb, o, 64);
    }
    if (rin == 0) {
      red[wave][h][4 * sub] = g.x; red[wave][h][4 * sub + 1] = g.y;
      red[wave][h][4 * sub + 2] = g.z; red[wave][h][4 * sub + 3] = g.w;
      if (sub == 0) redb[wave][h] = gbb;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H * R; i += 256) {
    const int h = i / R, c = i - h * R;
    atomicAdd(dWr + i, (red[0][h][c] + red[1][h][c]) + (red[2][h][c] + red[3][h][c]));
  }
  if (threadIdx.x < H) {
    const int h = threadIdx.x;
    atomicAdd(dbr + h, (redb[0][h] + redb[1][h]) + (redb[2][h] + redb[3][h]));
  }
}

static int rel_check(const char* who, int B, int Sq, int Sk, int R, int H) {
  MMNAS_REQUIRE(B > 0 && Sq > 0 && Sk > 0, MMNAS_E_SHAPE, "%s: B=%d Sq=%d Sk=%d", who, B, Sq, Sk);
  MMNAS_REQUIRE(R == 16 || R == 32 || R == 64 || R == 128 || R == 256, MMNAS_E_SHAPE,
                "%s: REL_SIZE=%d not in {16,32,64,128,256}", who, R);
  MMNAS_REQUIRE(H >= 1 && H <= 32, MMNAS_E_SHAPE, "%s: H=%d heads (1..32)", who, H);
  return MMNAS_OK;
}

}  // namespace mmnas

using namespace mmnas;

#define REL_DISPATCH(KERNEL, ...)                                                                          \
  do {                                                                                                     \
    const int hm = H <= 4 ? 4 : (H <= 8 ? 8 : (H <= 16 ? 16 : 32));                                       \
    switch (R) {                                                                                           \
      case 16: if (hm == 4) KERNEL(4, 4, __VA_ARGS__); else if (hm == 8) KERNEL(4, 8, __VA_ARGS__);        \
               else if (hm == 16) KERNEL(4, 16, __VA_ARGS__); else KERNEL(4, 32, __VA_ARGS__); break;      \
      case 32: if (hm == 4) KERNEL(8, 4, __VA_ARGS__); else if (hm == 8) KERNEL(8, 8, __VA_ARGS__);        \
               else if (hm == 16) KERNEL(8, 16, __VA_ARGS__); else KERNEL(8, 32, __VA_ARGS__); break;      \
      case 64: if (hm == 4) KERNEL(16, 4, __VA_ARGS__); else if (hm == 8) KERNEL(16, 8, __VA_ARGS__);      \
               else if (hm == 16) KERNEL(16, 16, __VA_ARGS__); else KERNEL(16, 32, __VA_ARGS__); break;    \
      case 128: if (hm == 4) KERNEL(32, 4, __VA_ARGS__); else if (hm == 8) KERNEL(32, 8, __VA_ARGS__);     \
               else if (hm == 16) KERNEL(32, 16, __VA_ARGS__); else KERNEL(32, 32, __VA_ARGS__); break;    \
      default: if (hm == 4) KERNEL(64, 4, __VA_ARGS__); else if (hm == 8) KERNEL(64, 8, __VA_ARGS__);      \
               else if (hm == 16) KERNEL(64, 16, __VA_ARGS__); else KERNEL(64, 32, __VA_ARGS__); break;    \
    }                                                                                                      \
  } while (0)

extern "C" int mmnas_rel_bias_fwd(const float* rel, const float* Wr, const float* br, float* biasT, int B, int Sq,
                                  int Sk, int R, int H, void* stream) {
  MMNAS_REQUIRE(rel && Wr && br && biasT, MMNAS_E_ARG, "rel_bias_fwd: null pointer");
  int rc = rel_check("rel_bias_fwd", B, Sq, Sk, R, H);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const long nrows = (long)B * Sq * Sk;
  const long need = (nrows * (R / 4) + 255) / 256;
  const int blocks = (int)(need < 4096 ? (need ? need : 1) : 4096);
  // algorithmic traffic: rel read once + bias written once
  ProfScope ps(MMNAS_K_REL_FWD, 2.0 * nrows * R * H, 4.0 * ((double)nrows * R + (double)nrows * H), st);
#define FWDK(LPR, HM, ...) MMNAS_LAUNCH((rel_bias_fwd_kernel<LPR, HM>), dim3(blocks), dim3(256), 0, st, __VA_ARGS__)
  REL_DISPATCH(FWDK, rel, Wr, br, biasT, B, Sq, Sk, H);
#undef FWDK
  return check_launch("rel_bias_fwd");
}

extern "C" int mmnas_rel_bias_bwd(const float* rel, const float* Wr, const float* br, const float* dbiasT,
                                  float* drel, float* dWr, float* dbr, int accumulate_drel, int B, int Sq, int Sk,
                                  int R, int H, void* stream) {
  MMNAS_REQUIRE(rel && Wr && br && dbiasT && dWr && dbr, MMNAS_E_ARG, "rel_bias_bwd: null pointer");
  int rc = rel_check("rel_bias_bwd", B, Sq, Sk, R, H);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const long nrows = (long)B * Sq * Sk;
  const long need = (nrows * (R / 4) + 255) / 256;
  const int blocks = (int)(need < 2048 ? (need ? need : 1) : 2048);
  // algorithmic traffic: rel read once, d_rel written once, dbias read once
  ProfScope ps(MMNAS_K_REL_BWD, 6.0 * nrows * R * H,
               4.0 * ((double)nrows * R * (drel ? 2.0 : 1.0) + (double)nrows * H), st);
#define BWDK(LPR, HM, ...) MMNAS_LAUNCH((rel_bias_bwd_kernel<LPR, HM>), dim3(blocks), dim3(256), 0, st, __VA_ARGS__)
  REL_DISPATCH(BWDK, rel, Wr, br, dbiasT, drel, dWr, dbr, accumulate_drel, B, Sq, Sk, H);
#undef BWDK
  return check_launch("rel_bias_bwd");
}

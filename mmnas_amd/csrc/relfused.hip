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
#include "common.h"

namespace mmnas {

constexpr int RF_R = 64;      // REL_SIZE handled by the fused path
constexpr int RF_CP = 8;      // raw channels padded (C + 1 <= 8: the extra column of ones yields dby)
constexpr int RF_HP = 32;     // heads padded to one MFMA tile

struct RelFusedK {
  const float* raw; const float* Wy; const float* by; const float* Wr; const float* br;
  float* biasT; const float* dbiasT; float* part;
  int B, Sq, Sk, C, H, nbq, nbk;
};

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
  const bool ok = rf_coords(p, blockIdx.x, threadIdx.x, b, q, k);
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

// workspace row layout per (workgroup, element-half): [ dWr: HP x R | dWy_ext: R x CP | dbr: HP ]
constexpr int RF_ROW = RF_HP * RF_R + RF_R * RF_CP + RF_HP;

// LDS image of hid/dhid: [element][64] with the column rotated by the element index -- unpadded (so two
// workgroups fit in a CU's 160 KB) yet conflict-free both for the per-thread row writes (lane = element)
// and for the MFMA operand reads (lane = column)
__device__ __forceinline__ int rf_sw(int e, int j) { return e * RF_R + ((j + e) & (RF_R - 1)); }

template <int C, int HP>   // HP = heads padded: 8 (H <= 8, 80 KB LDS -> 2 workgroups/CU) or 32
__global__ void __launch_bounds__(256) rel_fused_bwd_kernel(const RelFusedK p, long nbatch, const float* __restrict__ Wy,
                                                            const float* __restrict__ by, const float* __restrict__ Wr,
                                                            const float* __restrict__ br) {
  __shared__ float sHid[256 * RF_R];         // hid, then dhid, of the 256 elements of a batch
  __shared__ float sDpre[256 * HP];          // dpre[e][h], zero beyond H
  __shared__ float sRaw[256 * RF_CP];        // raw[e][c], 1 at c = C, zero beyond
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int jt = w & 1, eh = w >> 1;         // this wave's 32-column tile and 128-element half
  for (int i = tid; i < 256 * HP; i += 256) sDpre[i] = 0.f;
  f32x16 accWr, accWy;
#pragma unroll
  for (int r = 0; r < 16; ++r) { accWr[r] = 0.f; accWy[r] = 0.f; }
  float accbr = 0.f;                          // wave 0: lane l31 = head, hh = element parity
  const bool arow = l31 < HP;                 // A-operand rows beyond the padded heads are zero
  const int hcl = arow ? l31 : 0;

  for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    int b, q, k;
    const bool ok = rf_coords(p, batch, tid, b, q, k);
    float rawv[C], hid[RF_R];
    rf_hidden<C>(p, Wy, by, ok, b, q, k, rawv, hid);
    __syncthreads();                          // previous batch's MFMA reads are done
#pragma unroll
    for (int j = 0; j < RF_R; ++j) sHid[rf_sw(tid, j)] = ok ? hid[j] : 0.f;
#pragma unroll
    for (int c = 0; c < RF_CP; ++c) sRaw[tid * RF_CP + c] = !ok ? 0.f : (c < C ? rawv[c < C ? c : 0] : (c == C ? 1.f : 0.f));
    float dh[RF_R];
#pragma unroll
    for (int j = 0; j < RF_R; ++j) dh[j] = 0.f;
    for (int h = 0; h < p.H; ++h) {
      float r = br[h];
      const float* wr = Wr + h * RF_R;
      // unconditional (clamped) load, issued before the 64-FMA dot so its latency is covered
      const float db = p.dbiasT[(((size_t)b * p.H + h) * p.Sk + (ok ? k : 0)) * p.Sq + (ok ? q : 0)];
#pragma unroll
      for (int j = 0; j < RF_R; ++j) r += wr[j] * hid[j];
      const float dpre = (ok && r >= 1e-6f) ? db / r : 0.f;
      sDpre[tid * HP + h] = dpre;
#pragma unroll
      for (int j = 0; j < RF_R; ++j) dh[j] += dpre * wr[j];
    }
    __syncthreads();
    // dWr[h][j] += sum_e dpre[e][h] * hid[e][j]   (A = dpre^T, B = hid); operands fetched 8 steps ahead
#pragma unroll
    for (int t0 = 0; t0 < 64; t0 += 8) {
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = 128 * eh + 2 * (t0 + u) + hh;
        av[u] = sDpre[e * HP + hcl];
        bv[u] = sHid[rf_sw(e, 32 * jt + l31)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) accWr = mfma32(arow ? av[u] : 0.f, bv[u], accWr);
    }
    if (w == 0 && arow) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll 8
      for (int e = hh; e < 256; e += 4) { s0 += sDpre[e * HP + l31]; s1 += sDpre[(e + 2) * HP + l31]; }
      accbr += s0 + s1;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RF_R; ++j) sHid[rf_sw(tid, j)] = (ok && hid[j] > 0.f) ? dh[j] : 0.f;   // relu' gate
    __syncthreads();
    // dWy_ext[j][c] += sum_e dhid[e][j] * rawext[e][c]   (A = dhid^T, B = raw | 1)
    const int ccl = l31 < RF_CP ? l31 : 0;
#pragma unroll
    for (int t0 = 0; t0 < 64; t0 += 8) {
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = 128 * eh + 2 * (t0 + u) + hh;
        av[u] = sHid[rf_sw(e, 32 * jt + l31)];
        bv[u] = sRaw[e * RF_CP + ccl];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) accWy = mfma32(av[u], l31 < RF_CP ? bv[u] : 0.f, accWy);
    }
  }
  // partial row of this (workgroup, element-half): waves with the same eh fill disjoint column tiles
  float* row = p.part + ((size_t)blockIdx.x * 2 + eh) * RF_ROW;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = acc_row(r, hh);
    row[i * RF_R + 32 * jt + l31] = accWr[r];                       // dWr[h = i][j]
    if (l31 < RF_CP) row[RF_HP * RF_R + (32 * jt + i) * RF_CP + l31] = accWy[r];   // dWy_ext[j][c]
  }
  if (w == 0) {
    accbr += __shfl_xor(accbr, 32, 64);
    if (hh == 0) row[RF_HP * RF_R + RF_R * RF_CP + l31] = arow ? accbr : 0.f;
  }
  if (w == 2 && hh == 0) row[RF_HP * RF_R + RF_R * RF_CP + l31] = 0.f;    // eh = 1 rows carry no dbr
}

// sum the partial rows and add into the parameter gradients (single writer per output: plain +=)
__global__ void __launch_bounds__(256) rel_fused_reduce_kernel(const float* __restrict__ part, int nrows, int C, int H,
                                                               float* dWr, float* dbr, float* dWy, float* dby) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  float s = 0.f;
  if (col < RF_ROW) {
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = g;
    for (; r + 12 < nrows; r += 16) {
      s += part[(size_t)r * RF_ROW + col];
      s1 += part[(size_t)(r + 4) * RF_ROW + col];
      s2 += part[(size_t)(r + 8) * RF_ROW + col];
      s3 += part[(size_t)(r + 12) * RF_ROW + col];
    }
    for (; r < nrows; r += 4) s += part[(size_t)r * RF_ROW + col];
    s += (s1 + s2) + s3;
  }
  __shared__ float red[4][64];
  red[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g != 0 || col >= RF_ROW) return;
  const int cl = threadIdx.x & 63;
  const float v = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
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

static int rf_grid(long nbatch) { return (int)(nbatch < 512 ? nbatch : 512); }   // persistent: <= 2 per CU

}  // namespace mmnas

using namespace mmnas;

extern "C" int mmnas_rel_fused_supported(int C, int R, int H) { return R == RF_R && (C == 3 || C == 4) && H >= 1 && H <= RF_HP; }

extern "C" int mmnas_rel_fused_fwd(const float* raw, const float* Wy, const float* by, const float* Wr, const float* br,
                                   float* biasT, int B, int Sq, int Sk, int C, int R, int H, void* stream) {
  MMNAS_REQUIRE(raw && Wy && by && Wr && br && biasT, MMNAS_E_ARG, "rel_fused_fwd: null pointer");
  int rc = rf_check("rel_fused_fwd", B, Sq, Sk, C, R, H);
  if (rc) return rc;
  RelFusedK k;
  k.raw = raw; k.Wy = Wy; k.by = by; k.Wr = Wr; k.br = br; k.biasT = biasT; k.dbiasT = nullptr; k.part = nullptr;
  k.B = B; k.Sq = Sq; k.Sk = Sk; k.C = C; k.H = H; k.nbq = cdiv(Sq, 64); k.nbk = cdiv(Sk, 4);
  const long nbatch = (long)B * k.nbq * k.nbk;
  hipStream_t st = (hipStream_t)stream;
  const double n = (double)B * Sq * Sk;
  ProfScope ps(MMNAS_K_REL_FWD, 2.0 * n * (RF_R * (C + 1) + (double)H * RF_R), 4.0 * n * (C + H), st);
  if (C == 4) MMNAS_LAUNCH(rel_fused_fwd_kernel<4>, dim3((unsigned)nbatch), dim3(256), 0, st, k, Wy, by, Wr, br);
  else MMNAS_LAUNCH(rel_fused_fwd_kernel<3>, dim3((unsigned)nbatch), dim3(256), 0, st, k, Wy, by, Wr, br);
  return check_launch("rel_fused_fwd");
}

extern "C" size_t mmnas_rel_fused_bwd_ws_floats(int B, int Sq, int Sk) {
  const long nbatch = (long)B * cdiv(Sq, 64) * cdiv(Sk, 4);
  return (size_t)rf_grid(nbatch) * 2 * RF_ROW;
}

extern "C" int mmnas_rel_fused_bwd(const float* raw, const float* Wy, const float* by, const float* Wr, const float* br,
                                   const float* dbiasT, float* dWy, float* dby, float* dWr, float* dbr, float* ws,
                                   int B, int Sq, int Sk, int C, int R, int H, void* stream) {
  MMNAS_REQUIRE(raw && Wy && by && Wr && br && dbiasT && dWy && dby && dWr && dbr && ws, MMNAS_E_ARG,
                "rel_fused_bwd: null pointer");
  int rc = rf_check("rel_fused_bwd", B, Sq, Sk, C, R, H);
  if (rc) return rc;
  RelFusedK k;
  k.raw = raw; k.Wy = Wy; k.by = by; k.Wr = Wr; k.br = br; k.biasT = nullptr; k.dbiasT = dbiasT; k.part = ws;
  k.B = B; k.Sq = Sq; k.Sk = Sk; k.C = C; k.H = H; k.nbq = cdiv(Sq, 64); k.nbk = cdiv(Sk, 4);
  const long nbatch = (long)B * k.nbq * k.nbk;
  const int grid = rf_grid(nbatch);
  hipStream_t st = (hipStream_t)stream;
  const double n = (double)B * Sq * Sk;
  ProfScope ps(MMNAS_K_REL_BWD, 2.0 * n * (RF_R * (C + 1) + 3.0 * H * RF_R + RF_R * (C + 1)), 4.0 * n * (C + H), st);
  if (H <= 8) {
    if (C == 4) MMNAS_LAUNCH((rel_fused_bwd_kernel<4, 8>), dim3(grid), dim3(256), 0, st, k, nbatch, Wy, by, Wr, br);
    else MMNAS_LAUNCH((rel_fused_bwd_kernel<3, 8>), dim3(grid), dim3(256), 0, st, k, nbatch, Wy, by, Wr, br);
  } else {
    if (C == 4) MMNAS_LAUNCH((rel_fused_bwd_kernel<4, 32>), dim3(grid), dim3(256), 0, st, k, nbatch, Wy, by, Wr, br);
    else MMNAS_LAUNCH((rel_fused_bwd_kernel<3, 32>), dim3(grid), dim3(256), 0, st, k, nbatch, Wy, by, Wr, br);
  }
  MMNAS_LAUNCH(rel_fused_reduce_kernel, dim3(cdiv(RF_ROW, 64)), dim3(256), 0, st, ws, grid * 2, C, H, dWr, dbr, dWy,
                     dby);
  return check_launch("rel_fused_bwd");
}

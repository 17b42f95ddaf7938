// Stem / head kernels:
//   * make_mask (hygr_vqa.py:121-122): mask[r] = (sum_j |f[r,j]| == 0), one pass over the 52 MB region-feature
//     tensor instead of abs -> sum -> compare (two 52 MB passes plus a 52 MB temporary).
//   * AttFlat pooling (modules.py:78-84): masked softmax of the glimpse logits over the sequence and the
//     attention-weighted sum of the features, forward and backward, one workgroup per batch element
//     (replaces masked_fill, softmax, mul, sum, cat and their five autograd kernels).
#include "common.h"

namespace mmnas {

// one wave per row; a row is "padding" iff every element is +-0 (a NaN makes sum|f| NaN != 0: not padding)
__global__ void __launch_bounds__(256) row_is_zero_kernel(const float* __restrict__ f, uint8_t* __restrict__ mask,
                                                          long rows, int d) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* fr = f + row * d;
  bool nz = false;
  if ((d & 3) == 0 && (((uintptr_t)fr) & 15) == 0) {
    for (int c = lane * 4; c < d; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(fr + c);
      nz |= !(v.x == 0.f && v.y == 0.f && v.z == 0.f && v.w == 0.f);
    }
  } else {
    for (int c = lane; c < d; c += 64) nz |= !(fr[c] == 0.f);
  }
  const unsigned long long any = __ballot(nz);
  if (lane == 0) mask[row] = any ? 0 : 1;
}

constexpr int AF_MAXS = 1024;   // sequence positions handled by one workgroup

// probs[b,s,g] = softmax_s(logits[b,s,g] masked to -1e9);  pooled[b, g*d + j] = sum_s probs[b,s,g] x[b,s,j]
// One workgroup of AF_THREADS per (sample, glimpse): only B*G workgroups exist (64-128 on 256 CUs), so the kernel is
// latency-bound -- 16 waves and 4-8 independent loads per thread keep enough of the 200 KB sample in flight.
constexpr int AF_THREADS = 1024, AF_WAVES = AF_THREADS / 64;

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  v = is_max ? wave_max(v) : wave_sum(v);
  __syncthreads();   // red may still be read from the previous reduction
  if (lane == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < AF_WAVES; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
  return r;
}

__global__ void __launch_bounds__(AF_THREADS) attflat_pool_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ x,
                                                                      const uint8_t* __restrict__ mask, float* __restrict__ probs,
                                                                      float* __restrict__ pooled, int S, int d, int G,
                                                                      const int* __restrict__ off) {
  __shared__ float sp[AF_MAXS];
  __shared__ float red[AF_WAVES];
  __shared__ float part[AF_WAVES / 4][1024];   // partial column sums of the row groups (d <= 1024 per pass)
  const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
  // PACKED rows (off != NULL): sample b owns rows off[b] .. off[b+1] of logits / x / probs, all of them valid (no mask)
  const size_t row0 = off ? (size_t)off[b] : (size_t)b * S;
  if (off) S = off[b + 1] - off[b];
  const float* lg = logits + row0 * G + g;
  float m = -INFINITY;
  for (int s = tid; s < S; s += AF_THREADS) {
    float v = lg[(size_t)s * G];
    if (mask && !off && mask[(size_t)b * S + s]) v = -1e9f;
    sp[s] = v;
    m = fmaxf(m, v);
  }
  m = block_reduce(m, red, true);
  float sum = 0.f;
  for (int s = tid; s < S; s += AF_THREADS) {
    const float e = expf(sp[s] - m);
    sp[s] = e;
    sum += e;
  }
  sum = block_reduce(sum, red, false);
  const float inv = 1.0f / sum;
  for (int s = tid; s < S; s += AF_THREADS) {
    const float pr = sp[s] * inv;
    sp[s] = pr;
    probs[(row0 + s) * G + g] = pr;
  }
  __syncthreads();
  // pooled: 256 column threads x 4 row groups (rows s = rg mod 4), 4 independent accumulators each
  const float* xb = x + row0 * d;
  const int col = tid & 255, rg = tid >> 8;
  for (int j0 = 0; j0 < d; j0 += 1024) {
    for (int j = j0 + col; j < min(d, j0 + 1024); j += 256) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int s = rg;
      for (; s + 12 < S; s += 16) {
        a0 += sp[s] * xb[(size_t)s * d + j];
        a1 += sp[s + 4] * xb[(size_t)(s + 4) * d + j];
        a2 += sp[s + 8] * xb[(size_t)(s + 8) * d + j];
        a3 += sp[s + 12] * xb[(size_t)(s + 12) * d + j];
      }
      for (; s < S; s += 4) a0 += sp[s] * xb[(size_t)s * d + j];
      part[rg][j - j0] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    for (int j = j0 + tid; j < min(d, j0 + 1024); j += AF_THREADS)
      pooled[(size_t)b * G * d + (size_t)g * d + j] = (part[0][j - j0] + part[1][j - j0]) + (part[2][j - j0] + part[3][j - j0]);
    __syncthreads();
  }
}

// t[s,g] = x[b,s,:] . dpooled[b,g,:];  dlogits[b,s,g] = p (t - sum_s' p t);  dx[b,s,:] = sum_g p[b,s,g] dpooled[b,g,:]
// (a masked position's logit was REPLACED by -1e9, so its gradient is zero even when a fully masked sequence gives
//  it the probability 1/S)
__global__ void __launch_bounds__(AF_THREADS) attflat_pool_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ x,
                                                                      const uint8_t* __restrict__ mask,
                                                                      const float* __restrict__ dpooled, float* __restrict__ dlogits,
                                                                      float* __restrict__ dx, int S, int d, int G,
                                                                      const int* __restrict__ off) {
  __shared__ float st[AF_MAXS];
  __shared__ float red[AF_WAVES];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t row0 = off ? (size_t)off[b] : (size_t)b * S;
  if (off) S = off[b + 1] - off[b];
  const float* xb = x + row0 * d;
  for (int g = 0; g < G; ++g) {
    const float* dp = dpooled + (size_t)b * G * d + (size_t)g * d;
    // t[s]: one wave per row, lanes over the feature dimension
    for (int s = w; s < S; s += AF_WAVES) {
      float a = 0.f;
      for (int j = lane; j < d; j += 64) a += xb[(size_t)s * d + j] * dp[j];
      a = wave_sum(a);
      if (lane == 0) st[s] = a;
    }
    __syncthreads();
    float dot = 0.f;
    for (int s = tid; s < S; s += AF_THREADS) dot += probs[(row0 + s) * G + g] * st[s];
    dot = block_reduce(dot, red, false);
    for (int s = tid; s < S; s += AF_THREADS) {
      const float pr = probs[(row0 + s) * G + g];
      const bool masked = mask && !off && mask[(size_t)b * S + s];
      dlogits[(row0 + s) * G + g] = masked ? 0.f : pr * (st[s] - dot);
    }
    __syncthreads();
  }
  // dx: rows over waves, lanes over features
  for (int s = w; s < S; s += AF_WAVES) {
    for (int j = lane; j < d; j += 64) {
      float a = 0.f;
      for (int g = 0; g < G; ++g) a += probs[(row0 + s) * G + g] * dpooled[(size_t)b * G * d + (size_t)g * d + j];
      dx[(row0 + s) * d + j] = a;
    }
  }
}

// ------------------------------------------------------------------------------------------
// One glimpse (ATTFLAT_GLIMPSES = 1, every shipped configuration): the glimpse-logit layer of AttFlat's MLP
// (MLP.linear, modules.py:34-41) is a matrix-VECTOR product and its backward an outer product plus three reductions.
// On the GEMM kernel they were launches of 64-wide tiles with one live column or one K step (13-27 us each on the guarded
// path: N = 1 / K = 1 is no shape for vector loads); here they are one pass over h each.
//   forward : logit[r] = h[r,:] . w2 + b2                                  (both AttFlat sides in one launch)
//   backward: dh[r,c] = h[r,c] > 0 ? dlog[r] w2[c] gate_scale : 0          (gated: relu' and the dropout replay from h; the
//                                                                           per-operator path passes gated = 0, scale 1)
//             db1[c] += sum_r dh[r,c];  dW2[c] += sum_r dlog[r] h[r,c];  db2 += sum_r dlog[r]
// ------------------------------------------------------------------------------------------
struct Glimpse1Side { const float* h; const float* w2; const float* b2; float* logit; long rows; };

__global__ void __launch_bounds__(256) glimpse1_fwd_kernel(const Glimpse1Side s0, const Glimpse1Side s1, int MID) {
  const int lane = threadIdx.x & 63;
  long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool first = row < s0.rows;
  const Glimpse1Side& s = first ? s0 : s1;
  if (!first) row -= s0.rows;
  if (row >= s.rows) return;
  const float* hr = s.h + row * MID;
  float a = 0.f;
  for (int c = lane * 4; c < MID; c += 256) {
    const float4 hv = *reinterpret_cast<const float4*>(hr + c);
    const float4 wv = *reinterpret_cast<const float4*>(s.w2 + c);
    a += (hv.x * wv.x + hv.y * wv.y) + (hv.z * wv.z + hv.w * wv.w);
  }
  a = wave_sum(a);
  if (lane == 0) s.logit[row] = a + (s.b2 ? s.b2[0] : 0.f);
}

// Backward: many small workgroups (a thread walks <= 8 rows with its loads issued ahead: the pass is latency-bound, one
// block per CU with a serial row loop took 50-78 us); a workgroup's column sums go to partial row blockIdx of a
// [blocks][3][MID] buffer (planes db1 | dW2 | unused) that the NEXT gradient-pair launch reduces with a few extra
// workgroups (AuxReduce, as for the LayerNorm parameter gradients) -- same-address float atomics from hundreds of
// workgroups cost ~120 ns each.  db2 (one number) stays a column-sum launch of dlog.
constexpr int G1_ITER = 4;    // rows per thread
__global__ void __launch_bounds__(256) glimpse1_bwd_kernel(const float* __restrict__ dlog, const float* __restrict__ h,
                                                           const float* __restrict__ w2, float gate_scale, int gated,
                                                           float* __restrict__ dh, float* __restrict__ part, long rows, int MID) {
  __shared__ float red[2][1024];   // [db1 | dW2][slot * MID + column]: rs * MID <= 1024 (rs = 256 / (MID / 4) row slots)
  const int nc4 = MID >> 2;               // threads along the columns (<= 256)
  const int rs = 256 / nc4;               // row slots of the workgroup
  const int c4 = threadIdx.x % nc4, slot = threadIdx.x / nc4;
  const long r0 = (long)blockIdx.x * (rs * G1_ITER);
  float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sw = sb;
  if (slot < rs) {
    const float4 wv = *reinterpret_cast<const float4*>(w2 + 4 * c4);
    float dl[G1_ITER];
    float4 hv[G1_ITER];
#pragma unroll
    for (int i = 0; i < G1_ITER; ++i) {   // all loads first
      const long r = r0 + slot + (long)i * rs;
      const long rc = r < rows ? r : rows - 1;
      dl[i] = r < rows ? dlog[rc] : 0.f;
      hv[i] = *reinterpret_cast<const float4*>(h + rc * MID + 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < G1_ITER; ++i) {
      const long r = r0 + slot + (long)i * rs;
      float4 o;
      o.x = (!gated || hv[i].x > 0.f) ? dl[i] * wv.x * gate_scale : 0.f; o.y = (!gated || hv[i].y > 0.f) ? dl[i] * wv.y * gate_scale : 0.f;
      o.z = (!gated || hv[i].z > 0.f) ? dl[i] * wv.z * gate_scale : 0.f; o.w = (!gated || hv[i].w > 0.f) ? dl[i] * wv.w * gate_scale : 0.f;
      if (r < rows) *reinterpret_cast<float4*>(dh + r * MID + 4 * c4) = o;
      sb.x += o.x; sb.y += o.y; sb.z += o.z; sb.w += o.w;          // (rows behind the end: dl = 0)
      sw.x += dl[i] * hv[i].x; sw.y += dl[i] * hv[i].y; sw.z += dl[i] * hv[i].z; sw.w += dl[i] * hv[i].w;
    }
    *reinterpret_cast<float4*>(&red[0][slot * MID + 4 * c4]) = sb;
    *reinterpret_cast<float4*>(&red[1][slot * MID + 4 * c4]) = sw;
  }
  __syncthreads();
  float* mine = part + (size_t)blockIdx.x * 3 * MID;
  for (int i = threadIdx.x; i < 2 * MID; i += 256) {
    const int a = i < MID ? 0 : 1, c = i - a * MID;
    float v = 0.f;
    for (int q = 0; q < rs; ++q) v += red[a][q * MID + c];
    mine[a * MID + c] = v;
  }
}

// out[c][r] = in[r][c] (in [R][C] row-major): the loss gradient of an answer layer whose width is no multiple of 4 (3129
// answers) transposed once, so that both of the projection's gradient products get 16-byte-aligned operand rows
__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
  __shared__ float t[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < R && c0 + tx < C) t[i][tx] = in[(size_t)(r0 + i) * C + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < C && r0 + tx < R) out[(size_t)(c0 + i) * R + r0 + tx] = t[tx][i];
}
int transpose2d(const float* in, float* out, int R, int C, hipStream_t st) {
  MMNAS_REQUIRE(in && out && R > 0 && C > 0, MMNAS_E_ARG, "transpose2d: bad arguments");
  ProfScope ps(MMNAS_K_ROWOPS, 0.0, 8.0 * R * C, st);
  MMNAS_LAUNCH(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, st, in, out, R, C);
  return check_launch("transpose2d");
}

bool glimpse1_supported(int MID) { return MID >= 4 && MID % 4 == 0 && MID <= 1024; }

// logits of both AttFlat sides (side 1 may have rows = 0)
int glimpse1_fwd(const float* h0, const float* w0, const float* b0, float* l0, long rows0, const float* h1, const float* w1,
                 const float* b1, float* l1, long rows1, int MID, hipStream_t st) {
  MMNAS_REQUIRE(glimpse1_supported(MID) && h0 && w0 && l0 && rows0 > 0 && (rows1 == 0 || (h1 && w1 && l1)), MMNAS_E_ARG,
                "glimpse1_fwd: bad arguments");
  const Glimpse1Side s0{h0, w0, b0, l0, rows0}, s1{h1, w1, b1, l1, rows1};
  ProfScope ps(MMNAS_K_ROWOPS, 2.0 * (rows0 + rows1) * MID, 4.0 * (rows0 + rows1) * MID, st);
  MMNAS_LAUNCH(glimpse1_fwd_kernel, dim3((unsigned)((rows0 + rows1 + 3) / 4)), dim3(256), 0, st, s0, s1, MID);
  return check_launch("glimpse1_fwd");
}

int glimpse1_bwd_blocks(long rows, int MID) {
  const int rs = 256 / (MID >> 2);
  return (int)((rows + rs * G1_ITER - 1) / (rs * G1_ITER));
}

// dh and the partial column sums part[blocks][3][MID]; *aux names the pending reduction (db1 += plane 0, dW2 += plane 1)
int glimpse1_bwd(const float* dlog, const float* h, const float* w2, float gate_scale, int gated, float* dh, float* db1, float* dW2,
                 float* part, long rows, int MID, hipStream_t st, AuxReduce* aux) {
  MMNAS_REQUIRE(glimpse1_supported(MID) && dlog && h && w2 && dh && dW2 && part && aux && rows > 0, MMNAS_E_ARG, "glimpse1_bwd: bad arguments");
  const int nb = glimpse1_bwd_blocks(rows, MID);
  ProfScope ps(MMNAS_K_ROWOPS, 5.0 * rows * MID, 8.0 * rows * MID, st);
  MMNAS_LAUNCH(glimpse1_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, st, dlog, h, w2, gate_scale, gated, dh, part, rows, MID);
  aux->part = part; aux->nrows = nb; aux->d = MID;
  aux->out[0] = db1; aux->out[1] = dW2; aux->out[2] = nullptr;
  return check_launch("glimpse1_bwd");
}

}  // namespace mmnas

using namespace mmnas;

extern "C" int mmnas_row_is_zero(const float* f, uint8_t* mask, long rows, int d, void* stream) {
  MMNAS_REQUIRE(f && mask && rows > 0 && d > 0, MMNAS_E_ARG, "row_is_zero: bad arguments");
  MMNAS_LAUNCH(row_is_zero_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, f, mask, rows, d);
  return check_launch("row_is_zero");
}

extern "C" int mmnas_glimpse1_supported(int K) { return glimpse1_supported(K) ? 1 : 0; }
extern "C" size_t mmnas_glimpse1_bwd_ws_floats(long rows, int K) {
  return glimpse1_supported(K) && rows > 0 ? (size_t)glimpse1_bwd_blocks(rows, K) * 3 * K : 0;
}
extern "C" int mmnas_glimpse1_fwd(const float* x, const float* w, const float* b, float* y, long rows, int K, void* stream) {
  return glimpse1_fwd(x, w, b, y, rows, nullptr, nullptr, nullptr, nullptr, 0, K, (hipStream_t)stream);
}
extern "C" int mmnas_glimpse1_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, float* ws, long rows,
                                  int K, void* stream) {
  MMNAS_REQUIRE(ws, MMNAS_E_ARG, "glimpse1_bwd: null workspace");
  AuxReduce aux;
  int rc = glimpse1_bwd(dy, x, w, 1.0f, 0, dx, nullptr, dw, ws, rows, K, (hipStream_t)stream, &aux);
  if (rc) return rc;
  if ((rc = launch_aux_reduce(aux, (hipStream_t)stream))) return rc;
  return db ? mmnas_colsum(dy, db, (int)rows, 1, 1, stream) : MMNAS_OK;
}

extern "C" int mmnas_attflat_pool_fwd(const float* logits, const float* x, const uint8_t* mask, float* probs,
                                      float* pooled, int B, int S, int d, int G, void* stream) {
  MMNAS_REQUIRE(logits && x && probs && pooled, MMNAS_E_ARG, "attflat_pool_fwd: null pointer");
  MMNAS_REQUIRE(B > 0 && S > 0 && S <= AF_MAXS && d > 0 && G > 0, MMNAS_E_SHAPE, "attflat_pool_fwd: B=%d S=%d d=%d G=%d (S <= %d)",
                B, S, d, G, AF_MAXS);
  MMNAS_LAUNCH(attflat_pool_fwd_kernel, dim3(B, G), dim3(AF_THREADS), 0, (hipStream_t)stream, logits, x, mask, probs, pooled, S, d, G, (const int*)nullptr);
  return check_launch("attflat_pool_fwd");
}
namespace mmnas {
// the same over PACKED rows (sample b = rows off[b] .. off[b+1], S = the longest sample; no mask: every packed row is valid)
int attflat_pool_fwd_packed(const float* logits, const float* x, float* probs, float* pooled, int B, int S, int d, int G, const int* off, hipStream_t st) {
  MMNAS_REQUIRE(logits && x && probs && pooled && off, MMNAS_E_ARG, "attflat_pool_fwd: null pointer");
  MMNAS_REQUIRE(B > 0 && S > 0 && S <= AF_MAXS && d > 0 && G > 0, MMNAS_E_SHAPE, "attflat_pool_fwd: B=%d S=%d d=%d G=%d (S <= %d)", B, S, d, G, AF_MAXS);
  MMNAS_LAUNCH(attflat_pool_fwd_kernel, dim3(B, G), dim3(AF_THREADS), 0, st, logits, x, (const uint8_t*)nullptr, probs, pooled, S, d, G, off);
  return check_launch("attflat_pool_fwd");
}
int attflat_pool_bwd_packed(const float* probs, const float* x, const float* dpooled, float* dlogits, float* dx, int B, int S, int d, int G,
                            const int* off, hipStream_t st) {
  MMNAS_REQUIRE(probs && x && dpooled && dlogits && dx && off, MMNAS_E_ARG, "attflat_pool_bwd: null pointer");
  MMNAS_REQUIRE(B > 0 && S > 0 && S <= AF_MAXS && d > 0 && G > 0, MMNAS_E_SHAPE, "attflat_pool_bwd: B=%d S=%d d=%d G=%d (S <= %d)", B, S, d, G, AF_MAXS);
  MMNAS_LAUNCH(attflat_pool_bwd_kernel, dim3(B), dim3(AF_THREADS), 0, st, probs, x, (const uint8_t*)nullptr, dpooled, dlogits, dx, S, d, G, off);
  return check_launch("attflat_pool_bwd");
}
}  // namespace mmnas

extern "C" int mmnas_attflat_pool_bwd(const float* probs, const float* x, const uint8_t* mask, const float* dpooled,
                                      float* dlogits, float* dx, int B, int S, int d, int G, void* stream) {
  MMNAS_REQUIRE(probs && x && dpooled && dlogits && dx, MMNAS_E_ARG, "attflat_pool_bwd: null pointer");
  MMNAS_REQUIRE(B > 0 && S > 0 && S <= AF_MAXS && d > 0 && G > 0, MMNAS_E_SHAPE, "attflat_pool_bwd: B=%d S=%d d=%d G=%d (S <= %d)",
                B, S, d, G, AF_MAXS);
  MMNAS_LAUNCH(attflat_pool_bwd_kernel, dim3(B), dim3(AF_THREADS), 0, (hipStream_t)stream, probs, x, mask, dpooled, dlogits, dx, S, d, G, (const int*)nullptr);
  return check_launch("attflat_pool_bwd");
}

// ------------------------------------------------------------------------------------------
// Data path: the box-geometry relation features the reference's loaders compute per sample on
// the CPU (relation_embedding, load_data_vqa.py:224-239 / load_data_vgd.py:7-33) and ship as a [100,100,4] tensor
// (10 MB per batch of 64 over PCIe).  Batched on the GPU from the [B,S,4] boxes (6 KB per batch):
//   w_i = x2-x1+1, h_i = y2-y1+1, c = box centre
//   out[b,i,j] = ( log max(|cx_i-cx_j| / w_i, 1e-3), log max(|cy_i-cy_j| / h_i, 1e-3), log(w_i/w_j), log(h_i/h_j) )
// for i, j < nobj[b]; zero elsewhere (the loaders zero-pad to S).
// ------------------------------------------------------------------------------------------
namespace mmnas {
__global__ void __launch_bounds__(256) relation_embedding_kernel(const float* __restrict__ bbox, const int* __restrict__ nobj,
                                                                 float* __restrict__ out, int S) {
  const int b = blockIdx.y, i = blockIdx.x;
  const int n = nobj ? nobj[b] : S;
  const float* bi = bbox + ((size_t)b * S + i) * 4;
  const float wi = (bi[2] - bi[0]) + 1.f, hi = (bi[3] - bi[1]) + 1.f;
  const float cxi = (bi[0] + bi[2]) * 0.5f, cyi = (bi[1] + bi[3]) * 0.5f;
  float4* o = reinterpret_cast<float4*>(out + ((size_t)b * S + i) * S * 4);
  for (int j = threadIdx.x; j < S; j += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n && j < n) {
      const float* bj = bbox + ((size_t)b * S + j) * 4;
      const float wj = (bj[2] - bj[0]) + 1.f, hj = (bj[3] - bj[1]) + 1.f;
      const float cxj = (bj[0] + bj[2]) * 0.5f, cyj = (bj[1] + bj[3]) * 0.5f;
      v.x = logf(fmaxf(fabsf((cxi - cxj) / wi), 1e-3f));
      v.y = logf(fmaxf(fabsf((cyi - cyj) / hi), 1e-3f));
      v.z = logf(wi / wj);
      v.w = logf(hi / hj);
    }
    o[j] = v;
  }
}
}  // namespace mmnas

extern "C" int mmnas_relation_embedding(const float* bbox, const int* nobj, float* out, int B, int S, void* stream) {
  MMNAS_REQUIRE(bbox && out && B > 0 && S > 0, MMNAS_E_ARG, "relation_embedding: bad arguments");
  MMNAS_REQUIRE((((uintptr_t)out) & 15) == 0, MMNAS_E_ARG, "relation_embedding: output not 16-byte aligned");
  MMNAS_LAUNCH(relation_embedding_kernel, dim3(S, B), dim3(256), 0, (hipStream_t)stream, bbox, nobj, out, S);
  return check_launch("relation_embedding");
}

// ------------------------------------------------------------------------------------------
// Data path: the token-relation features of the loaders (semantic_embedding, load_data_vqa.py:36-58), batched on the
// GPU from the token indices and the GloVe table (the reference builds a [14,14,3] tensor per sample on the CPU):
//   g_i = emb[ques_ix[b,i]] for the first n = nwords[b] (<= S) tokens,
//   out[b,i,j] = ( |g_i - g_j|_2,  <g_i, g_j> / (sqrt|g_i| sqrt|g_j| + 1e-6),  |i - j| / n ),  zero for i or j >= n
// (the reference divides by the square ROOTS of the norms: kept).  One workgroup per sample; rows staged in LDS.
// ------------------------------------------------------------------------------------------
namespace mmnas {
constexpr int SE_MAXS = 64, SE_MAXE = 320;
__global__ void __launch_bounds__(256) semantic_embedding_kernel(const long* __restrict__ ques, const int* __restrict__ nwords,
                                                                 const float* __restrict__ emb, float* __restrict__ out, int S,
                                                                 int E, long V) {
  extern __shared__ float g[];          // [S][E + 1] rows, then [S] sqrt-norms
  const int b = blockIdx.x, tid = threadIdx.x, ld = E + 1;
  float* sn = g + (size_t)S * ld;
  const int n = min(max(nwords[b], 0), S);
  for (int i = tid; i < n * E; i += 256) {
    const int r = i / E, c = i - r * E;
    long t = ques[(size_t)b * S + r];
    if (t < 0 || t >= V) t = 0;
    g[r * ld + c] = emb[t * E + c];
  }
  __syncthreads();
  for (int r = tid >> 6; r < n; r += 4) {   // one wave per row: squared norm
    float s = 0.f;
    for (int c = tid & 63; c < E; c += 64) { const float v = g[r * ld + c]; s += v * v; }
    s = wave_sum(s);
    if ((tid & 63) == 0) sn[r] = sqrtf(sqrtf(s));   // sqrt(|g_r|_2)
  }
  __syncthreads();
  for (int e = tid; e < S * S; e += 256) {
    const int i = e / S, j = e - i * S;
    float l2 = 0.f, cs = 0.f, ps = 0.f;
    if (i < n && j < n) {
      float d2 = 0.f, dot = 0.f;
      for (int c = 0; c < E; ++c) {
        const float a = g[i * ld + c], bb = g[j * ld + c];
        const float d = a - bb;
        d2 += d * d;
        dot += a * bb;
      }
      l2 = sqrtf(d2);
      cs = dot / (sn[i] * sn[j] + 1e-6f);
      ps = fabsf((float)(i - j)) / (float)n;
    }
    float* o = out + (((size_t)b * S + i) * S + j) * 3;
    o[0] = l2; o[1] = cs; o[2] = ps;
  }
}
}  // namespace mmnas

extern "C" int mmnas_semantic_embedding(const long* ques_ix, const int* nwords, const float* emb, float* out, int B, int S,
                                        int E, long V, void* stream) {
  MMNAS_REQUIRE(ques_ix && nwords && emb && out && B > 0, MMNAS_E_ARG, "semantic_embedding: bad arguments");
  MMNAS_REQUIRE(S >= 1 && S <= SE_MAXS && E >= 1 && E <= SE_MAXE && V >= 1, MMNAS_E_SHAPE,
                "semantic_embedding: S=%d (<= %d) E=%d (<= %d)", S, SE_MAXS, E, SE_MAXE);
  const size_t lds = ((size_t)S * (E + 1) + S) * sizeof(float);
  MMNAS_LAUNCH(semantic_embedding_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, ques_ix, nwords, emb, out, S, E, V);
  return check_launch("semantic_embedding");
}

// ------------------------------------------------------------------------------------------
// Supernet plumbing: write the binary gates of all nodes (MixedOp.binarize, mixed.py:131-158: alpha_gate = one-hot
// of the sampled operator) in one launch whose indices travel in the kernel arguments -- no host->device copy,
// hence no stream synchronisation, per NAS step.
// ------------------------------------------------------------------------------------------
namespace mmnas {
struct OneHotArgs { int idx[128]; };
__global__ void onehot_rows_kernel(float* __restrict__ out, int rows, int width, OneHotArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * width) return;
  const int r = i / width, c = i - r * width;
  out[i] = a.idx[r] == c ? 1.f : 0.f;
}
}  // namespace mmnas

extern "C" int mmnas_onehot_rows(float* out, int rows, int width, const int* idx_host, void* stream) {
  MMNAS_REQUIRE(out && idx_host && rows > 0 && rows <= 128 && width > 0, MMNAS_E_ARG, "onehot_rows: rows=%d (1..128) width=%d", rows, width);
  OneHotArgs a;
  for (int r = 0; r < 128; ++r) a.idx[r] = r < rows ? idx_host[r] : -1;
  MMNAS_LAUNCH(onehot_rows_kernel, dim3(cdiv((long)rows * width, 256)), dim3(256), 0, (hipStream_t)stream, out, rows, width, a);
  return check_launch("onehot_rows");
}

// ------------------------------------------------------------------------------------------
// Embedding backward (nn.Embedding of the language stem, hygr_vqa.py:85,105): dW[idx[t], :] += dy[t, :].
// ATen builds a dense [V, E] gradient (24 MB at V = 20000: zero-fill + scatter kernel, 61 us) that autograd then adds
// onto the parameter's gradient (another 48 MB pass); the rows of the ~900 tokens of a batch are all that changes, so
// they are added straight into the gradient buffer.
// ------------------------------------------------------------------------------------------
namespace mmnas {
__global__ void __launch_bounds__(256) embedding_bwd_kernel(const long* __restrict__ idx, const float* __restrict__ dy,
                                                            float* __restrict__ dW, long n_tok, int E, long V) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_tok * E) return;
  const long t = i / E;
  const int c = (int)(i - t * E);
  const long row = idx[t];
  if (row >= 0 && row < V) atomicAdd(dW + row * E + c, dy[i]);
}
}  // namespace mmnas

namespace mmnas {
// The same scatter-add with a FIXED summation order and no atomics, in two levels (the padding token alone is half of a
// batch's tokens: one workgroup adding ~450 rows one after the other took 180 us):
//   1. chunks of 64 tokens: workgroup t owns token t's row inside its chunk if t is the first token of the chunk holding
//      it, and writes the sum of the chunk's rows with that index, in token order, to part[t];
//   2. workgroup t owns the row globally if t is the first token of the whole batch holding it, and adds the chunk sums
//      part[first token of the row in chunk c], c ascending, onto dW (scaled).
// Used by the data-parallel exchange of the embedding gradient (dp.RowExchange): every rank applies the gathered
// (index, dy) pairs of all ranks itself, and the ranks' tables must stay bitwise identical -- which float atomics do not
// promise.  One wave's ballot over a chunk's 64 indices finds the owners.
__global__ void __launch_bounds__(256) embedding_bwd_det1_kernel(const long* __restrict__ idx, const float* __restrict__ dy,
                                                                 float* __restrict__ part, int n_tok, int E, long V) {
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const long row = idx[t];
  if (row < 0 || row >= V) return;
  const int c0 = t & ~63, j = c0 + lane;
  // (every wave forms the chunk's match mask itself: no LDS, no barrier)
  unsigned long long mm = __ballot(j < n_tok && idx[j] == row);
  if (mm & ((1ull << (t - c0)) - 1ull)) return;          // an earlier token of the chunk owns the row
  float acc[4] = {0.f, 0.f, 0.f, 0.f};                    // columns tid, tid + 256, ... (E <= 1024)
  while (mm) {
    const float* src = dy + (size_t)(c0 + __builtin_ctzll(mm)) * E;
    mm &= mm - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (tid + 256 * k < E) acc[k] += src[tid + 256 * k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (tid + 256 * k < E) part[(size_t)t * E + tid + 256 * k] = acc[k];
}

__global__ void __launch_bounds__(256) embedding_bwd_det2_kernel(const long* __restrict__ idx, const float* __restrict__ part,
                                                                 float* __restrict__ dW, int n_tok, int E, long V, float scale) {
  __shared__ int first_of[4];
  __shared__ int earlier;
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row = idx[t];
  if (row < 0 || row >= V) return;
  if (tid == 0) earlier = 0;
  __syncthreads();
  for (int j = tid; j < t; j += 256)
    if (idx[j] == row) earlier = 1;   // (benign race: every writer stores 1)
  __syncthreads();
  if (earlier) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int base = t & ~63; base < n_tok; base += 256) {   // four chunks per round, one per wave
    const int j = base + 64 * wave + lane;
    const unsigned long long mm = __ballot(j < n_tok && idx[j] == row);
    if (lane == 0) first_of[wave] = mm ? base + 64 * wave + __builtin_ctzll(mm) : -1;
    __syncthreads();
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      const int f = first_of[w];
      if (f < 0) continue;
      const float* src = part + (size_t)f * E;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (tid + 256 * k < E) acc[k] += src[tid + 256 * k];
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (tid + 256 * k < E) dW[row * E + tid + 256 * k] += acc[k] * scale;
}
}  // namespace mmnas

extern "C" size_t mmnas_embedding_bwd_det_ws_floats(long n_tok, int E) { return (size_t)(n_tok > 0 ? n_tok : 0) * (size_t)(E > 0 ? E : 0); }

extern "C" int mmnas_embedding_bwd_det(const long* idx, const float* dy, float* dW, float* ws, long n_tok, int E, long V, float scale,
                                       void* stream) {
  MMNAS_REQUIRE(idx && dy && dW && ws && n_tok > 0 && E > 0 && V > 0, MMNAS_E_ARG, "embedding_bwd_det: bad arguments");
  MMNAS_REQUIRE(E <= 1024 && n_tok < (1l << 30), MMNAS_E_SHAPE, "embedding_bwd_det: E=%d (<= 1024) n_tok=%ld", E, n_tok);
  hipStream_t st = (hipStream_t)stream;
  MMNAS_LAUNCH(embedding_bwd_det1_kernel, dim3((unsigned)n_tok), dim3(256), 0, st, idx, dy, ws, (int)n_tok, E, V);
  MMNAS_LAUNCH(embedding_bwd_det2_kernel, dim3((unsigned)n_tok), dim3(256), 0, st, idx, (const float*)ws, dW, (int)n_tok, E, V, scale);
  return check_launch("embedding_bwd_det");
}

extern "C" int mmnas_embedding_bwd(const long* idx, const float* dy, float* dW, long n_tok, int E, long V, void* stream) {
  MMNAS_REQUIRE(idx && dy && dW && n_tok > 0 && E > 0 && V > 0, MMNAS_E_ARG, "embedding_bwd: bad arguments");
  MMNAS_LAUNCH(embedding_bwd_kernel, dim3((unsigned)cdiv(n_tok * E, 256)), dim3(256), 0, (hipStream_t)stream, idx, dy, dW, n_tok, E, V);
  return check_launch("embedding_bwd");
}


// ---- BCEWithLogitsLoss(reduction='sum') (search_vqa.py:211): loss and gradient, one kernel each ----
namespace mmnas {
__global__ void __launch_bounds__(256) bce_logits_sum_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                             float* __restrict__ loss, size_t n) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    s += fmaxf(v, 0.f) - v * t[i] + log1pf(expf(-fabsf(v)));
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1]) + (red[2] + red[3]));
}
__global__ void __launch_bounds__(256) bce_logits_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                             const float* __restrict__ go, float* __restrict__ dx, size_t n) {
  const float g = go ? go[0] : 1.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dx[i] = g * (1.0f / (1.0f + expf(-x[i])) - t[i]);
}
}  // namespace mmnas

extern "C" int mmnas_bce_logits_sum_fwd(const float* logits, const float* target, float* loss, size_t n, void* stream) {
  MMNAS_REQUIRE(logits && target && loss, MMNAS_E_ARG, "bce_logits_sum_fwd: null pointer");
  if (n == 0) return MMNAS_OK;
  const int blocks = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
  MMNAS_LAUNCH(mmnas::bce_logits_sum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, target, loss, n);
  return mmnas::check_launch("bce_logits_sum_fwd");
}
extern "C" int mmnas_bce_logits_bwd(const float* logits, const float* target, const float* go, float* dlogits, size_t n, void* stream) {
  MMNAS_REQUIRE(logits && target && dlogits, MMNAS_E_ARG, "bce_logits_bwd: null pointer");
  if (n == 0) return MMNAS_OK;
  const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
  MMNAS_LAUNCH(mmnas::bce_logits_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, target, go, dlogits, n);
  return mmnas::check_launch("bce_logits_bwd");
}

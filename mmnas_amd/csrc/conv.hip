// 1-D convolutions over the sequence axis of x[B,S,d] (channels last), zero "same" padding:
// SepConv = depthwise k-tap stencil + pointwise GEMM (modules.py:431-462), StdConv = dense k-tap
// Conv1d (modules.py:465-491).  The dense conv runs on the MFMA GEMM with NO window buffer: on the zero-padded
// input (pad_seq below: every sequence followed / surrounded by its padding rows) the im2col matrix
// col[m, t d + c] = xp[m + t, c] is just xp read with a row stride of d instead of k d -- overlapping rows -- which
// mmnas_gemm takes as it is (lda = d, K = k d): forward NT, data gradient NN on the padded output gradient with the
// taps reversed, weight gradient ONE TN product (ops.ConvSeqFn).  The explicit im2col / col2im kernels stay for
// shapes outside the buffer-load path and as the A/B baseline (tools/conv_bench.py); the depthwise stencil is a
// streaming HBM-bound kernel.  k in {3,5,7,11}; these operators are registry-only (in no shipped search space or
// arch/*.json).
#include "common.h"

namespace mmnas {

__global__ void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int S, int d, int k) {
  const int pad = k / 2, d4 = d / 4;
  const size_t n = (size_t)B * S * k * d4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % d4);
    const int t = (int)((i / d4) % k);
    const size_t m = i / ((size_t)d4 * k);
    const int s = (int)(m % S);
    const int ss = s + t - pad;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ss >= 0 && ss < S) v = *reinterpret_cast<const float4*>(x + (m + (ss - s)) * d + 4 * c4);
    *reinterpret_cast<float4*>(col + (m * k + t) * d + 4 * c4) = v;
  }
}

__global__ void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int S, int d, int k) {
  const int pad = k / 2, d4 = d / 4;
  const size_t n = (size_t)B * S * d4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % d4);
    const size_t m = i / d4;
    const int s = (int)(m % S);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < k; ++t) {
      const int so = s - t + pad;  // the output row whose window tap t reads row s
      if (so >= 0 && so < S) {
        const float4 v = *reinterpret_cast<const float4*>(dcol + ((m + (so - s)) * k + t) * d + 4 * c4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    *reinterpret_cast<float4*>(dx + m * d + 4 * c4) = a;
  }
}

// xp[r, :] for r < rows_total: sequence b = r / Sp, position j = r % Sp; x[b, j - front] when 0 <= j - front < S and
// r < B Sp, zero otherwise (the padding rows of every sequence and the slack rows behind the last one)
__global__ void pad_seq_kernel(const float4* __restrict__ x, float4* __restrict__ xp, int B, int S, int d4, int front, int Sp,
                               long rows_total) {
  const size_t n = (size_t)rows_total * d4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const long r = (long)(i / d4);
    const int c4 = (int)(i - (size_t)r * d4);
    const long b = r / Sp;
    const int s = (int)(r - b * Sp) - front;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b < B && s >= 0 && s < S) v = x[((size_t)b * S + s) * d4 + c4];
    xp[i] = v;
  }
}

// y[m,c] = bias[c] + sum_t w[c,t] * x[b, s+t-pad, c]
__global__ void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ bias, float* __restrict__ y, int B, int S, int d, int k) {
  const int pad = k / 2;
  const size_t n = (size_t)B * S * d;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % d);
    const size_t m = i / d;
    const int s = (int)(m % S);
    float a = bias ? bias[c] : 0.f;
    for (int t = 0; t < k; ++t) {
      const int ss = s + t - pad;
      if (ss >= 0 && ss < S) a += w[c * k + t] * x[(m + (ss - s)) * d + c];
    }
    y[i] = a;
  }
}

// dx[m,c] = sum_t w[c,t] * dy[b, s-t+pad, c]
__global__ void dwconv_bwd_x_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                    int B, int S, int d, int k) {
  const int pad = k / 2;
  const size_t n = (size_t)B * S * d;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % d);
    const size_t m = i / d;
    const int s = (int)(m % S);
    float a = 0.f;
    for (int t = 0; t < k; ++t) {
      const int so = s - t + pad;
      if (so >= 0 && so < S) a += w[c * k + t] * dy[(m + (so - s)) * d + c];
    }
    dx[i] = a;
  }
}

// dw[c,t] += sum_m dy[m,c] x[b,s+t-pad,c];  db[c] += sum_m dy[m,c].  thread = channel, grid.y = row split
template <int KMAX>
__global__ void dwconv_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                    float* __restrict__ db, int B, int S, int d, int k, int rows_per_block) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= d) return;
  const int pad = k / 2;
  const long M = (long)B * S;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float acc[KMAX];
#pragma unroll
  for (int t = 0; t < KMAX; ++t) acc[t] = 0.f;
  float accb = 0.f;
  for (long m = r0; m < r1; ++m) {
    const int s = (int)(m % S);
    const float g = dy[m * d + c];
    accb += g;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
      const int ss = s + t - pad;
      if (t < k && ss >= 0 && ss < S) acc[t] += g * x[(m + (ss - s)) * d + c];
    }
  }
#pragma unroll
  for (int t = 0; t < KMAX; ++t)
    if (t < k) atomicAdd(dw + c * k + t, acc[t]);
  if (db) atomicAdd(db + c, accb);
}

static inline int nblocks(size_t n) { size_t b = (n + 255) / 256; return (int)(b < 4096 ? (b ? b : 1) : 4096); }

static int conv_check(const char* who, int B, int S, int d, int k) {
  MMNAS_REQUIRE(B > 0 && S > 0 && d > 0, MMNAS_E_SHAPE, "%s: B=%d S=%d d=%d", who, B, S, d);
  MMNAS_REQUIRE(k >= 1 && k <= 11 && (k & 1), MMNAS_E_SHAPE, "%s: kernel size %d (odd, <= 11)", who, k);
  return MMNAS_OK;
}

}  // namespace mmnas

using namespace mmnas;

extern "C" int mmnas_im2col_seq(const float* x, float* col, int B, int S, int d, int k, void* stream) {
  MMNAS_REQUIRE(x && col, MMNAS_E_ARG, "im2col_seq: null pointer");
  int rc = conv_check("im2col_seq", B, S, d, k);
  if (rc) return rc;
  MMNAS_REQUIRE(d % 4 == 0, MMNAS_E_SHAPE, "im2col_seq: d=%d %% 4", d);
  MMNAS_LAUNCH(im2col_kernel, dim3(nblocks((size_t)B * S * k * d / 4)), dim3(256), 0, (hipStream_t)stream, x, col,
                     B, S, d, k);
  return check_launch("im2col_seq");
}

extern "C" int mmnas_pad_seq(const float* x, float* xp, int B, int S, int d, int front, int Sp, long rows_total, void* stream) {
  MMNAS_REQUIRE(x && xp, MMNAS_E_ARG, "pad_seq: null pointer");
  MMNAS_REQUIRE(B > 0 && S > 0 && d > 0 && d % 4 == 0 && front >= 0 && Sp >= S + front && rows_total >= (long)B * Sp, MMNAS_E_SHAPE,
                "pad_seq: B=%d S=%d d=%d front=%d Sp=%d rows=%ld", B, S, d, front, Sp, rows_total);
  MMNAS_LAUNCH(pad_seq_kernel, dim3(nblocks((size_t)rows_total * d / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
               (float4*)xp, B, S, d / 4, front, Sp, rows_total);
  return check_launch("pad_seq");
}

extern "C" int mmnas_col2im_seq(const float* dcol, float* dx, int B, int S, int d, int k, void* stream) {
  MMNAS_REQUIRE(dcol && dx, MMNAS_E_ARG, "col2im_seq: null pointer");
  int rc = conv_check("col2im_seq", B, S, d, k);
  if (rc) return rc;
  MMNAS_REQUIRE(d % 4 == 0, MMNAS_E_SHAPE, "col2im_seq: d=%d %% 4", d);
  MMNAS_LAUNCH(col2im_kernel, dim3(nblocks((size_t)B * S * d / 4)), dim3(256), 0, (hipStream_t)stream, dcol, dx,
                     B, S, d, k);
  return check_launch("col2im_seq");
}

extern "C" int mmnas_dwconv_seq_fwd(const float* x, const float* w, const float* bias, float* y, int B, int S, int d,
                                    int k, void* stream) {
  MMNAS_REQUIRE(x && w && y, MMNAS_E_ARG, "dwconv_seq_fwd: null pointer");
  int rc = conv_check("dwconv_seq_fwd", B, S, d, k);
  if (rc) return rc;
  MMNAS_LAUNCH(dwconv_fwd_kernel, dim3(nblocks((size_t)B * S * d)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                     y, B, S, d, k);
  return check_launch("dwconv_seq_fwd");
}

extern "C" int mmnas_dwconv_seq_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                                    int B, int S, int d, int k, void* stream) {
  MMNAS_REQUIRE(x && w && dy && dx && dw, MMNAS_E_ARG, "dwconv_seq_bwd: null pointer");
  int rc = conv_check("dwconv_seq_bwd", B, S, d, k);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  MMNAS_LAUNCH(dwconv_bwd_x_kernel, dim3(nblocks((size_t)B * S * d)), dim3(256), 0, st, dy, w, dx, B, S, d, k);
  const long M = (long)B * S;
  int splits = (int)((M + 63) / 64);
  if (splits > 256) splits = 256;
  const int rpb = (int)((M + splits - 1) / splits);
  MMNAS_LAUNCH(dwconv_bwd_w_kernel<11>, dim3(cdiv(d, 256), cdiv(M, rpb)), dim3(256), 0, st, x, dy, dw, db, B, S, d,
                     k, rpb);
  return check_launch("dwconv_seq_bwd");
}

// Row-panel product with the LayerNorm inside:  z = drop(A W^T + bias) + residual,  y = LayerNorm(z)  in ONE launch.
// Replaces the merge / second FFN projection (an mmnas_gemm launch with the dropout + residual epilogue, modules.py:186-187,
// 261-271, 351-362) AND the mmnas_layernorm_fwd launch behind it (modules.py:44-56) -- SURVEY 2.1 "merge projection +
// epilogue: the row spans one workgroup when N = d <= 1024, so the LayerNorm statistics are in-kernel" -- for N = d = 256
// (the supernet's width): 24 launches and 24 dependent launch boundaries per supernet step, and z is not re-read.
//
// MI355X mapping.  A workgroup owns a PANEL of 32 rows x all 256 columns: 8 waves, wave w = columns 32 w .. 32 w + 31
// (one 32x32 accumulator tile, 16 registers), two waves per SIMD.  M = 6400 rows -> 200 workgroups on 256 CUs; the busiest
// CU holds 32 x 256 outputs -- exactly what the busiest CU of the 64^2 tiling holds (400 tiles on 256 CUs = 2 tiles), so
// the product itself costs what it cost, and the epilogue replaces a launch.
//   * K loop = gemm.hip's lean loop: operands global -> registers (16-byte buffer loads, rows behind M read as zero, the
//     K-tile's byte offset in the scalar offset) -> split into three bf16 parts (x = h + m + l exactly) -> LDS image of
//     gemm_split.h -> v_mfma_f32_32x32x16_bf16, six products per fp32 product, smallest cross terms first, fp32
//     accumulate; two register stages in flight ahead of the MFMA block, LDS double buffer, one barrier per K-tile.
//   * the MFMA operands change roles (W fragment as the A operand): accumulators come out transposed, lane = ROW of the
//     panel, registers = 4 runs of 4 consecutive columns.  A row's 32 columns of this wave are in ONE lane pair's
//     registers: the row sum is 15 adds + one cross-half swap; the 8 waves' partial sums meet in LDS (8 x 32 floats).
//   * LayerNorm as rowops.hip computes it (two passes: mean, then the centred sum of squares; Bessel-corrected std, eps
//     added to the std) on the values in registers; z (saved for the backward) and y leave as 16-byte buffer stores.
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "gemm_split.h"

namespace mmnas {

struct GemmLnK {
  const float* A; const float* W; const float* bias; const float* residual; const float* ln_a; const float* ln_b;
  float* z; float* y;
  int M, K, lda, ldb, ldz, ldres, ldy;
  float eps;
  DropCfg drop;
  const unsigned short* Wp;   // PLANES: W as three pre-split bf16 planes [3][256][K] (mmnas_split_planes), else unused
};

constexpr int LN_BM = 32, LN_BN = 256, LN_BK = 32, LN_NT = 512;
constexpr int LN_RSW = 3 * 16 + 4;   // words per LDS row: three runs of 32 bf16 + 16 B pad (gemm_split.h)
#ifndef LN_PF
#define LN_PF 2     // register stages of operand loads in flight ahead of the MFMA block (even; K / 32 must be a multiple)
#endif
#ifndef LN_DSW
#define LN_DSW 0    // LDS stores scheduled behind each MFMA from the third on (0: left to the scheduler)
#endif
#ifndef LN_VPM
#define LN_VPM 10   // vector instructions scheduled behind each MFMA of a half-iteration (tuning)
#endif

// PLANES: the weight operand arrives as three pre-split bf16 planes and goes global -> LDS by LDS-DMA (global_load_lds_dwordx4:
// no register stage, no conversion, no ds_write) -- in this tiling W is 8/9 of a K-tile's operand bytes, so 8/9 of the
// conversion instructions and LDS stores of the loop go away (the 64^2 tiling of gemm.hip splits them evenly between its
// operands; its BDMA form bought 3-10 %).  An LDS-DMA instruction writes 1 KiB lane-linearly: the W image has no row pad,
// row r = 3 runs of 64 B (192 B), the 16-byte chunk j of a run at position j ^ ((r >> 2) & 3) -- applied to the per-lane
// SOURCE address and again on the fragment read (gemm.hip's BDMA image: conflict-free fragment reads).
constexpr int LN_RSWB = 48;
template <bool PLANES>
__global__ void __launch_bounds__(LN_NT, 1) gemm_ln_kernel(const GemmLnK p) {
  __shared__ __attribute__((aligned(16))) float As0[LN_BM * LN_RSW], As1[LN_BM * LN_RSW];
  __shared__ __attribute__((aligned(16))) float Bs0[LN_BN * (PLANES ? LN_RSWB : LN_RSW)], Bs1[LN_BN * (PLANES ? LN_RSWB : LN_RSW)];
  __shared__ __attribute__((aligned(16))) float Aspare[LN_BM * LN_RSW];
  __shared__ float red[2][8][LN_BM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int m0 = blockIdx.x * LN_BM;
  const int M = p.M, nq = p.K / LN_BK;

  // ---- operand offsets (bytes, K-tile 0); ~0u = outside the matrix (the buffer range check answers with zeros) ----
  const bool lda_thread = tid < 256;   // the A tile is 256 16-byte loads: waves 0-3 (wave-uniform)
  unsigned offa, offb[4];
  {
    const int row = kc_row(tid & 255), kq = tid & 7, gr = m0 + row;
    offa = (lda_thread && gr < M) ? (unsigned)(gr * p.lda + 4 * kq) * 4u : ~0u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = tid + LN_NT * i;
    offb[i] = (unsigned)(kc_row(f) * p.ldb + 4 * (f & 7)) * 4u;
  }
  const unsigned bytesa = (unsigned)M * (unsigned)p.lda * 4u, bytesb = (unsigned)LN_BN * (unsigned)p.ldb * 4u;

  // PLANES: piece pc = 6 wave + i of the 48 KiB W image: this lane's 16 bytes sit at byte 1024 pc + 16 lane
  size_t dma_off[6];
  if (PLANES) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int o = (wave * 6 + i) * 1024 + lane * 16;
      const int r = o / 192, w = o - r * 192, c = w >> 6, j = ((w & 63) >> 4) ^ ((r >> 2) & 3);
      dma_off[i] = ((size_t)c * (size_t)LN_BN * (size_t)p.ldb + (size_t)r * (size_t)p.ldb + 8 * j) * 2u;
    }
  }
  float4 rA[LN_PF], rB[LN_PF][4];
  // W tile kt -> LDS buffer `buf` by LDS-DMA (PLANES); the destination of a piece is wave-uniform
  auto dma = [&](int kt, bool live, int buf) __attribute__((always_inline)) {
    if (!PLANES || !live) return;
    const char* src = reinterpret_cast<const char*>(p.Wp) + (size_t)kt * 64u;
    float* const bd = buf ? Bs1 : Bs0;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int i = 0; i < 6; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + dma_off[i]),
                                       (__attribute__((address_space(3))) void*)(bd + (wv * 6 + i) * 256), 16, 0, 0);
  };
  auto gload = [&](int kt, bool live, const int st) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, live ? bytesa : 0u, 0x00020000);
    const unsigned ko = (unsigned)kt * (LN_BK * 4u);
    rA[st] = buf_load4(a_rs, offa, ko);
    if (PLANES) return;
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, live ? bytesb : 0u, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) rB[st][i] = buf_load4(b_rs, offb[i], ko);
  };
  auto lstore = [&](int buf, const int st) __attribute__((always_inline)) {
    unsigned* ua = reinterpret_cast<unsigned*>(buf ? As1 : As0);
    unsigned* ub = reinterpret_cast<unsigned*>(buf ? Bs1 : Bs0);
    // (waves 4-7 hold no part of the A tile: they store their zeros into a spare image instead of branching around the store --
    //  a branch would end the scheduling region that interleaves these stores with the MFMAs)
    split_store_kc<LN_BM, true, 3>(lda_thread ? ua : reinterpret_cast<unsigned*>(Aspare), rA[st], tid & 255);
    if (PLANES) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) split_store_kc<LN_BN, true, 3>(ub, rB[st][i], tid + LN_NT * i);
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  auto mfma_block = [&](int buf) __attribute__((always_inline)) {
    const float* a = buf ? As1 : As0;
    const float* b = buf ? Bs1 : Bs0;
    const int brow = 32 * wave + l31;
#pragma unroll
    for (int s = 0; s < LN_BK / 16; ++s) {
      bf16x8 af[3], bf[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        af[c] = *reinterpret_cast<const bf16x8*>(a + l31 * LN_RSW + c * 16 + swz(s * 8 + hh * 4, l31));
        bf[c] = PLANES ? *reinterpret_cast<const bf16x8*>(b + brow * LN_RSWB + c * 16 + 4 * ((s * 2 + hh) ^ ((brow >> 2) & 3)))
                       : *reinterpret_cast<const bf16x8*>(b + brow * LN_RSW + c * 16 + swz(s * 8 + hh * 4, brow));
      }
      // smallest cross terms first; part c of A with part e of W is kept while c + e < 3 (transposed: W as the A operand)
#pragma unroll
      for (int o = 2; o >= 0; --o)
#pragma unroll
        for (int c = 0; c <= o; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[o - c], af[c], acc, 0, 0, 0);
    }
  };

  // ---- K loop: two register stages ahead of the MFMA block, unrolled by two so the stages are static ----
  // ---- epilogue operands, fetched HERE (the residual rows, bias, LayerNorm parameters: 16 16-byte loads per lane that the
  //      epilogue would otherwise wait a full memory round trip for behind the last K-tile): this lane holds row m0 + l31,
  //      register 4 g + e = column 32 wave + 8 g + 4 hh + e ----
  const int row = m0 + l31;
  const int cb = 32 * wave + 4 * hh;
  const bool rok = row < M;
  const bool has_res = p.residual != nullptr;
  const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, has_res ? (unsigned)M * (unsigned)p.ldres * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t z_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.z, 0, p.z ? (unsigned)M * (unsigned)p.ldz * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, (unsigned)M * (unsigned)p.ldy * 4u, 0x00020000);
  float4 resv[4], biasv[4], lav[4], lbv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int col = cb + 8 * g;
    resv[g] = buf_load4(r_rs, rok ? (unsigned)(row * p.ldres + col) * 4u : ~0u);
    biasv[g] = p.bias ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    lav[g] = *reinterpret_cast<const float4*>(p.ln_a + col);
    lbv[g] = *reinterpret_cast<const float4*>(p.ln_b + col);
  }
  // (nq is even -- host check -- so a half-iteration is ONE scheduling region: the 12 MFMAs of K-tile t and the conversion +
  //  LDS stores of K-tile t + 1 are issued interleaved, one MFMA, ten vector instructions, one store.  Left in program order
  //  every wave of the workgroup multiplied, then every wave converted, barrier: the matrix pipe idled through the conversion
  //  and the vector pipe through the products -- 2400 cycles per K-tile measured against 770 of MFMA time.)
#define LN_INTERLEAVE()                                      \
  do {                                                       \
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);       \
    _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {      \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
      __builtin_amdgcn_sched_group_barrier(0x002, LN_VPM, 0); \
      if (LN_DSW && i_ >= 2) __builtin_amdgcn_sched_group_barrier(0x200, LN_DSW, 0); \
      if (i_ == 2) __builtin_amdgcn_sched_group_barrier(0x100, 6, 0); \
    }                                                        \
  } while (0)
  // (PLANES: the W tile of K-tile t + 1 travels global -> LDS buffer (t + 1) & 1 while K-tile t is multiplied; the barrier that
  //  ends a half-iteration follows a vmcnt(0): every lane's pieces have landed before anybody reads the image)
#define LN_DMA_LANDED() do { if (PLANES) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
  dma(0, true, 0);
#pragma unroll
  for (int st = 0; st < LN_PF; ++st) gload(st, st < nq, st);
  lstore(0, 0);
  LN_DMA_LANDED();
  __syncthreads();
  // LN_PF register stages: the loads of K-tile q + LN_PF are issued when K-tile q is multiplied (stage q % LN_PF went to LDS
  // one step earlier); unrolled by LN_PF (even) so that stages and LDS buffers are static
  for (int t = 0; t < nq; t += LN_PF) {
#pragma unroll
    for (int st = 0; st < LN_PF; ++st) {
      const int q = t + st;
      dma(q + 1, q + 1 < nq, (st + 1) & 1);
      gload(q + LN_PF, q + LN_PF < nq, st);
      mfma_block(st & 1);
      lstore((st + 1) & 1, (st + 1) % LN_PF);    // (behind the last K-tile: zeros into a buffer nobody reads again)
      if (!PLANES) LN_INTERLEAVE();
      LN_DMA_LANDED();
      __syncthreads();
    }
  }

  // ---- epilogue (its operands were fetched in front of the K loop) ----
  const bool has_drop = p.drop.thresh != 0;
  const uint32_t dpre = drop_pre(p.drop, (uint32_t)row * (uint32_t)LN_BN + (uint32_t)cb);
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float r4[4] = {resv[g].x, resv[g].y, resv[g].z, resv[g].w};
    const float b4[4] = {biasv[g].x, biasv[g].y, biasv[g].z, biasv[g].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float val = acc[4 * g + e] + b4[e];
      if (has_drop) val *= drop_mult_pre(p.drop, dpre + (uint32_t)(8 * g + e) * DROP_G);
      val += r4[e];
      v[4 * g + e] = val;
      s += val;
    }
    u32x4 w4;
    w4.x = __float_as_uint(v[4 * g]); w4.y = __float_as_uint(v[4 * g + 1]); w4.z = __float_as_uint(v[4 * g + 2]); w4.w = __float_as_uint(v[4 * g + 3]);
    __builtin_amdgcn_raw_buffer_store_b128(w4, z_rs, rok ? (unsigned)(row * p.ldz + cb + 8 * g) * 4u : ~0u, 0, 0);
  }
  // row statistics over the 256 columns: lane pair (l31, hh) -> wave partial -> the 8 waves through LDS
  s += __shfl_xor(s, 32, 64);
  if (hh == 0) red[0][wave][l31] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) mean += red[0][w][l31];
  mean *= 1.0f / (float)LN_BN;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] -= mean; ss += v[i] * v[i]; }
  ss += __shfl_xor(ss, 32, 64);
  if (hh == 0) red[1][wave][l31] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) tot += red[1][w][l31];
  const float sd = sqrtf(tot / (float)(LN_BN - 1));
  const float inv = 1.0f / (sd + p.eps);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    u32x4 w4;
    w4.x = __float_as_uint(lav[g].x * v[4 * g] * inv + lbv[g].x);
    w4.y = __float_as_uint(lav[g].y * v[4 * g + 1] * inv + lbv[g].y);
    w4.z = __float_as_uint(lav[g].z * v[4 * g + 2] * inv + lbv[g].z);
    w4.w = __float_as_uint(lav[g].w * v[4 * g + 3] * inv + lbv[g].w);
    __builtin_amdgcn_raw_buffer_store_b128(w4, y_rs, rok ? (unsigned)(row * p.ldy + cb + 8 * g) * 4u : ~0u, 0, 0);
  }
}

static int g_gemm_ln = -1;   // -1: not read yet (MMNAS_GEMM_LN, default 0: opt-in); mmnas_set_gemm_ln() overrides
static int g_gemm_ln_minm = 2048;
static int g_gemm_ln_maxk = 256;
static bool gemm_ln_on() {
  if (g_gemm_ln < 0) {
    const char* e = getenv("MMNAS_GEMM_LN");
    // OPT-IN (round 6, profiles/r06_ab.txt): on one box, alternating runs, the supernet step is 4.556 / 4.549 / 4.564 ms with
    // the panel kernel and 4.561 / 4.551 / 4.554 ms without -- neutral (the launch it removes is paid back by a product that
    // runs 3.7 us longer inside the step than the 64^2 tiling with 2-3 workgroups per CU); the ragged stream loses 0.03 ms
    // (3900 valid rows = 122 panels on 256 CUs).  VERDICT r5 item 1(b): "if it measures neutral, keep it opt-in".
    g_gemm_ln = (e && e[0] == '1') ? 1 : 0;
    const char* m = getenv("MMNAS_GEMM_LN_MINM");   // products with fewer rows keep the two-launch form (tuning)
    g_gemm_ln_minm = m && m[0] ? atoi(m) : 2048;
    const char* k = getenv("MMNAS_GEMM_LN_MAXK");
    g_gemm_ln_maxk = k && k[0] ? atoi(k) : 256;    // (measured, tools/gemm_ln_bench.py: -3.4 us at K = 256, +-0.5 at 512, +1.5 at 1024)    // (896 rows = 28 panels on 256 CUs: the tiled product + LayerNorm win there)
  }
  return g_gemm_ln != 0;
}

// The product of *d (the merge / last FFN projection as the operators describe it) qualifies for the panel kernel:
// NT, one group, one K-segment, N = 256, K a multiple of 32, 16-byte aligned operands, epilogue = bias / dropout / residual.
bool gemm_ln_applies(const mmnas_gemm_desc* d) {
  if (!d || !gemm_ln_on()) return false;
  if (d->layout != MMNAS_GEMM_NT || d->ngroups != 1 || d->nseg != 1 || d->N != LN_BN) return false;
  if (d->K < LN_PF * LN_BK || d->K % (LN_PF * LN_BK) != 0 || d->lda % 4 != 0 || d->ldb % 4 != 0 || d->ldc % 4 != 0) return false;
  if (d->relu || d->accumulate || d->split_k > 1 || d->alpha != 1.f) return false;
  if (d->b_planes && d->ldb % 8 != 0) return false;
  const mmnas_gemm_group& g = d->g[0];
  if (g.gate || g.colsum || g.M < 1 || g.M < g_gemm_ln_minm || d->K > g_gemm_ln_maxk) return false;
  if (g.residual && d->ldres % 4 != 0) return false;
  if (((uintptr_t)g.A[0] | (uintptr_t)g.B[0] | (uintptr_t)g.C | (uintptr_t)g.residual | (uintptr_t)g.bias) & 15) return false;
  if ((double)g.M * d->lda * 4.0 >= 4294967296.0 || (double)g.M * d->ldc * 4.0 >= 4294967296.0) return false;
  return true;
}

int gemm_ln(const mmnas_gemm_desc* d, const float* ln_a, const float* ln_b, float* y, int ldy, float eps, hipStream_t st) {
  MMNAS_REQUIRE(gemm_ln_applies(d), MMNAS_E_ARG, "gemm_ln: the product does not qualify for the row-panel kernel");
  MMNAS_REQUIRE(ln_a && ln_b && y && ldy % 4 == 0 && ldy >= LN_BN, MMNAS_E_ARG, "gemm_ln: LayerNorm operands");
  MMNAS_REQUIRE((((uintptr_t)ln_a | (uintptr_t)ln_b | (uintptr_t)y) & 15) == 0, MMNAS_E_ARG, "gemm_ln: 16-byte alignment");
  const mmnas_gemm_group& g = d->g[0];
  GemmLnK k;
  memset(&k, 0, sizeof(k));
  k.A = g.A[0]; k.W = g.B[0]; k.bias = g.bias; k.residual = g.residual; k.ln_a = ln_a; k.ln_b = ln_b;
  k.z = g.C; k.y = y;
  k.M = g.M; k.K = d->K; k.lda = d->lda; k.ldb = d->ldb; k.ldz = d->ldc; k.ldres = d->ldres; k.ldy = ldy;
  k.eps = eps;
  k.drop = make_drop(d->drop_p, g.drop_seed ? g.drop_seed : d->drop_seed, d->drop_site);
  const double M = g.M;
  ProfScope ps(MMNAS_K_GEMM, 2.0 * M * LN_BN * d->K, 4.0 * (M * d->K + (double)LN_BN * d->K + (g.residual ? 3.0 : 2.0) * M * LN_BN), st, "gemm_ln");
  if (d->b_planes) {
    k.Wp = (const unsigned short*)g.B[0]; k.W = nullptr;
    MMNAS_LAUNCH(gemm_ln_kernel<true>, dim3(cdiv(g.M, LN_BM)), dim3(LN_NT), 0, st, k);
  } else {
    MMNAS_LAUNCH(gemm_ln_kernel<false>, dim3(cdiv(g.M, LN_BM)), dim3(LN_NT), 0, st, k);
  }
  return check_launch("gemm_ln");
}

}  // namespace mmnas

using namespace mmnas;

extern "C" int mmnas_set_gemm_ln(int on) {
  gemm_ln_on();
  const int old = g_gemm_ln;
  g_gemm_ln = on ? 1 : 0;
  return old;
}

extern "C" int mmnas_gemm_ln(const mmnas_gemm_desc* d, const float* ln_a, const float* ln_b, float* y, float eps, void* stream) {
  MMNAS_REQUIRE(d && ln_a && ln_b && y, MMNAS_E_ARG, "mmnas_gemm_ln: null pointer");
  MMNAS_REQUIRE(d->ngroups == 1, MMNAS_E_ARG, "mmnas_gemm_ln: one group");
  if (gemm_ln_applies(d)) return gemm_ln(d, ln_a, ln_b, y, d->N, eps, (hipStream_t)stream);
  MMNAS_REQUIRE(d->g[0].C && d->ldc == d->N, MMNAS_E_ARG, "mmnas_gemm_ln: the two-launch form needs z with ldc = N");
  int rc = mmnas_gemm(d, stream);
  if (rc) return rc;
  return mmnas_layernorm_fwd(d->g[0].C, ln_a, ln_b, y, d->g[0].M, d->N, eps, stream);
}

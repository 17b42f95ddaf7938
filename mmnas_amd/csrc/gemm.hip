// Grouped fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD on gfx950) with a
// fused epilogue.  Replaces the ATen mm/addmm behind every nn.Linear of the reference operators
// (modules.py:18,38,172-175,219) and their autograd backward.
//
// Design (MI355X):
//   * workgroup = 256 threads = 4 waves in a 2x2 arrangement; tile BM x BN x 32 with
//     BM = BN = 128 (each wave 64x64 = 2x2 MFMA tiles, 64 accumulator VGPRs) or 64 (one MFMA tile
//     per wave) for problems that would not fill 256 CUs with 128^2 tiles.
//   * operands are staged global -> registers -> LDS (16-B vector loads issued one K-tile ahead,
//     written to the other LDS buffer after the MFMA block: one barrier per K-tile).
//   * FAST path (every shape of the VQA/VGD/ITM workloads): operand loads are bounds-checked
//     buffer loads (buffer_load_dwordx4 through a per-operand descriptor): rows beyond M/N get an
//     out-of-range offset and read as zero, so the K loop has no branches and its 8 loads issue
//     back to back.  Shapes with K % 32 != 0, unaligned bases or leading dimensions take the
//     generic path (guarded scalar loads) -- same tiles, same epilogue.
//   * K-contiguous operands sit in LDS as [row][36] (pad 4: ds_read_b128 fragment reads are
//     conflict-free because 36/4 = 9 is odd); row-contiguous operands (B of NN, A and B of TN) sit
//     as [k][rows] and are read with conflict-free ds_read_b32.  The reduction index inside an
//     8-wide K group is permuted identically for A and B (lane half hh takes k = 8s+4hh+t), which
//     is legal because both operands see the same permutation.
//   * blockIdx.x -> tile mapping is XCD-aware: the 8 XCDs (private 4 MiB L2 each) get contiguous
//     runs of tiles, so an XCD re-reads only its own A row-panels and the (small) weight matrix.
#include "common.h"

namespace mmnas {

struct GemmGroupK {
  int M;
  const float* A[3];
  const float* B[3];
  float* C;
  const float* bias;
  const float* residual;
  const float* gate;
};

struct GemmK {
  int ngroups, nseg, N, K;
  int lda, ldb, ldc, ldres, ldgate;
  int relu, split_k, k_per_split, tiles_n, ntiles;
  int avec, bvec;  // generic path: 16-byte vector loads legal for the A / B operand
  float alpha, gate_scale;
  DropCfg drop;
  GemmGroupK g[3];
};

constexpr int BK = 32;
constexpr int LDK = 36;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
  float4 f;
  f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
  return f;
}

template <int BM, int BN, bool AKC, bool BKC, bool FAST>
__global__ void __launch_bounds__(256) gemm_kernel(const GemmK p) {
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  constexpr int A_SZ = BM * LDK, B_SZ = BN * LDK;  // >= BK*BM for the [k][row] form
  constexpr int NA = BM / 32, NB = BN / 32;        // float4 loads per thread per tile
  __shared__ __attribute__((aligned(16))) float As[2 * A_SZ];
  __shared__ __attribute__((aligned(16))) float Bs[2 * B_SZ];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- which problem / tile ----
  const int grp = blockIdx.z;
  const GemmGroupK& G = p.g[grp];
  const int Mg = G.M;
  int tile, split;
  {  // XCD-aware bijective remap (blocks b and b+8 share an XCD) over the (K-slice, tile) space, slice
     // major: an XCD owns a contiguous run of tiles of as few K-slices as possible, so with split-K
     // each XCD's L2 streams only its own slices of A and B instead of all of both (8x fabric re-reads)
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, in = bid >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + in;
    split = v / p.ntiles;
    tile = v - split * p.ntiles;
  }
  const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (m0 >= Mg) return;  // uniform for the whole workgroup

  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int ntk = (kend - kbeg + BK - 1) / BK;
  const int T = ntk * p.nseg;
  if (T <= 0) return;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[NA], rb[NB];

  // FAST path: per-thread byte offsets of its loads inside the operand (k0 = 0), ~0u when the row is
  // outside the matrix (the buffer range check then returns zeros)
  unsigned offa[NA], offb[NB];
  unsigned stepa = 0, stepb = 0;  // bytes per K-tile
  unsigned bytesa = 0, bytesb = 0;
  if (FAST) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + 256 * i;
      if (AKC) {
        const int row = f >> 3, kq = f & 7, gr = m0 + row;
        offa[i] = gr < Mg ? (unsigned)(gr * p.lda + kbeg + 4 * kq) * 4u : ~0u;
      } else {
        const int k = f / (BM / 4), rq = f - k * (BM / 4), gr = m0 + 4 * rq;
        offa[i] = gr < Mg ? (unsigned)((kbeg + k) * p.lda + gr) * 4u : ~0u;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + 256 * i;
      if (BKC) {
        const int row = f >> 3, kq = f & 7, gr = n0 + row;
        offb[i] = gr < p.N ? (unsigned)(gr * p.ldb + kbeg + 4 * kq) * 4u : ~0u;
      } else {
        const int k = f / (BN / 4), rq = f - k * (BN / 4), gr = n0 + 4 * rq;
        offb[i] = gr < p.N ? (unsigned)((kbeg + k) * p.ldb + gr) * 4u : ~0u;
      }
    }
    stepa = AKC ? BK * 4u : (unsigned)p.lda * BK * 4u;
    stepb = BKC ? BK * 4u : (unsigned)p.ldb * BK * 4u;
    bytesa = (unsigned)(AKC ? Mg : p.K) * (unsigned)p.lda * 4u;
    bytesb = (unsigned)(BKC ? p.N : p.K) * (unsigned)p.ldb * 4u;
  }

  auto gload = [&](int t) {
    const int seg = t / ntk;
    const int kt = t - seg * ntk;
    const float* __restrict__ Ap = G.A[seg];
    const float* __restrict__ Bp = G.B[seg];
    if (FAST) {
      const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, bytesa, 0x00020000);
      const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, bytesb, 0x00020000);
      const unsigned ka = (unsigned)kt * stepa, kb = (unsigned)kt * stepb;
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[i] = buf_load4(ra_src, offa[i] == ~0u ? ~0u : offa[i] + ka);
#pragma unroll
      for (int i = 0; i < NB; ++i) rb[i] = buf_load4(rb_src, offb[i] == ~0u ? ~0u : offb[i] + kb);
      return;
    }
    const int k0 = kbeg + kt * BK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (AKC) {
        const int row = f >> 3, kq = f & 7;
        const int gr = m0 + row, gk = k0 + 4 * kq;
        if (gr < Mg && gk < kend) {
          const float* src = Ap + (size_t)gr * p.lda + gk;
          if (p.avec) v = *reinterpret_cast<const float4*>(src);
          else {
            v.x = src[0];
            if (gk + 1 < kend) v.y = src[1];
            if (gk + 2 < kend) v.z = src[2];
            if (gk + 3 < kend) v.w = src[3];
          }
        }
      } else {
        const int k = f / (BM / 4), rq = f - k * (BM / 4);
        const int gk = k0 + k, gr = m0 + 4 * rq;
        if (gk < kend && gr < Mg) {
          const float* src = Ap + (size_t)gk * p.lda + gr;
          if (p.avec) v = *reinterpret_cast<const float4*>(src);
          else {
            v.x = src[0];
            if (gr + 1 < Mg) v.y = src[1];
            if (gr + 2 < Mg) v.z = src[2];
            if (gr + 3 < Mg) v.w = src[3];
          }
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + 256 * i;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (BKC) {
        const int row = f >> 3, kq = f & 7;
        const int gr = n0 + row, gk = k0 + 4 * kq;
        if (gr < p.N && gk < kend) {
          const float* src = Bp + (size_t)gr * p.ldb + gk;
          if (p.bvec) v = *reinterpret_cast<const float4*>(src);
          else {
            v.x = src[0];
            if (gk + 1 < kend) v.y = src[1];
            if (gk + 2 < kend) v.z = src[2];
            if (gk + 3 < kend) v.w = src[3];
          }
        }
      } else {
        const int k = f / (BN / 4), rq = f - k * (BN / 4);
        const int gk = k0 + k, gr = n0 + 4 * rq;
        if (gk < kend && gr < p.N) {
          const float* src = Bp + (size_t)gk * p.ldb + gr;
          if (p.bvec) v = *reinterpret_cast<const float4*>(src);
          else {
            v.x = src[0];
            if (gr + 1 < p.N) v.y = src[1];
            if (gr + 2 < p.N) v.z = src[2];
            if (gr + 3 < p.N) v.w = src[3];
          }
        }
      }
      rb[i] = v;
    }
  };

  auto lstore = [&](int buf) {
    float* a = As + buf * A_SZ;
    float* b = Bs + buf * B_SZ;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int f = tid + 256 * i;
      if (AKC) {
        const int row = f >> 3, kq = f & 7;
        *reinterpret_cast<float4*>(a + row * LDK + 4 * kq) = ra[i];
      } else {
        const int k = f / (BM / 4), rq = f - k * (BM / 4);
        *reinterpret_cast<float4*>(a + k * BM + 4 * rq) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int f = tid + 256 * i;
      if (BKC) {
        const int row = f >> 3, kq = f & 7;
        *reinterpret_cast<float4*>(b + row * LDK + 4 * kq) = rb[i];
      } else {
        const int k = f / (BN / 4), rq = f - k * (BN / 4);
        *reinterpret_cast<float4*>(b + k * BN + 4 * rq) = rb[i];
      }
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const int buf = t & 1;
    if (t + 1 < T) gload(t + 1);  // in flight during the MFMA block
    const float* a = As + buf * A_SZ;
    const float* b = Bs + buf * B_SZ;
#pragma unroll
    for (int s = 0; s < BK / 8; ++s) {
      float af[TM][4], bf[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WM + i * 32 + l31;
        if (AKC) {
          const float4 v = *reinterpret_cast<const float4*>(a + row * LDK + 8 * s + 4 * hh);
          af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) af[i][u] = a[(8 * s + 4 * hh + u) * BM + row];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = wn * WN + j * 32 + l31;
        if (BKC) {
          const float4 v = *reinterpret_cast<const float4*>(b + col * LDK + 8 * s + 4 * hh);
          bf[j][0] = v.x; bf[j][1] = v.y; bf[j][2] = v.z; bf[j][3] = v.w;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) bf[j][u] = b[(8 * s + 4 * hh + u) * BN + col];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(af[i][u], bf[j][u], acc[i][j]);
    }
    if (t + 1 < T) lstore(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----
  // All conditions on kernel arguments are wave-uniform and hoisted out of the element loops; the
  // residual / gate operands of a 32x32 sub-tile are fetched as one batch of 16 independent loads
  // (clamped row index instead of a branch) before any arithmetic, so their latency overlaps.
  const bool atomic = p.split_k > 1;
  const bool has_res = G.residual != nullptr, has_gate = G.gate != nullptr;
  const bool has_drop = p.drop.thresh != 0, has_relu = p.relu != 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WN + j * 32 + l31;
      const bool cok = col < p.N;
      const int colc = cok ? col : p.N - 1;
      const int rbase = m0 + wm * WM + i * 32 + 4 * hh;
      if (atomic) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (cok && row < Mg) atomicAdd(G.C + (size_t)row * p.ldc + col, acc[i][j][r] * p.alpha);
        }
        continue;
      }
      float resv[16], gatev[16];
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(rbase + (r & 3) + 8 * (r >> 2), Mg - 1);
          resv[r] = G.residual[(size_t)row * p.ldres + colc];
        }
      }
      if (has_gate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(rbase + (r & 3) + 8 * (r >> 2), Mg - 1);
          gatev[r] = G.gate[(size_t)row * p.ldgate + colc];
        }
      }
      const float bv = G.bias ? G.bias[colc] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        float v = acc[i][j][r] * p.alpha + bv;
        if (has_relu) v = fmaxf(v, 0.f);
        if (has_drop) v *= drop_mult(p.drop, (uint32_t)row * (uint32_t)p.N + (uint32_t)col);
        if (has_gate) v = gatev[r] > 0.f ? v * p.gate_scale : 0.f;
        if (has_res) v += resv[r];
        if (cok && row < Mg) G.C[(size_t)row * p.ldc + col] = v;
      }
    }
  }
}

template <int BM, int BN, bool FAST>
static int launch(GemmK& k, int layout, int maxM, hipStream_t st) {
  const int tiles_m = cdiv(maxM, BM);
  k.ntiles = tiles_m * k.tiles_n;
  dim3 grid(k.ntiles * k.split_k, 1, k.ngroups), block(256);
  switch (layout) {
    case MMNAS_GEMM_NT: hipLaunchKernelGGL((gemm_kernel<BM, BN, true, true, FAST>), grid, block, 0, st, k); break;
    case MMNAS_GEMM_NN: hipLaunchKernelGGL((gemm_kernel<BM, BN, true, false, FAST>), grid, block, 0, st, k); break;
    default: hipLaunchKernelGGL((gemm_kernel<BM, BN, false, false, FAST>), grid, block, 0, st, k); break;
  }
  return check_launch("gemm");
}

}  // namespace mmnas

using namespace mmnas;

extern "C" int mmnas_gemm(const mmnas_gemm_desc* d, void* stream) {
  MMNAS_REQUIRE(d != nullptr, MMNAS_E_ARG, "mmnas_gemm: null descriptor");
  MMNAS_REQUIRE(d->ngroups >= 1 && d->ngroups <= 3 && d->nseg >= 1 && d->nseg <= 3, MMNAS_E_ARG,
                "mmnas_gemm: ngroups=%d nseg=%d out of range", d->ngroups, d->nseg);
  MMNAS_REQUIRE(d->layout >= 0 && d->layout <= 2, MMNAS_E_ARG, "mmnas_gemm: bad layout %d", d->layout);
  MMNAS_REQUIRE(d->N > 0 && d->K > 0, MMNAS_E_SHAPE, "mmnas_gemm: N=%d K=%d", d->N, d->K);
  const bool tn = d->layout == MMNAS_GEMM_TN;
  int split = d->split_k < 1 ? 1 : d->split_k;
  MMNAS_REQUIRE(split == 1 || tn, MMNAS_E_ARG, "mmnas_gemm: split_k only for the TN layout");

  GemmK k;
  k.ngroups = d->ngroups; k.nseg = d->nseg; k.N = d->N; k.K = d->K;
  k.lda = d->lda; k.ldb = d->ldb; k.ldc = d->ldc; k.ldres = d->ldres; k.ldgate = d->ldgate;
  k.relu = d->relu; k.alpha = d->alpha; k.gate_scale = d->gate_scale;
  k.drop = make_drop(d->drop_p, d->drop_seed, d->drop_site);
  // vector (16-byte) operand loads need aligned bases, leading dims and extents; otherwise the
  // kernel falls back to guarded scalar loads (odd shapes such as the 3129-way answer projection)
  const bool akc = d->layout != MMNAS_GEMM_TN, bkc = d->layout == MMNAS_GEMM_NT;
  int avec = (d->lda % 4 == 0) && (akc ? d->K % 4 == 0 : 1);
  int bvec = (d->ldb % 4 == 0) && (bkc ? d->K % 4 == 0 : d->N % 4 == 0);
  int maxM = 0;
  double maxbytes = 0;
  for (int g = 0; g < d->ngroups; ++g) {
    const mmnas_gemm_group& s = d->g[g];
    if (!akc && s.M % 4 != 0) avec = 0;
    MMNAS_REQUIRE(s.M > 0 && s.C, MMNAS_E_ARG, "mmnas_gemm: group %d M=%d C=%p", g, s.M, (void*)s.C);
    k.g[g].M = s.M; k.g[g].C = s.C; k.g[g].bias = s.bias; k.g[g].residual = s.residual; k.g[g].gate = s.gate;
    for (int i = 0; i < 3; ++i) {
      k.g[g].A[i] = s.A[i]; k.g[g].B[i] = s.B[i];
      if (i < d->nseg) {
        MMNAS_REQUIRE(s.A[i] && s.B[i], MMNAS_E_ARG, "mmnas_gemm: group %d segment %d null operand", g, i);
        if (((uintptr_t)s.A[i] & 15) != 0) avec = 0;
        if (((uintptr_t)s.B[i] & 15) != 0) bvec = 0;
      }
    }
    if (s.M > maxM) maxM = s.M;
    const double ab = 4.0 * (akc ? (double)(s.M + 128) : (double)d->K) * d->lda;
    if (ab > maxbytes) maxbytes = ab;
  }
  const double bb = 4.0 * (bkc ? (double)(d->N + 128) : (double)d->K) * d->ldb;
  if (bb > maxbytes) maxbytes = bb;
  k.avec = avec; k.bvec = bvec;
  // tile choice (measured, tools/gemm_bench.py): 64^2 tiles win on every shape of the VQA workloads
  // (M = 6400, N,K <= 2048: 75-115 TF/s vs 69-108 with 128^2) because they give 4x the workgroups
  // to balance over 256 CUs; 128^2 (half the LDS/L2 traffic per flop) only pays once there are
  // >= 4 full waves of them (4096^3: 129-134 TF/s vs 117-125)
  const long t128 = (long)cdiv(maxM, 128) * cdiv(d->N, 128) * d->ngroups;
  bool big = t128 * (tn ? split : 1) >= 1024;
  {  // tuning / test knob: MMNAS_GEMM_TILE=64|128 forces the tile shape
    const char* e = getenv("MMNAS_GEMM_TILE");
    const int force = e ? atoi(e) : 0;
    if (force == 128) big = true;
    if (force == 64) big = false;
  }
  // K slices are multiples of the K tile so every slice starts on a tile boundary
  const int kps = ((cdiv(d->K, split) + BK - 1) / BK) * BK;
  split = cdiv(d->K, kps);
  k.split_k = split; k.k_per_split = kps;
  if (split > 1)
    MMNAS_REQUIRE(!d->relu && d->drop_p == 0.f, MMNAS_E_ARG, "mmnas_gemm: no relu/dropout epilogue with split_k");
  // branch-free buffer-load path: aligned vector loads, K a multiple of the K tile, 32-bit byte offsets
  bool fast = avec && bvec && (d->K % BK == 0) && maxbytes < 4.0e9;
  if (getenv("MMNAS_GEMM_GENERIC")) fast = false;
  hipStream_t st = (hipStream_t)stream;
  double sumM = 0;
  for (int g = 0; g < d->ngroups; ++g) sumM += d->g[g].M;
  // algorithmic work: 2*M*N*K flops per product; minimum traffic = operands once + result once
  ProfScope ps(MMNAS_K_GEMM, 2.0 * sumM * d->N * d->K * d->nseg,
               4.0 * (sumM * d->K * d->nseg + (double)d->N * d->K * d->nseg * d->ngroups + sumM * d->N), st);
  if (big) {
    k.tiles_n = cdiv(d->N, 128);
    return fast ? launch<128, 128, true>(k, d->layout, maxM, st) : launch<128, 128, false>(k, d->layout, maxM, st);
  }
  k.tiles_n = cdiv(d->N, 64);
  return fast ? launch<64, 64, true>(k, d->layout, maxM, st) : launch<64, 64, false>(k, d->layout, maxM, st);
}

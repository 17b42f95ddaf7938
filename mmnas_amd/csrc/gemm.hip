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
//   * scheduling is stream-K: the (output tile, K-tile) pairs of the whole (grouped) problem form one
//     linear sequence of U work units, tile-major; workgroup v computes units [v*P, (v+1)*P).  The
//     workloads' shapes never fill 256 CUs evenly with whole tiles (M = 6400 -> 50 row tiles, 800
//     tiles of 64^2 on 1024 slots, ...), so tiles are cut along K where the balance needs it.  A
//     workgroup that computed only part of a tile's reduction stores its partial accumulators to a
//     workspace slot and bumps the tile's arrival counter; the LAST arriver sums the partials of all
//     contributors in contributor order (bitwise reproducible, no float atomics, nobody waits) and
//     runs the epilogue.  Weight gradients (TN, reduction over the 6400 rows) use the same mechanism
//     instead of split-K atomics.
//   * workgroup -> unit-range mapping is XCD-aware: the 8 XCDs (private 4 MiB L2 each) get contiguous
//     runs of units, so an XCD re-reads only its own A row-panels and the (small) weight matrix.
#include <stddef.h>
#include <string.h>
#include <map>
#include <mutex>
#include <utility>
#include <algorithm>
#include <climits>
#include "common.h"
#include "gemm_split.h"

namespace mmnas {

struct GemmGroupK {
  int M;
  int tile0;  // first linear tile of this group
  const float* A[3];
  const float* B[3];
  float* C;
  const float* bias;
  const float* residual;
  const float* gate;
  float* colsum;
  uint32_t seed_lo, site_key;   // this group's dropout stream (DropCfg fields; the launch's when the group names no seed)
};

constexpr int MAXG = MMNAS_GEMM_MAX_GROUPS;

struct GemmK {
  int ngroups, nseg, N, K;
  int lda, ldb, ldc, ldres, ldgate;
  int relu, accumulate;
  int tiles_n, ntiles;  // tiles along N; tiles of all groups
  int xcd_remap;        // 0: workgroup v = blockIdx.x (tuning experiments)
  int gm;               // tile order inside a group: blocks of gm row-panels walked column by column (L2 reuse)
  int ntk, T;           // K-tiles per segment; K-tiles per output tile (= ntk * nseg)
  int mode;             // MODE_TILE: workgroup v computes tile v.  MODE_SPLIT: "C +=" split-K, workgroup v adds the
                        // piece (slice v / ntiles, tile v % ntiles) of P K-tiles.  MODE_STREAM: workgroup v computes
                        // the work units [v*P, (v+1)*P) of the tile-major (tile, K-tile) sequence of U units
  int n_full, full_per, sk_per;  // MODE_STREAM hybrid: tiles [0, n_full) are computed whole, one per workgroup; only the
                        // tail tiles are streamed.  Per XCD: sk_per streaming workgroups (launched first), then full_per whole tiles
  int P;                // K-tiles per piece / work units per workgroup
  int U;                // work units in total (= ntiles * T; stream-K needs it below 2^31)
  int gtile0[MAXG];     // first linear tile of every group (INT_MAX past ngroups): a copy of g[].tile0 inside the header's first
  int tn_shift;         // cache lines, so that the group of a tile is found without touching the group records.  tn_shift (LEAN
                        // kernels): log2 of tiles_n
  unsigned nt_magic;    // (LEAN TN kernels) ceil(2^32 / ntiles): piece v is slice (v * nt_magic) >> 32 -- exact below 2^16 x 2^16
  float* ws;            // partial-tile slots: 2 per workgroup, BM*BN floats each
  int* cnt;             // per-tile arrival counters (zero between launches)
  int avec, bvec;       // generic path: 16-byte vector loads legal for the A / B operand
  float alpha, gate_scale;
  DropCfg drop;
  // LSTM time-step epilogues (EPI template parameter; lstm_* entry points below).  M = batch rows.
  //   EPI 1 (forward, N = 4H with column 4j+gate = gate `gate` of unit j): pre-activations = result + residual;
  //     gates -> lg_gates[M,4H], c = f c_prev + i g -> lg_cout[M,H], h = o tanh(c) -> C[M,H] and lg_h2 (ld lg_ldh2)
  //   EPI 2 (backward, N = H): dh = result + residual; with the step's saved gates, c, c_prev and the running dc
  //     (lg_dc, updated in place) writes the pre-activation gradients lg_gates[M,4H]; C is not written
  const float* lg_cprev;
  const float* lg_c;
  float* lg_cout;
  float* lg_gates;
  float* lg_h2;
  float* lg_dc;
  const float* lg_act;
  int lg_ldh2, lg_pad;
  // (the groups last: the first three end inside the 512 bytes the warm-up loads touch)
  GemmGroupK g[MAXG];
};

enum { MODE_TILE = 0, MODE_SPLIT = 1, MODE_STREAM = 2 };

__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

#ifndef MMNAS_OCC64
#define MMNAS_OCC64 4
#endif
#ifndef MMNAS_OCC_NS
#define MMNAS_OCC_NS 3      // workgroups per CU of the 64^2 split-operand kernels
#endif
#ifndef MMNAS_OCC128
#define MMNAS_OCC128 2
#endif
#ifndef MMNAS_BK
#define MMNAS_BK 32
#endif
constexpr int BK = MMNAS_BK;   // K-tile depth
constexpr int LDK = BK + 4;    // LDS row stride of a K-contiguous operand
constexpr int KQ = BK / 4;     // float4 chunks per K-contiguous row

// Partial tiles travel between workgroups that may sit on different XCDs (each XCD has its own L2):
// device-scope relaxed atomic accesses (sc1: write-through stores, L2-missing loads) move just these bytes
// coherently -- a release/acquire FENCE would write back / invalidate the whole L2 (measured: +200 us).
typedef unsigned long long u64;
__device__ __forceinline__ void st_agent(u64* ptr, float a, float b) {
  __hip_atomic_store(ptr, ((u64)__float_as_uint(b) << 32) | __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld_agent(const u64* ptr) {
  return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// thread tid, load pair j of a row-contiguous ([k][rows]) tile of BR rows -> k-pair kp (rows k = 2 kp, 2 kp + 1) and row quad
// rq (rows 4 rq .. 4 rq + 3).  MMNAS_TMAP: 2 (default) = a 32-lane group holds 4 k-pairs x 8 quads and the lanes of quads
// 4-7 store their four rows in the order 1,0,3,2 (t_flip: 20 (e ^ 1) - 20 e = +-20 moves the bank by 4) -- a wave's load
// instruction then covers 4 k-rows x 256 contiguous bytes as the lane-linear map does; 1 = 8 k-pairs x 4 quads, no flip
// (8 k-rows x 128 B per instruction: measured 4-15 % slower on the weight-gradient products, whose operands stream from
// HBM); 0 = the lane-linear map of round 2 (2-way store conflicts; A/B only).
#ifndef MMNAS_TMAP
#define MMNAS_TMAP 2
#endif
template <int BR>
__device__ __forceinline__ void t_map(int tid, int j, int& kp, int& rq) {
#if MMNAS_TMAP == 2
  const int hi = tid >> 5;   // lane bit 5 and the wave
  constexpr int RB = BR / 32;   // quads beyond the 8 of a lane group: 2 (BR 64) or 4 (BR 128)
  rq = (tid & 7) | ((hi % RB) << 3);
  kp = ((tid >> 3) & 3) + 4 * (hi / RB) + (1024 / BR) * j;
#elif MMNAS_TMAP == 1
  rq = (tid >> 3) % (BR / 4);
  kp = (tid & 7) + 8 * ((tid >> 3) / (BR / 4)) + (1024 / BR) * j;
#else
  kp = tid / (BR / 4) + (1024 / BR) * j;
  rq = tid % (BR / 4);
#endif
}
__device__ __forceinline__ int t_flip(int rq) { return MMNAS_TMAP == 2 ? (rq >> 2) & 1 : 0; }

template <int NS>
__device__ __forceinline__ void split_put(unsigned* dst, float x0, float x1, int row, int kp) {
  constexpr int RSW = NS * 16 + 4;
  unsigned a0, a1 = 0, a2 = 0;
  split_pair<NS>(x0, x1, a0, a1, a2);
  unsigned* d = dst + row * RSW + swz(kp, row);
  d[0] = a0;
  if (NS > 1) d[16] = a1;
  if (NS > 2) d[32] = a2;
}
template <int BR, int NS>
__device__ __forceinline__ void split_store_t(unsigned* dst, const float4 e, const float4 o, int tid, int j) {
  int kp, rq;
  t_map<BR>(tid, j, kp, rq);
  const int row = 4 * rq, fl = t_flip(rq);
  // (fl: this lane's rows go out in the order 1,0,3,2 -- the select is on the inputs, 8 v_cndmask per pair of loads)
  split_put<NS>(dst, fl ? e.y : e.x, fl ? o.y : o.x, row + fl, kp);
  split_put<NS>(dst, fl ? e.x : e.y, fl ? o.x : o.y, row + 1 - fl, kp);
  split_put<NS>(dst, fl ? e.w : e.z, fl ? o.w : o.z, row + 2 + fl, kp);
  split_put<NS>(dst, fl ? e.z : e.w, fl ? o.z : o.w, row + 3 - fl, kp);
}

#ifdef MMNAS_DBG_STAMP
__device__ unsigned long long g_stamp_acc[8];
__device__ unsigned long long g_life_acc[8];   // whole-workgroup phases: arguments, set-up, first tile in LDS, K loop, epilogue
#endif

// NS > 0: the operands are split into NS bf16 parts while they are written to LDS and the products run on
// v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate per instruction), fp32 accumulation as before:
//   NS = 1: x ~ h = bf16(x) (8 mantissa bits), ONE product hh                    -- "bf16x1": the single-pass reduced-
//           precision flavour of BASELINE configs[4] ("fp16 MFMA"); never a default, never behind a parity claim
//   NS = 2: x = h + l  (16 mantissa bits kept), products hh + hl + lh            -- "bf16x3"
//   NS = 3: x = h + m + l (all 24 bits),        products hh + hm + mh + mm + hl + lh -- "bf16x6", fp32-grade
// (the dropped cross terms are below 2^-16 resp. 2^-24 of |a||b|).  NS = 3 is the default path (every shape measured is
// faster than on the fp32 MFMA and the error against fp64 is not larger); NS = 0 / 2 by MMNAS_GEMM_SPLIT: Tuning::split.
template <int BM, int BN, int NS>
struct GemmShape {
  static constexpr int RSW = NS ? NS * 16 + 4 : LDK;   // LDS row stride in 4-byte words
  static constexpr int A_SZ = BM * RSW, B_SZ = BN * RSW;  // >= BK*BM for the [k][row] form
};

// The work of workgroup `bid` of `nwg` on problem p (the kernel's own blockIdx / gridDim, or its position inside one
// section of a two-section launch); `koff` = byte offset of p inside the kernel-argument segment.
// PF: K-tiles of operand loads kept in flight ahead of the MFMA block (1: the tile after the current one; 2: two tiles,
// a second register stage -- the products of the d = 256 supernet have 8 K-tiles and 1-2 resident workgroups per CU, so
// with one tile in flight every iteration waits out a full L2 / Infinity-Cache round trip, ~2x its 0.43 us of MFMA).
// BDMA (NT, 64^2 tiles, NS = 3, PF = 1 only): the B operand -- a weight matrix -- arrives as three PRE-SPLIT bf16 planes
// (mmnas_split_planes: [3][N][ldb] bf16, plane c = part c of every element) and goes global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR stage, no conversion VALU, no ds_write).  An LDS-DMA instruction writes 1 KiB
// lane-linearly, so the B image has no row pad: row r = 3 runs of 64 B (192 B), the 16-B chunk j of a run stored at
// position j ^ ((r >> 2) & 3) -- the swizzle is applied on the per-lane SOURCE address and again on the fragment read.
// Bank check of the fragment reads (ds_read_b128, 64 banks, lane groups of 16 rows {0-3,12-15,20-27}, ...): the row base
// 48 r mod 64 takes the four values 0/48/32/16 by r & 3, rows r, r+4, r+8, r+12 of one residue take the four chunk
// positions by (r >> 2) & 3 -- 16 rows x 4 words cover the 64 banks once: conflict-free.
// LEAN (64^2 tiles, NS > 0, PF = 2): the same K loop between a short set-up and a short epilogue.  Levels: 1 = whole tiles
// only (NT / NN; no stream-K code at all: 65 SGPRs, no spills) or, for TN, split-K pieces added by buffer atomics; 2 = whole
// tiles + hybrid / stream-K schedules (NT / NN); 3 = whole-tile NN products with column sums (natural accumulator layout,
// element-wise buffer epilogue).
// A workgroup of the supernet's products lives ~22000 cycles of which the general set-up is ~2400 and the general epilogue
// ~3900 -- both pure instruction issue (~600 and ~900 instructions at one per four cycles for a lone wave of a SIMD):
//   * tile order without divisions: a group's tiles are numbered row panel by row panel, tiles_n (a power of two, host
//     check) column tiles each: an XCD's contiguous run of tiles is a run of whole A panels and it reads the (small, host
//     check) B matrix once;
//   * the MFMA operands change roles (B fragment as the A operand): the accumulators come out transposed, lane = row of C,
//     registers = 4 runs of 4 consecutive columns, so the epilogue is four 16-byte buffer stores (and as many loads of the
//     residual / gate / old value) with one row address per lane, predicated by the buffer range check.
template <int BM, int BN, bool AKC, bool BKC, bool FAST, int NS, int EPI = 0, int PF = 1, bool BDMA = false, int LEAN = 0>
__device__ __forceinline__ void gemm_body(const GemmK& p, const int bid, const int nwg, const int koff,
                                          float* __restrict__ As, float* __restrict__ Bs, float* __restrict__ As1,
                                          float* __restrict__ Bs1, int& s_old) {
  static_assert(PF == 1 || (PF == 2 && FAST && (NS == 0 || NS == 1 || NS == 3)), "the two-stage prefetch exists for the buffer-load path (fp32, bf16x1 and bf16x6)");
  static_assert(NS == 0 || (FAST && BK == 32), "the bf16-split path exists for the buffer-load path only");
  static_assert(!BDMA || (NS == 3 && AKC && BKC && FAST && PF == 1 && BM == 64 && BN == 64 && EPI == 0), "LDS-DMA weight planes: NT 64^2 bf16x6 only");
  static_assert(!LEAN || (NS > 0 && (AKC || !BKC) && FAST && PF == 2 && BM == 64 && BN == 64 && EPI == 0 && !BDMA), "lean form: 64^2 split-operand kernels");
  // LEAN, TN (weight gradients): split-K pieces only ("C +=" by float atomics), one K-segment; the adds are buffer atomics
  // on one per-lane offset (range check = predication) instead of 16 guarded 64-bit address computations
  constexpr bool LEAN_TN = LEAN && !AKC;
  // LEAN 3 (NN, whole tiles): the accumulators keep their natural layout (lane = column) and the epilogue works element by
  // element through buffer loads / stores on one per-lane offset -- for the products whose epilogue takes column sums (the
  // hidden-layer gradient of an FFN with its bias gradient): a column's rows sit in one lane's registers
  constexpr bool LEAN_EL = LEAN == 3;
  constexpr bool LEAN_SW = LEAN && AKC && !LEAN_EL;   // transposed accumulators + 16-byte epilogue rows
  constexpr bool LEAN_WT = (LEAN_SW && LEAN == 1) || LEAN_EL;   // whole tiles only: no stream-K code at all
  constexpr int RSWB = 48;   // BDMA: words per row of the B image (3 runs of 16 words, no pad)
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
  // split rows: NS parts of 32 bf16 (64 B) + 16 B pad -> 144 / 208 B, an odd number of 16-B words (conflict-free b128)
  constexpr int RSW = GemmShape<BM, BN, NS>::RSW;
  constexpr int NA = BM * BK / 1024, NB = BN * BK / 1024;   // float4 loads per thread per tile

#ifdef MMNAS_DBG_STAMP
  unsigned long long life_prev = __builtin_readcyclecounter();
#define MMNAS_LIFE(i) do { const unsigned long long c_ = __builtin_readcyclecounter(); if (threadIdx.x == 0 && bid == MMNAS_DBG_STAMP) g_life_acc[i] += c_ - life_prev; life_prev = c_; } while (0)
#else
#define MMNAS_LIFE(i) do { } while (0)
#endif
  // The kernel arguments (~1.2 KB with nine groups; the header and the first three groups are 10 cache lines) are written by the
  // runtime just before the launch (device memory under HIP_FORCE_DEV_KERNARG=1, this image's default; host-visible memory
  // otherwise): the first touch of each line is a miss of up to ~1 us, and the group record is read in a dependent step.
  // Touch every line up front so the misses overlap (measured in round 2: -3.5 us per launch).
  {
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr() + koff;
    unsigned t0, t1, t2, t3, t4, t5, t6, t7, t8, t9;
    asm volatile(
        "s_load_dword %0, %10, 0x0\n\ts_load_dword %1, %10, 0x40\n\ts_load_dword %2, %10, 0x80\n\t"
        "s_load_dword %3, %10, 0xc0\n\ts_load_dword %4, %10, 0x100\n\ts_load_dword %5, %10, 0x140\n\t"
        "s_load_dword %6, %10, 0x180\n\ts_load_dword %7, %10, 0x1c0\n\ts_load_dword %8, %10, 0x200\n\t"
        "s_load_dword %9, %10, 0x240\n\ts_waitcnt lgkmcnt(0)"
        : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9)   // early-clobber: the
                                                  // loads return while later ones are still being issued from %10
        : "s"(ka)
        : "memory");
  }
  static_assert(offsetof(GemmK, g) + 3 * sizeof(GemmGroupK) > 0x200 && offsetof(GemmK, g) + 3 * sizeof(GemmGroupK) <= 0x280,
                "update the kernel-argument warm-up loads (header + the first three groups; further groups -- the architecture "
                "step's node-wide launches -- take their first-touch miss)");

  MMNAS_LIFE(0);   // kernel-argument warm-up
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- this workgroup's run of work units (XCD-aware bijective remap: blocks b and b+8 share an XCD,
  //      so an XCD owns a contiguous run of units = neighbouring tiles / K-slices of few tiles) ----
  // Every scalar of the header that the set-up and the K loop read, loaded here in ONE batch and kept: left to the compiler
  // each branch of the set-up loads its own fields and waits for them, and under register pressure it re-loads rather than
  // keeps them -- 19 dependent scalar-memory round trips (~200 cycles each on cache hits) stood between the kernel's first
  // instruction and its first operand load, 3900 cycles of a workgroup's ~10000 on the supernet's products
  // (tools/gemm_stamps.py).  The empty asm makes each value opaque, so it cannot be re-materialised by a second load.
  struct Hdr { int N, K, lda, ldb, tiles_n, ntiles, xcd_remap, gm, ntk, T, mode, n_full, full_per, sk_per, P, U, gt[MAXG]; };
  Hdr h;
  h.N = p.N; h.K = p.K; h.lda = p.lda; h.ldb = p.ldb; h.tiles_n = p.tiles_n; h.ntiles = p.ntiles; h.xcd_remap = p.xcd_remap;
  h.gm = p.gm; h.ntk = p.ntk; h.T = p.T; h.mode = p.mode; h.n_full = p.n_full; h.full_per = p.full_per; h.sk_per = p.sk_per;
  h.P = p.P; h.U = p.U;
#pragma unroll
  for (int g = 0; g < MAXG; ++g) h.gt[g] = p.gtile0[g];
  asm volatile("" : "+s"(h.N), "+s"(h.K), "+s"(h.lda), "+s"(h.ldb), "+s"(h.tiles_n), "+s"(h.ntiles), "+s"(h.xcd_remap), "+s"(h.gm),
               "+s"(h.ntk), "+s"(h.T), "+s"(h.mode), "+s"(h.n_full), "+s"(h.full_per), "+s"(h.sk_per), "+s"(h.P));
  static_assert(MAXG == 9, "the list of group boundaries below names nine groups");
  asm volatile("" : "+s"(h.U), "+s"(h.gt[1]), "+s"(h.gt[2]), "+s"(h.gt[3]), "+s"(h.gt[4]), "+s"(h.gt[5]), "+s"(h.gt[6]), "+s"(h.gt[7]),
               "+s"(h.gt[8]));
  int v;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, in = bid >> 3;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + in;
    if (!h.xcd_remap) v = bid;
  }
  const int mode = LEAN_TN ? (int)MODE_SPLIT : (LEAN_WT ? (int)MODE_TILE : h.mode);
  int u = 0, uend = 1;  // MODE_TILE / MODE_SPLIT: a single piece
  int vs = v;           // MODE_STREAM: index among the streaming workgroups
  int whole = -1;       // hybrid: the whole tile of this workgroup
  if (mode == MODE_STREAM) {
    if (h.n_full > 0) {  // grid = 8 * (sk_per + full_per); bid % 8 labels the XCD, low indices start first
      const int xcd = bid & 7, li = bid >> 3;
      if (li < h.sk_per) vs = xcd * h.sk_per + li;
      else whole = xcd * h.full_per + (li - h.sk_per);
    }
    if (whole < 0) { u = vs * h.P; uend = min(h.U, u + h.P); }
  }

  while (u < uend) {
    int tile, q0, nq;
    if (mode == MODE_TILE || whole >= 0) {
      tile = whole >= 0 ? whole : v; q0 = 0; nq = h.T;
      u = uend;
    } else if (!LEAN_SW && !LEAN_EL && mode == MODE_SPLIT) {  // slice-major: neighbouring workgroups (one XCD) add into different tiles
      const int sl = LEAN_TN ? (int)__umulhi((unsigned)v, p.nt_magic) : v / h.ntiles;   // and stream the same K-slice of both operands through its L2
      tile = v - sl * h.ntiles;
      q0 = sl * h.P;
      nq = min(h.P, h.T - q0);
      u = uend;
      if (nq <= 0) break;
    } else {
      const int tt = u / h.T;  // tile among the streamed (tail) tiles
      tile = h.n_full + tt;
      q0 = u - tt * h.T;       // first K-tile of the piece
      nq = min(h.T - q0, uend - u);
      u += nq;
    }
    int grp = 0;   // (gtile0 rises with the group and is INT_MAX past the last one: the group is a count, no branches)
#pragma unroll
    for (int g = 1; g < MAXG; ++g) grp += tile >= h.gt[g] ? 1 : 0;
    const GemmGroupK& G = p.g[grp];
    const int Mg = G.M;
    const int tl = tile - G.tile0;
    // group fields into registers once per tile (re-reading them through the kernel-argument pointer inside
    // the epilogue made the compiler reload the pointer and drain vmcnt before every store / atomic)
    float* __restrict__ const Cp = G.C;
    const float* __restrict__ const biasp = G.bias;
    const float* __restrict__ const resp = G.residual;
    const float* __restrict__ const gatep = G.gate;
    float* __restrict__ const csp = G.colsum;
    const uint32_t gseed_lo = G.seed_lo, gsite_key = G.site_key;
    const float* const Aseg[3] = {G.A[0], G.A[1], G.A[2]};
    const float* const Bseg[3] = {G.B[0], G.B[1], G.B[2]};
    // Tiles of a problem are numbered in blocks of gm row-panels x all column tiles, column by column inside a
    // block: the workgroups resident on an XCD at one time (consecutive numbers) then share gm A-panels and only
    // (resident / gm) B-panels, instead of one A-panel per tiles_n workgroups and ALL of B (N = 2048: the whole
    // 4 MB weight matrix fell out of the 4 MB L2 between row-panels -- 7x the algorithmic fabric reads).
    int tile_m, tile_n;
    if (LEAN) {   // row panel by row panel; tiles_n = 1 << tn_shift
      tile_m = tl >> p.tn_shift;
      tile_n = tl - (tile_m << p.tn_shift);
    } else {
      const int tiles_m = cdiv_dev(Mg, BM);
      const int per_block = h.gm * h.tiles_n;
      const int blk = tl / per_block, in = tl - blk * per_block;
      const int rows = min(h.gm, tiles_m - blk * h.gm);   // the last block may be shorter
      tile_n = in / rows;
      tile_m = blk * h.gm + (in - tile_n * rows);
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 rA[2][NA], rB[2][NB];   // register stages of operand prefetch (the second one: PF == 2; dead otherwise)

    // FAST path: per-thread byte offsets of its loads inside the operand (k = 0), ~0u when the row is
    // outside the matrix (the buffer range check then returns zeros)
    unsigned offa[NA], offb[NB];
    size_t bdma_off[3] = {0, 0, 0};   // BDMA: byte offset of this lane's chunk of piece i inside the plane block (k = 0)
    unsigned stepa = 0, stepb = 0;  // bytes per K-tile
    unsigned bytesa = 0, bytesb = 0;
    if (FAST) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int f = tid + 256 * i;
        if (AKC) {
          const int row = NS ? kc_row(f) : f / KQ, kq = f % KQ, gr = m0 + row;
          offa[i] = gr < Mg ? (unsigned)(gr * h.lda + 4 * kq) * 4u : ~0u;
        } else if (NS) {  // loads 2j / 2j+1 of a thread: rows k = 2kp, 2kp+1 of the same 4 columns (packed as bf16 pairs)
          int kp, rq;
          t_map<BM>(tid, i >> 1, kp, rq);
          const int gr = m0 + 4 * rq;
          offa[i] = gr < Mg ? (unsigned)((2 * kp + (i & 1)) * h.lda + gr) * 4u : ~0u;
        } else {
          const int k = f / (BM / 4), rq = f - k * (BM / 4), gr = m0 + 4 * rq;
          offa[i] = gr < Mg ? (unsigned)(k * h.lda + gr) * 4u : ~0u;
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int f = tid + 256 * i;
        if (BDMA) {
          offb[i] = 0;
        } else if (BKC) {
          const int row = NS ? kc_row(f) : f / KQ, kq = f % KQ, gr = n0 + row;
          offb[i] = gr < h.N ? (unsigned)(gr * h.ldb + 4 * kq) * 4u : ~0u;
        } else if (NS) {
          int kp, rq;
          t_map<BN>(tid, i >> 1, kp, rq);
          const int gr = n0 + 4 * rq;
          offb[i] = gr < h.N ? (unsigned)((2 * kp + (i & 1)) * h.ldb + gr) * 4u : ~0u;
        } else {
          const int k = f / (BN / 4), rq = f - k * (BN / 4), gr = n0 + 4 * rq;
          offb[i] = gr < h.N ? (unsigned)(k * h.ldb + gr) * 4u : ~0u;
        }
      }
      if (BDMA) {   // piece p = 3 wave + i of the 12 KiB image: this lane's 16 bytes sit at byte p * 1024 + 16 lane
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int o = (wave * 3 + i) * 1024 + lane * 16;
          const int r = o / 192, w = o - r * 192, c = w >> 6, j = ((w & 63) >> 4) ^ ((r >> 2) & 3);
          bdma_off[i] = ((size_t)c * (size_t)h.N * (size_t)h.ldb + (size_t)(n0 + r) * (size_t)h.ldb + 8 * j) * 2u;
        }
      }
      stepa = AKC ? BK * 4u : (unsigned)h.lda * BK * 4u;
      stepb = BKC ? BK * 4u : (unsigned)h.ldb * BK * 4u;
      bytesa = (unsigned)(AKC ? Mg : h.K) * (unsigned)h.lda * 4u;
      bytesb = (unsigned)(BKC ? h.N : h.K) * (unsigned)h.ldb * 4u;
    }

    // unit q of the tile: segment q / ntk, K-tile q % ntk; `live` false (FAST path only): every offset is put out of
    // range, the buffer bounds check answers with zeros and no memory request is made -- a branch-free "no load"
    int seg_c = 0, kt_c = 0;   // segment / K-tile of the next unit gload_to is asked for
    if (LEAN_TN) kt_c = q0;
    else if (q0 != 0) { seg_c = q0 / h.ntk; kt_c = q0 - seg_c * h.ntk; }
    auto gload_to = [&](int q, bool live, const int st, const int dbuf = 0) __attribute__((always_inline)) {
      float4* const ra = rA[st];
      float4* const rb = rB[st];
      // (the calls of a tile ask for consecutive units q0, q0 + 1, ...: segment and K-tile are carried along instead of
      //  divided out of q every time -- the scalar division cost every wave ~100 cycles per K-tile)
      const int seg = seg_c, kt = kt_c;
      if (++kt_c == h.ntk) { kt_c = 0; ++seg_c; }
      const float* __restrict__ Ap = seg == 0 ? Aseg[0] : (seg == 1 ? Aseg[1] : Aseg[2]);
      const float* __restrict__ Bp = seg == 0 ? Bseg[0] : (seg == 1 ? Bseg[1] : Bseg[2]);
      if (FAST) {
        const __amdgpu_buffer_rsrc_t ra_src = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, bytesa, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb_src = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, bytesb, 0x00020000);
        const unsigned ka = (unsigned)kt * stepa, kb = (unsigned)kt * stepb;
        if (LEAN_SW || LEAN_EL) {
          // (NT / NN; measured neutral to negative on the transposing loads of TN)  the K-tile's byte offset rides in the instruction's scalar offset -- the range check covers voffset + soffset and
          // a voffset of ~0u stays out of range (tools/probe/soffset_probe.hip) -- and "no load" is a descriptor of zero
          // records: no per-load compare / select / add, and the loads keep their places between the MFMAs
          const __amdgpu_buffer_rsrc_t la_src = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, live ? bytesa : 0u, 0x00020000);
          const __amdgpu_buffer_rsrc_t lb_src = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, live ? bytesb : 0u, 0x00020000);
#pragma unroll
          for (int i = 0; i < NA; ++i) ra[i] = buf_load4(la_src, offa[i], ka);
#pragma unroll
          for (int i = 0; i < NB; ++i) rb[i] = buf_load4(lb_src, offb[i], kb);
          return;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = buf_load4(ra_src, (offa[i] == ~0u || !live) ? ~0u : offa[i] + ka);
        if (BDMA) {
          // the K-tile's 64 bytes of every run: + 64 kt bytes on the source; the LDS destination is wave-uniform
          const char* bsrc = reinterpret_cast<const char*>(Bp) + (size_t)kt * 64u;
          float* const bd = dbuf ? Bs1 : Bs;
          const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
          for (int i = 0; i < 3; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc + bdma_off[i]),
                                             (__attribute__((address_space(3))) void*)(bd + (wv * 3 + i) * 256), 16, 0, 0);
          return;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = buf_load4(rb_src, (offb[i] == ~0u || !live) ? ~0u : offb[i] + kb);
        return;
      }
      const int k0 = kt * BK, kend = h.K;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int f = tid + 256 * i;
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (AKC) {
          const int row = f / KQ, kq = f % KQ;
          const int gr = m0 + row, gk = k0 + 4 * kq;
          if (gr < Mg && gk < kend) {
            const float* src = Ap + (size_t)gr * h.lda + gk;
            if (p.avec) v4 = *reinterpret_cast<const float4*>(src);
            else {
              v4.x = src[0];
              if (gk + 1 < kend) v4.y = src[1];
              if (gk + 2 < kend) v4.z = src[2];
              if (gk + 3 < kend) v4.w = src[3];
            }
          }
        } else {
          const int k = f / (BM / 4), rq = f - k * (BM / 4);
          const int gk = k0 + k, gr = m0 + 4 * rq;
          if (gk < kend && gr < Mg) {
            const float* src = Ap + (size_t)gk * h.lda + gr;
            if (p.avec) v4 = *reinterpret_cast<const float4*>(src);
            else {
              v4.x = src[0];
              if (gr + 1 < Mg) v4.y = src[1];
              if (gr + 2 < Mg) v4.z = src[2];
              if (gr + 3 < Mg) v4.w = src[3];
            }
          }
        }
        ra[i] = v4;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int f = tid + 256 * i;
        float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BKC) {
          const int row = f / KQ, kq = f % KQ;
          const int gr = n0 + row, gk = k0 + 4 * kq;
          if (gr < h.N && gk < kend) {
            const float* src = Bp + (size_t)gr * h.ldb + gk;
            if (p.bvec) v4 = *reinterpret_cast<const float4*>(src);
            else {
              v4.x = src[0];
              if (gk + 1 < kend) v4.y = src[1];
              if (gk + 2 < kend) v4.z = src[2];
              if (gk + 3 < kend) v4.w = src[3];
            }
          }
        } else {
          const int k = f / (BN / 4), rq = f - k * (BN / 4);
          const int gk = k0 + k, gr = n0 + 4 * rq;
          if (gk < kend && gr < h.N) {
            const float* src = Bp + (size_t)gk * h.ldb + gr;
            if (p.bvec) v4 = *reinterpret_cast<const float4*>(src);
            else {
              v4.x = src[0];
              if (gr + 1 < h.N) v4.y = src[1];
              if (gr + 2 < h.N) v4.z = src[2];
              if (gr + 3 < h.N) v4.w = src[3];
            }
          }
        }
        rb[i] = v4;
      }
    };
    auto gload = [&](int q, int dbuf = 0) __attribute__((always_inline)) { gload_to(q, true, 0, dbuf); };

    auto lstore_from = [&](int buf, const int st) __attribute__((always_inline)) {
      const float4* const ra = rA[st];
      const float4* const rb = rB[st];
      float* a = buf ? As1 : As;   // (the two buffers are separate LDS objects: a store into one provably does not alias
      float* b = buf ? Bs1 : Bs;   //  a fragment read of the other)
      if (NS) {
        unsigned* ua = reinterpret_cast<unsigned*>(a);
        unsigned* ub = reinterpret_cast<unsigned*>(b);
        if (AKC) {
#pragma unroll
          for (int i = 0; i < NA; ++i) split_store_kc<BM, true, NS ? NS : 2>(ua, ra[i], tid + 256 * i);
        } else {
#pragma unroll
          for (int j = 0; j < NA / 2; ++j) split_store_t<BM, NS ? NS : 2>(ua, ra[2 * j], ra[2 * j + 1], tid, j);
        }
        if (BDMA) {
          // (the B image was written by the LDS-DMA loads of gload_to)
        } else if (BKC) {
#pragma unroll
          for (int i = 0; i < NB; ++i) split_store_kc<BN, true, NS ? NS : 2>(ub, rb[i], tid + 256 * i);
        } else {
#pragma unroll
          for (int j = 0; j < NB / 2; ++j) split_store_t<BN, NS ? NS : 2>(ub, rb[2 * j], rb[2 * j + 1], tid, j);
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int f = tid + 256 * i;
        if (AKC) {
          const int row = f / KQ, kq = f % KQ;
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
          const int row = f / KQ, kq = f % KQ;
          *reinterpret_cast<float4*>(b + row * LDK + 4 * kq) = rb[i];
        } else {
          const int k = f / (BN / 4), rq = f - k * (BN / 4);
          *reinterpret_cast<float4*>(b + k * BN + 4 * rq) = rb[i];
        }
      }
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) { lstore_from(buf, 0); };

    auto mfma_block = [&](int buf) __attribute__((always_inline)) {
#ifdef MMNAS_DBG_NOMFMA   // timing experiment only (wrong results): the K loop without fragment reads and MFMAs
      return;
#endif
      const float* a = buf ? As1 : As;
      const float* b = buf ? Bs1 : Bs;
      if (NS) {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
          bf16x8 af[TM][NS ? NS : 1], bf[TN][NS ? NS : 1];
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int c = 0; c < NS; ++c)
              af[i][c] = *reinterpret_cast<const bf16x8*>(a + (wm * WM + i * 32 + l31) * RSW + c * 16 + swz(s * 8 + hh * 4, wm * WM + i * 32 + l31));
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int c = 0; c < NS; ++c)
              bf[j][c] = BDMA ? *reinterpret_cast<const bf16x8*>(b + (wn * WN + j * 32 + l31) * RSWB + c * 16 + 4 * ((s * 2 + hh) ^ (((wn * WN + j * 32 + l31) >> 2) & 3)))
                              : *reinterpret_cast<const bf16x8*>(b + (wn * WN + j * 32 + l31) * RSW + c * 16 + swz(s * 8 + hh * 4, wn * WN + j * 32 + l31));
          // smallest cross terms first; part c of A with part e of B is kept while c + e < NS
#pragma unroll
          for (int o = NS - 1; o >= 0; --o)
#pragma unroll
            for (int c = 0; c <= o; ++c)
#pragma unroll
              for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                  acc[i][j] = LEAN_SW ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][o - c], af[i][c], acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][c], bf[j][o - c], acc[i][j], 0, 0, 0);
        }
      } else
#pragma unroll
      for (int s = 0; s < BK / 8; ++s) {
        float af[TM][4], bf[TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wm * WM + i * 32 + l31;
          if (AKC) {
            const float4 v4 = *reinterpret_cast<const float4*>(a + row * LDK + 8 * s + 4 * hh);
            af[i][0] = v4.x; af[i][1] = v4.y; af[i][2] = v4.z; af[i][3] = v4.w;
          } else {
#pragma unroll
            for (int w = 0; w < 4; ++w) af[i][w] = a[(8 * s + 4 * hh + w) * BM + row];
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = wn * WN + j * 32 + l31;
          if (BKC) {
            const float4 v4 = *reinterpret_cast<const float4*>(b + col * LDK + 8 * s + 4 * hh);
            bf[j][0] = v4.x; bf[j][1] = v4.y; bf[j][2] = v4.z; bf[j][3] = v4.w;
          } else {
#pragma unroll
            for (int w = 0; w < 4; ++w) bf[j][w] = b[(8 * s + 4 * hh + w) * BN + col];
          }
        }
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(af[i][w], bf[j][w], acc[i][j]);
      }
    };

    if (PF == 2) {
      // two register stages: while tile t is multiplied out of LDS, tile t+1 sits in (or is arriving into) one stage
      // and the loads of tile t+2 are issued into the other; the loop is unrolled by two so the stages are static
      MMNAS_LIFE(1);   // tile / group / offset set-up
      gload_to(q0, true, 0);
      gload_to(q0 + 1, nq > 1, 1);
      lstore_from(0, 0);
      __syncthreads();
      MMNAS_LIFE(2);   // first tile: loads -> conversion -> LDS -> barrier
#ifdef MMNAS_DBG_STAMP   // timing experiment only: cycle stamps of one wave per phase of the first half-iteration (tools/gemm_stamps.py)
#define MMNAS_STAMP(i) do { const unsigned long long c_ = __builtin_readcyclecounter(); stamp_acc[i] += (unsigned)(c_ - stamp_prev); stamp_prev = c_; } while (0)
      const bool stamp_on = tid == 0 && bid == MMNAS_DBG_STAMP;
      unsigned stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      unsigned long long stamp_prev = __builtin_readcyclecounter();
#else
#define MMNAS_STAMP(i) do { } while (0)
#endif
      for (int t = 0; t < nq; t += 2) {
        MMNAS_STAMP(0);                      // second half of the previous iteration (all of it)
        gload_to(q0 + t + 2, t + 2 < nq, 0);
        MMNAS_STAMP(1);                      // load issue
        mfma_block(0);
#ifdef MMNAS_DBG_STAMP
        asm volatile("s_nop 0" ::: "memory");
#endif
        MMNAS_STAMP(2);                      // fragment reads + MFMAs (issue; the last MFMAs may still run)
        if (t + 1 >= nq) break;
#ifdef MMNAS_DBG_STAMP
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
        MMNAS_STAMP(3);                      // wait for the loads of tile t+1
        lstore_from(1, 1);
        MMNAS_STAMP(4);                      // conversion + LDS store issue
#ifdef MMNAS_DBG_STAMP
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        MMNAS_STAMP(5);                      // LDS stores landed
        __syncthreads();
        MMNAS_STAMP(6);                      // barrier
        gload_to(q0 + t + 3, t + 3 < nq, 1);
        mfma_block(1);
        if (t + 2 < nq) lstore_from(0, 0);
        __syncthreads();
      }
      __syncthreads();
#ifdef MMNAS_DBG_STAMP
      if (stamp_on)
        for (int i = 0; i < 8; ++i) g_stamp_acc[i] += stamp_acc[i];
#endif
      MMNAS_LIFE(3);   // K loop
    } else {
      gload(q0, 0);
      lstore(0);
      __syncthreads();
      for (int t = 0; t < nq; ++t) {
        const int buf = t & 1;
        if (t + 1 < nq) gload(q0 + t + 1, buf ^ 1);  // in flight during the MFMA block (BDMA: B lands in the other buffer)
        mfma_block(buf);
        if (t + 1 < nq) lstore(buf ^ 1);
        __syncthreads();
      }
    }

    // ---- partial tile: hand the accumulators over; the last contributor to arrive finishes the tile ----
    const bool atomic_out = LEAN_TN || (!LEAN && nq != h.T && p.accumulate);  // "C +=" results: a partial tile simply adds its share
    if (!LEAN_TN && !LEAN_WT && nq != h.T && !atomic_out) {
      const int t0 = (tile - h.n_full) * h.T;  // first unit of the tile in the streamed sequence
      const int v_lo = t0 / h.P, v_hi = (t0 + h.T - 1) / h.P;  // contributors (streaming workgroup indices), inclusive
      // a workgroup has at most two partial tiles: the one it starts inside (slot 2v) and the one it
      // ends inside (slot 2v+1)
      // slot image: [register pair][thread] of 8-byte words -- every store / load instruction covers 512
      // contiguous bytes
      u64* slot = reinterpret_cast<u64*>(p.ws) + (size_t)(2 * vs + (q0 == 0 ? 1 : 0)) * (BM * BN / 2) + tid;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int c = 0; c < 8; ++c) st_agent(slot + ((i * TN + j) * 8 + c) * 256, acc[i][j][2 * c], acc[i][j][2 * c + 1]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every lane's partial has been written through ...
      __syncthreads();                                    // ... before the workgroup announces its arrival
      if (tid == 0) s_old = __hip_atomic_fetch_add(p.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const bool last = s_old == v_hi - v_lo;
      __syncthreads();  // s_old is rewritten by the next partial tile
      if (!last) continue;
      if (tid == 0) __hip_atomic_store(p.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
      for (int c = v_lo; c <= v_hi; ++c) {
        const bool head = c * h.P > t0;  // contributor c starts inside this tile
        const u64* src = reinterpret_cast<const u64*>(p.ws) + (size_t)(2 * c + (head ? 0 : 1)) * (BM * BN / 2) + tid;
        u64 part[TM * TN * 8];  // the whole partial in flight at once: the loads miss L2 by design (~2 us each)
#pragma unroll
        for (int e = 0; e < TM * TN * 8; ++e) part[e] = ld_agent(src + e * 256);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
              const u64 w = part[(i * TN + j) * 8 + c8];
              acc[i][j][2 * c8] += __uint_as_float((unsigned)w);
              acc[i][j][2 * c8 + 1] += __uint_as_float((unsigned)(w >> 32));
            }
      }
    }

    // ---- epilogue ----
    // All conditions on kernel arguments are wave-uniform and hoisted out of the element loops; the
    // residual / gate / accumulate operands of a 32x32 sub-tile are fetched as one batch of 16
    // independent loads (clamped row index instead of a branch) before any arithmetic.
    const bool has_res = resp != nullptr, has_gate = gatep != nullptr, has_acc = p.accumulate != 0;
    const bool has_drop = p.drop.thresh != 0, has_relu = p.relu != 0;
    DropCfg gdrop = p.drop;
    gdrop.seed_lo = gseed_lo; gdrop.site_key = gsite_key;
    // (LSTM epilogues: the extra operands into registers once, as for the group fields above)
    const float* __restrict__ const lg_cprev = EPI ? p.lg_cprev : nullptr;
    const float* __restrict__ const lg_c = EPI ? p.lg_c : nullptr;
    const float* __restrict__ const lg_act = EPI ? p.lg_act : nullptr;
    float* __restrict__ const lg_cout = EPI ? p.lg_cout : nullptr;
    float* __restrict__ const lg_gates = EPI ? p.lg_gates : nullptr;
    float* __restrict__ const lg_h2 = EPI ? p.lg_h2 : nullptr;
    float* __restrict__ const lg_dc = EPI ? p.lg_dc : nullptr;
    const int lg_ldh2 = EPI ? p.lg_ldh2 : 0;
    if constexpr (LEAN_TN) {
      // lane = column n0 + 32 wn + l31, register r = row m0 + 32 wm + 4 hh + (r & 3) + 8 (r >> 2)
      const int col = n0 + wn * 32 + l31;
      const int rbase = m0 + wm * 32 + 4 * hh;
      const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)Cp, 0, (unsigned)Mg * (unsigned)p.ldc * 4u, 0x00020000);
      const unsigned ldc4 = (unsigned)p.ldc * 4u;
      // rows behind Mg lie behind the buffer's range (the adds are dropped); columns behind N are masked off -- an offset of
      // ~0u is NOT out of range for a buffer atomic (measured: memory aperture violation), unlike for loads and stores
      const unsigned base = (unsigned)(rbase * p.ldc + col) * 4u;
      if (col < h.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc[0][0][r] * p.alpha, c_rs, base + (unsigned)((r & 3) + 8 * (r >> 2)) * ldc4, 0, 0);
      }
#ifdef MMNAS_DBG_STAMP
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      MMNAS_LIFE(4);
      continue;
    }
    if constexpr (LEAN_EL) {
      // lane = column n0 + 32 wn + l31, register r = row m0 + 32 wm + 4 hh + (r & 3) + 8 (r >> 2): one byte offset per lane,
      // + a scalar multiple of the leading dimension per register; rows behind M lie behind the buffers' ranges
      const int col = n0 + wn * 32 + l31;
      const int rbase = m0 + wm * 32 + 4 * hh;
      const bool cok = col < h.N;
      const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)Cp, 0, (unsigned)Mg * (unsigned)p.ldc * 4u, 0x00020000);
      const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc((void*)resp, 0, has_res ? (unsigned)Mg * (unsigned)p.ldres * 4u : 0u, 0x00020000);
      const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc((void*)gatep, 0, has_gate ? (unsigned)Mg * (unsigned)p.ldgate * 4u : 0u, 0x00020000);
      const unsigned cbase = cok ? (unsigned)(rbase * p.ldc + col) * 4u : ~0u;
      const unsigned rbase_o = cok ? (unsigned)(rbase * p.ldres + col) * 4u : ~0u;
      const unsigned gbase = cok ? (unsigned)(rbase * p.ldgate + col) * 4u : ~0u;
      const unsigned ldc4 = (unsigned)p.ldc * 4u, ldr4 = (unsigned)p.ldres * 4u, ldg4 = (unsigned)p.ldgate * 4u;
      float resv[16], gatev[16], oldv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned k = (unsigned)((r & 3) + 8 * (r >> 2));
        resv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rs, rbase_o, k * ldr4, 0));
        gatev[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(g_rs, gbase, k * ldg4, 0));
        oldv[r] = has_acc ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(c_rs, cbase, k * ldc4, 0)) : 0.f;
      }
      const float bv = biasp ? biasp[cok ? col : 0] : 0.f;
      float cs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        float val = acc[0][0][r] * p.alpha + bv;
        if (has_relu) val = fmaxf(val, 0.f);
        if (has_drop) val *= drop_mult(gdrop, (uint32_t)row * (uint32_t)h.N + (uint32_t)col);
        if (has_gate) val = gatev[r] > 0.f ? val * p.gate_scale : 0.f;
        if (has_res) val += resv[r];
        if (has_acc) val += oldv[r];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), c_rs, cbase, (unsigned)((r & 3) + 8 * (r >> 2)) * ldc4, 0);
        if (row < Mg) cs += val;
      }
      if (csp != nullptr) {   // the two lane halves hold the other 16 rows of a column
        cs += __shfl_xor(cs, 32, 64);
        if (hh == 0 && cok) atomicAdd(csp + col, cs);
      }
#ifdef MMNAS_DBG_STAMP
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      MMNAS_LIFE(4);
      continue;
    }
    if constexpr (LEAN_SW) {
      // transposed accumulators: this lane holds row m0 + 32 wm + l31, register 4 g + e = column n0 + 32 wn + 8 g + 4 hh + e
      const int row = m0 + wm * 32 + l31;
      const int cb = n0 + wn * 32 + 4 * hh;
      const bool rok = row < Mg;
      const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)Cp, 0, (unsigned)Mg * (unsigned)p.ldc * 4u, 0x00020000);
      const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc((void*)resp, 0, has_res ? (unsigned)Mg * (unsigned)p.ldres * 4u : 0u, 0x00020000);
      const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc((void*)gatep, 0, has_gate ? (unsigned)Mg * (unsigned)p.ldgate * 4u : 0u, 0x00020000);
      unsigned offc[4];
      float4 resv[4], gatev[4], oldv[4], biasv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = cb + 8 * g;
        const bool ok = rok && col < h.N;
        offc[g] = ok ? (unsigned)(row * p.ldc + col) * 4u : ~0u;
        // (an absent operand has a zero-length buffer: the loads answer with zeros and make no memory request)
        resv[g] = buf_load4(r_rs, ok ? (unsigned)(row * p.ldres + col) * 4u : ~0u);
        gatev[g] = buf_load4(g_rs, ok ? (unsigned)(row * p.ldgate + col) * 4u : ~0u);
        oldv[g] = has_acc ? buf_load4(c_rs, offc[g]) : make_float4(0.f, 0.f, 0.f, 0.f);
        biasv[g] = biasp ? *reinterpret_cast<const float4*>(biasp + (col < h.N ? col : 0)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = cb + 8 * g;
        float o4[4];
        const float r4[4] = {resv[g].x, resv[g].y, resv[g].z, resv[g].w};
        const float g4[4] = {gatev[g].x, gatev[g].y, gatev[g].z, gatev[g].w};
        const float a4[4] = {oldv[g].x, oldv[g].y, oldv[g].z, oldv[g].w};
        const float b4[4] = {biasv[g].x, biasv[g].y, biasv[g].z, biasv[g].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float val = acc[0][0][4 * g + e] * p.alpha + b4[e];
          if (has_relu) val = fmaxf(val, 0.f);
          if (has_drop) val *= drop_mult(gdrop, (uint32_t)row * (uint32_t)h.N + (uint32_t)(col + e));
          if (has_gate) val = g4[e] > 0.f ? val * p.gate_scale : 0.f;
          if (has_res) val += r4[e];
          if (has_acc) val += a4[e];
          o4[e] = val;
        }
        u32x4 w;
        w.x = __float_as_uint(o4[0]); w.y = __float_as_uint(o4[1]); w.z = __float_as_uint(o4[2]); w.w = __float_as_uint(o4[3]);
        __builtin_amdgcn_raw_buffer_store_b128(w, c_rs, offc[g], 0, 0);
      }
#ifdef MMNAS_DBG_STAMP
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      MMNAS_LIFE(4);
      continue;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * WN + j * 32 + l31;
        const bool cok = col < h.N;
        const int colc = cok ? col : h.N - 1;
        const int rbase = m0 + wm * WM + i * 32 + 4 * hh;
        if (EPI == 1) {   // LSTM forward step: column = 4 * unit + gate; the 4 lanes of a quad hold one unit's gates
          const int unit = col >> 2, gate = col & 3, Hn = h.N >> 2;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            const bool ok = cok && row < Mg;
            const int rowc = row < Mg ? row : Mg - 1;
            const float pre = acc[i][j][r] + resp[(size_t)rowc * p.ldres + colc];
            // i, f, o: logistic; g: tanh
            const float a = gate == 2 ? tanhf(pre) : 1.0f / (1.0f + __expf(-pre));
            const int q = lane & ~3;
            const float gi = __shfl(a, q, 64), gf = __shfl(a, q + 1, 64), gg = __shfl(a, q + 2, 64), go = __shfl(a, q + 3, 64);
            if (ok) {
              lg_gates[(size_t)row * h.N + col] = a;
              if (gate == 0) {
                const float c = gf * lg_cprev[(size_t)row * Hn + unit] + gi * gg;
                const float hv = go * tanhf(c);
                lg_cout[(size_t)row * Hn + unit] = c;
                Cp[(size_t)row * p.ldc + unit] = hv;
                lg_h2[(size_t)row * lg_ldh2 + unit] = hv;
              }
            }
          }
          continue;
        }
        if (EPI == 2) {   // LSTM backward step: column = unit
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            if (!(cok && row < Mg)) continue;
            const size_t o = (size_t)row * h.N + col;
            const float dh = acc[i][j][r] + resp[(size_t)row * p.ldres + col];
            const float4 g4 = *reinterpret_cast<const float4*>(lg_act + 4 * o);   // i, f, g, o (activated)
            const float c = lg_c[o], cp = lg_cprev[o];
            const float tc = tanhf(c);
            const float dc = lg_dc[o] + dh * g4.w * (1.0f - tc * tc);
            float4 d;
            d.x = dc * g4.z * g4.x * (1.0f - g4.x);        // d pre_i
            d.y = dc * cp * g4.y * (1.0f - g4.y);          // d pre_f
            d.z = dc * g4.x * (1.0f - g4.z * g4.z);        // d pre_g
            d.w = dh * tc * g4.w * (1.0f - g4.w);          // d pre_o
            *reinterpret_cast<float4*>(lg_gates + 4 * o) = d;
            lg_dc[o] = dc * g4.y;
          }
          continue;
        }
        if (atomic_out) {
          // "C +=" piece: plain adds.  (Kept free of anything that waits on memory: a bias load here made the
          // compiler drain vmcnt -- i.e. all earlier atomics -- first.)  Bias / residual ride on the piece that
          // holds the tile's first K-tile, a path the operators never take.
          if (q0 == 0 && (biasp != nullptr || has_res)) {
            const float bv0 = biasp ? biasp[colc] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = rbase + (r & 3) + 8 * (r >> 2);
              if (cok && row < Mg)
                atomicAdd(Cp + (size_t)row * p.ldc + col,
                          acc[i][j][r] * p.alpha + bv0 + (has_res ? resp[(size_t)row * p.ldres + col] : 0.f));
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = rbase + (r & 3) + 8 * (r >> 2);
              if (cok && row < Mg) atomicAdd(Cp + (size_t)row * p.ldc + col, acc[i][j][r] * p.alpha);
            }
          }
          continue;
        }
        float resv[16], gatev[16], oldv[16];
        if (has_res) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = min(rbase + (r & 3) + 8 * (r >> 2), Mg - 1);
            resv[r] = resp[(size_t)row * p.ldres + colc];
          }
        }
        if (has_gate) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = min(rbase + (r & 3) + 8 * (r >> 2), Mg - 1);
            gatev[r] = gatep[(size_t)row * p.ldgate + colc];
          }
        }
        if (has_acc) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = min(rbase + (r & 3) + 8 * (r >> 2), Mg - 1);
            oldv[r] = Cp[(size_t)row * p.ldc + colc];
          }
        }
        const float bv = biasp ? biasp[colc] : 0.f;
        float cs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          float val = acc[i][j][r] * p.alpha + bv;
          if (has_relu) val = fmaxf(val, 0.f);
          if (has_drop) val *= drop_mult(gdrop, (uint32_t)row * (uint32_t)h.N + (uint32_t)col);
          if (has_gate) val = gatev[r] > 0.f ? val * p.gate_scale : 0.f;
          if (has_res) val += resv[r];
          if (has_acc) val += oldv[r];
          if (cok && row < Mg) { Cp[(size_t)row * p.ldc + col] = val; cs += val; }
        }
        if (csp != nullptr) {   // column sums of the stored tile: the two lane halves hold the other 16 rows of a column
          cs += __shfl_xor(cs, 32, 64);
          if (hh == 0 && cok) atomicAdd(csp + col, cs);
        }
      }
    }
#ifdef MMNAS_DBG_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    MMNAS_LIFE(4);   // epilogue, stores landed
  }
}

template <int BM, int BN, bool AKC, bool BKC, bool FAST, int NS, int EPI = 0, int PF = 1, bool BDMA = false, int LEAN = 0>
__global__ void __launch_bounds__(256, NS ? (BM == 128 ? (BN == 64 ? 2 : 1) : MMNAS_OCC_NS) : (BM == 128 ? MMNAS_OCC128 : MMNAS_OCC64)) gemm_kernel(const GemmK p) {
  __shared__ __attribute__((aligned(16))) float As[GemmShape<BM, BN, NS>::A_SZ], As1[GemmShape<BM, BN, NS>::A_SZ];
  __shared__ __attribute__((aligned(16))) float Bs[GemmShape<BM, BN, NS>::B_SZ], Bs1[GemmShape<BM, BN, NS>::B_SZ];
  __shared__ int s_old;
  gemm_body<BM, BN, AKC, BKC, FAST, NS, EPI, PF, BDMA, LEAN>(p, blockIdx.x, gridDim.x, 0, As, Bs, As1, Bs1, s_old);
}

// Two independent problems in ONE launch: the data gradient (NN) and the weight gradient (TN) of a linear layer.
// A product of the workloads' sizes is a single round of workgroups that all load, compute and store in lockstep,
// so each launch pays its start (dispatch, kernel arguments, a 16 MB burst of first loads) and its end (the store
// burst, write-back of the XCD L2s) with the MFMA pipes idle: ~15 us of a 40-60 us kernel.  Here the second
// problem's workgroups are dispatched as the first problem's finish, so one of those two idle phases disappears.
// Workgroups [0, nwg0p) belong to q0 (nwg0p = nwg0 rounded up to 8, so blockIdx % 8 keeps naming the XCD in both
// sections), the rest to q1.
// Pending column reduction riding on a pair launch (AuxReduce): job = (output w, block of 16 columns); 16 lanes x 16
// row slices per workgroup.  out[w][c] += sum_rows part[row][w][c]
struct AuxReduceK {
  const float* part;
  float* out[3];
  int nrows, d, ncb, njobs;   // ncb = column blocks per output; njobs = 3 * ncb
};

__device__ __forceinline__ void aux_reduce_body(const AuxReduceK& a, int job, float* red /* >= 16*17 floats */) {
  const int w = job / a.ncb, cb = job - w * a.ncb;
  float* __restrict__ out = w == 0 ? a.out[0] : (w == 1 ? a.out[1] : a.out[2]);
  if (!out) return;
  const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int c = cb * 16 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < a.d) {
    const float* __restrict__ p = a.part + (size_t)w * a.d + c;
    const size_t rs = (size_t)3 * a.d;
    int b = g;
    for (; b + 48 < a.nrows; b += 64) {
      s0 += p[(size_t)b * rs];
      s1 += p[(size_t)(b + 16) * rs];
      s2 += p[(size_t)(b + 32) * rs];
      s3 += p[(size_t)(b + 48) * rs];
    }
    for (; b < a.nrows; b += 16) s0 += p[(size_t)b * rs];
  }
  red[g * 17 + cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && c < a.d) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i * 17 + cl];
    out[c] += t;
  }
}

template <int BM, int BN, int NS, int PF = 1, int LEAN0 = 0, int LEAN1 = 0>
__global__ void __launch_bounds__(256, NS ? MMNAS_OCC_NS : MMNAS_OCC64) gemm_pair_kernel(const GemmK q0, const GemmK q1, const int nwg0,
                                                                                    const int nwg0p, const AuxReduceK aux,
                                                                                    const int naux8) {
  __shared__ __attribute__((aligned(16))) float As[GemmShape<BM, BN, NS>::A_SZ], As1[GemmShape<BM, BN, NS>::A_SZ];
  __shared__ __attribute__((aligned(16))) float Bs[GemmShape<BM, BN, NS>::B_SZ], Bs1[GemmShape<BM, BN, NS>::B_SZ];
  __shared__ int s_old;
  int bid = blockIdx.x;
  if (bid < naux8) {   // (a multiple of 8, so blockIdx % 8 keeps naming the XCD in the sections behind it)
    if (bid < aux.njobs) aux_reduce_body(aux, bid, As);
    return;
  }
  bid -= naux8;
  if (bid < nwg0p) {
    if (bid < nwg0) gemm_body<BM, BN, true, false, true, NS, 0, PF, false, LEAN0>(q0, bid, nwg0, 0, As, Bs, As1, Bs1, s_old);
  } else {
    gemm_body<BM, BN, false, false, true, NS, 0, PF, false, LEAN1>(q1, bid - nwg0p, (int)gridDim.x - naux8 - nwg0p, (int)sizeof(GemmK), As, Bs, As1, Bs1, s_old);
  }
}

// ---- stream-K workspace: partial-tile slots + arrival counters, one per stream (launches on one stream
//      are ordered, so they can share it; two streams must not) ----
static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && e[0] ? atoi(e) : dflt;
}

// Tuning / test knobs, read from the environment once (mmnas_gemm_reload_tuning() re-reads them):
//   MMNAS_GEMM_TILE=64|128 force the tile shape      MMNAS_GEMM_GENERIC=1 force the guarded-load path
//   MMNAS_GEMM_SK=0|1|2    stream-K / split-K never, automatic, always
//   MMNAS_GEMM_WGS=n       co-resident workgroup budget (default 1024 for 64^2 tiles, 512 for 128^2)
//   MMNAS_GEMM_MIN_UNITS=n fewest K-tiles a workgroup is given (default 4)
//   MMNAS_GEMM_GM=n        row-panels per tile-order block (default 8)     MMNAS_GEMM_XCD=0 identity workgroup mapping
//   MMNAS_GEMM_SPLIT=0|1|3|6 products on the fp32 MFMA / as 1 (bf16-rounded operands: reduced precision) / 3 / 6 (default: fp32-grade)
//                          bf16 MFMA products of split operands
//   MMNAS_GEMM_PAIR=0      mmnas_gemm_pair launches its two products separately
//   MMNAS_GEMM_LEAN=0|1|2|3 bit 0: NT / NN products, bit 1: split-K TN products on the lean kernels (default 3)
//   MMNAS_GEMM_PF=1|2      K-tiles of operand loads in flight ahead of the MFMA block (64^2 fp32 buffer-load path)
struct Tuning { int tile, generic, sk, wgs, min_units, gm, xcd, split, pair, split_slots, split_p, pf, wide_min, split_minwg, hyb_t, lean, lean_maxb; bool loaded; };
static Tuning g_tune = {0, 0, 1, 0, 4, 0, 1, 3, 1, 0, 24, 2, 200, 256, 16, 3, 8 << 20, false};
static void load_tuning() {
  g_tune.tile = env_int("MMNAS_GEMM_TILE", 0);
  g_tune.generic = getenv("MMNAS_GEMM_GENERIC") != nullptr;
  g_tune.sk = env_int("MMNAS_GEMM_SK", 1);
  g_tune.wgs = env_int("MMNAS_GEMM_WGS", 0);
  g_tune.min_units = env_int("MMNAS_GEMM_MIN_UNITS", 4);
  if (g_tune.min_units < 1) g_tune.min_units = 1;
  g_tune.gm = env_int("MMNAS_GEMM_GM", 0);
  g_tune.xcd = env_int("MMNAS_GEMM_XCD", 1);
  const int sp = env_int("MMNAS_GEMM_SPLIT", 6);
  g_tune.split = sp == 3 ? 2 : (sp == 6 ? 3 : (sp == 1 ? 1 : 0));   // number of bf16 parts per operand
  g_tune.pair = env_int("MMNAS_GEMM_PAIR", 1);
  g_tune.split_slots = env_int("MMNAS_GEMM_SPLIT_SLOTS", 0);
  g_tune.split_p = env_int("MMNAS_GEMM_SPLIT_P", 24);           // K-tiles per split-K piece
  if (g_tune.split_p < 1) g_tune.split_p = 1;
  g_tune.split_minwg = env_int("MMNAS_GEMM_SPLIT_MINWG", 256);  // fewest split-K pieces in total (x tiles) when pieces of split_p would be fewer
  g_tune.pf = env_int("MMNAS_GEMM_PF", 2) == 1 ? 1 : 2;            // register stages of operand prefetch (64^2 fp32 path)
  g_tune.hyb_t = env_int("MMNAS_GEMM_HYB_T", 16);                // fewest K-tiles per output tile for the whole-tiles + streamed-tail hybrid
  g_tune.wide_min = env_int("MMNAS_GEMM_WIDE_MIN", 200);         // fewest 128x64 tiles for that shape to be chosen
  g_tune.lean = env_int("MMNAS_GEMM_LEAN", 3);                   // bit 0: lean NT / NN kernels (short set-up, 16-byte epilogue rows); bit 1: lean TN
  g_tune.lean_maxb = env_int("MMNAS_GEMM_LEAN_MAXB", 8 << 20);   // largest B matrix (bytes) the lean tile order is used for
  g_tune.loaded = true;
}

constexpr int MAX_WGS = 1024;                                  // 256 CUs x 4 workgroups of 64^2 tiles
constexpr size_t WS_SLOT_FLOATS = (size_t)2 * 512 * 128 * 128;  // 2 slots x 512 workgroups x 128^2 (= 2 x 1024 x 64^2 x 2)
constexpr int MAX_CNT_TILES = 1 << 16;

struct SkWorkspace { float* ws; int* cnt; };
static std::mutex g_ws_mu;
static std::map<std::pair<int, hipStream_t>, SkWorkspace> g_ws;

static int get_workspace(hipStream_t st, SkWorkspace* out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("gemm: hipGetDevice failed"); return MMNAS_E_LAUNCH; }
  std::lock_guard<std::mutex> lk(g_ws_mu);
  auto key = std::make_pair(dev, st);
  auto it = g_ws.find(key);
  if (it == g_ws.end()) {
    SkWorkspace w{nullptr, nullptr};
    if (hipMalloc((void**)&w.ws, WS_SLOT_FLOATS * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&w.cnt, MAX_CNT_TILES * sizeof(int)) != hipSuccess ||
        hipMemset(w.cnt, 0, MAX_CNT_TILES * sizeof(int)) != hipSuccess) {
      set_error("gemm: cannot allocate the stream-K workspace (%zu MiB)", WS_SLOT_FLOATS * sizeof(float) >> 20);
      return MMNAS_E_LAUNCH;
    }
    it = g_ws.emplace(key, w).first;
  }
  *out = it->second;
  return MMNAS_OK;
}

// the same workspace for the other kernels of the library that hand partial results between workgroups (small.hip)
int sk_workspace(hipStream_t st, float** ws, size_t* ws_floats, int** cnt, int* ncnt) {
  SkWorkspace w;
  const int rc = get_workspace(st, &w);
  if (rc) return rc;
  *ws = w.ws; *ws_floats = WS_SLOT_FLOATS; *cnt = w.cnt; *ncnt = MAX_CNT_TILES;
  return MMNAS_OK;
}

template <int BM, int BN, bool FAST, int NS, int PF = 1>
static int launch(GemmK& k, int layout, int nwg, hipStream_t st) {
  dim3 grid(nwg), block(256);
  switch (layout) {
    case MMNAS_GEMM_NT: MMNAS_LAUNCH((gemm_kernel<BM, BN, true, true, FAST, NS, 0, PF>), grid, block, 0, st, k); break;
    case MMNAS_GEMM_NN: MMNAS_LAUNCH((gemm_kernel<BM, BN, true, false, FAST, NS, 0, PF>), grid, block, 0, st, k); break;
    default: MMNAS_LAUNCH((gemm_kernel<BM, BN, false, false, FAST, NS, 0, PF>), grid, block, 0, st, k); break;
  }
  return check_launch("gemm");
}

}  // namespace mmnas

using namespace mmnas;

extern "C" int mmnas_gemm_reload_tuning(void) {
  load_tuning();
  return MMNAS_OK;
}

namespace mmnas {

struct GemmPlan {
  GemmK k;
  int nwg, layout;
  bool big, fast, wide;   // big: 128^2 tiles; wide: 128 x 64 tiles (BM x BN); neither: 64^2
  bool bdma;              // B = pre-split bf16 planes, loaded by LDS-DMA (NT, 64^2, bf16x6)
  int lean;               // the lean kernel (gemm_body<..., LEAN>): 1 whole tiles (NT / NN) or split-K pieces (TN), 2 hybrid / stream-K
  double flops, bytes;
  char tag[96];
};

static int plan_gemm(const mmnas_gemm_desc* d, hipStream_t st, GemmPlan& out) {
  if (!g_tune.loaded) load_tuning();
  MMNAS_REQUIRE(d != nullptr, MMNAS_E_ARG, "mmnas_gemm: null descriptor");
  MMNAS_REQUIRE(d->ngroups >= 1 && d->ngroups <= MAXG && d->nseg >= 1 && d->nseg <= 3, MMNAS_E_ARG,
                "mmnas_gemm: ngroups=%d nseg=%d out of range", d->ngroups, d->nseg);
  MMNAS_REQUIRE(d->layout >= 0 && d->layout <= 2, MMNAS_E_ARG, "mmnas_gemm: bad layout %d", d->layout);
  MMNAS_REQUIRE(d->N > 0 && d->K > 0, MMNAS_E_SHAPE, "mmnas_gemm: N=%d K=%d", d->N, d->K);
  const bool tn = d->layout == MMNAS_GEMM_TN;
  // split_k > 1 is the historical way to ask for "add onto C" (the split itself is now chosen here)
  const bool accumulate = d->accumulate != 0 || d->split_k > 1;
  if (accumulate) {
    MMNAS_REQUIRE(!d->relu && d->drop_p == 0.f, MMNAS_E_ARG, "mmnas_gemm: no relu/dropout epilogue when accumulating onto C");
    for (int g = 0; g < d->ngroups; ++g)
      MMNAS_REQUIRE(!d->g[g].gate, MMNAS_E_ARG, "mmnas_gemm: no gate epilogue when accumulating onto C");
  }

  GemmK& k = out.k;
  memset(&k, 0, sizeof(k));
  k.ngroups = d->ngroups; k.nseg = d->nseg; k.N = d->N; k.K = d->K;
  k.lda = d->lda; k.ldb = d->ldb; k.ldc = d->ldc; k.ldres = d->ldres; k.ldgate = d->ldgate;
  k.relu = d->relu; k.accumulate = accumulate; k.alpha = d->alpha; k.gate_scale = d->gate_scale;
  k.drop = make_drop(d->drop_p, d->drop_seed, d->drop_site);
  // vector (16-byte) operand loads need aligned bases, leading dims and extents; otherwise the
  // kernel falls back to guarded scalar loads (odd shapes such as the 3129-way answer projection)
  const bool akc = d->layout != MMNAS_GEMM_TN, bkc = d->layout == MMNAS_GEMM_NT;
  int avec = (d->lda % 4 == 0) && (akc ? d->K % 4 == 0 : 1);
  int bvec = (d->ldb % 4 == 0) && (bkc ? d->K % 4 == 0 : d->N % 4 == 0);
  double maxbytes = 0, sumM = 0;
  for (int g = 0; g < d->ngroups; ++g) {
    const mmnas_gemm_group& s = d->g[g];
    if (!akc && s.M % 4 != 0) avec = 0;
    MMNAS_REQUIRE(s.M > 0 && s.C, MMNAS_E_ARG, "mmnas_gemm: group %d M=%d C=%p", g, s.M, (void*)s.C);
    k.g[g].M = s.M; k.g[g].C = s.C; k.g[g].bias = s.bias; k.g[g].residual = s.residual; k.g[g].gate = s.gate;
    k.g[g].colsum = s.colsum;
    {
      const DropCfg gd = s.drop_seed ? make_drop(d->drop_p, s.drop_seed, d->drop_site) : k.drop;
      k.g[g].seed_lo = gd.seed_lo; k.g[g].site_key = gd.site_key;
    }
    MMNAS_REQUIRE(!(s.colsum && accumulate), MMNAS_E_ARG, "mmnas_gemm: no column sums when accumulating onto C");
    for (int i = 0; i < 3; ++i) {
      k.g[g].A[i] = s.A[i]; k.g[g].B[i] = s.B[i];
      if (i < d->nseg) {
        MMNAS_REQUIRE(s.A[i] && s.B[i], MMNAS_E_ARG, "mmnas_gemm: group %d segment %d null operand", g, i);
        if (((uintptr_t)s.A[i] & 15) != 0) avec = 0;
        if (((uintptr_t)s.B[i] & 15) != 0) bvec = 0;
      }
    }
    sumM += s.M;
    const double ab = 4.0 * (akc ? (double)(s.M + 128) : (double)d->K) * d->lda;
    if (ab > maxbytes) maxbytes = ab;
  }
  const double bb = 4.0 * (bkc ? (double)(d->N + 128) : (double)d->K) * d->ldb;
  if (bb > maxbytes) maxbytes = bb;
  k.avec = avec; k.bvec = bvec;
  // branch-free buffer-load path: aligned vector loads, K a multiple of the K tile, 32-bit byte offsets
  // (TN: both operands are [K][rows]; a K that is no multiple of the K-tile simply ends inside the last tile, whose rows
  //  behind K lie outside the operands' buffer ranges and read as zero -- the row count of a PACKED ragged batch is
  //  arbitrary, and it is the reduction length of every weight gradient)
  bool fast = avec && bvec && (d->K % BK == 0 || tn) && maxbytes < 4.0e9;
  if (g_tune.generic) fast = false;
  k.ntk = cdiv(d->K, BK);
  k.T = k.ntk * d->nseg;

  // ---- schedule: tile shape, K-slices S, units per workgroup P ----
  // Measured on the VQA shapes (tools/gemm_bench.py, profiles/):
  //   * 128^2 tiles move half the LDS / L2 bytes per flop of 64^2 ones (123-134 vs 96-119 TF/s) but only pay
  //     once there are >= 4 full waves of them; below that 64^2 tiles, 4 workgroups per CU.
  //   * whole tiles, one per workgroup (P = T), whenever they fill the CUs evenly: the dispatcher balances
  //     them dynamically and a finishing workgroup's epilogue overlaps its successor's prologue.
  //   * stream-K (P not a multiple of T) when whole tiles would leave the CUs unevenly loaded and there is
  //     enough reduction per tile to amortise the hand-over of partial tiles (~5 us exposed at the end):
  //     the M = 896 (question-side) products, and K >= 1536 products on M = 6400.
  //   * "C +=" products (weight gradients: few tiles, reduction over the 6400 rows): plain split-K, the
  //     pieces added with float atomics, dealt out slice-major (an XCD's workgroups add into different
  //     tiles and stream the same K-slice of both operands through its L2).
  auto ntiles_for = [&](int bt) {
    long n = 0;
    for (int g = 0; g < d->ngroups; ++g) n += (long)cdiv(d->g[g].M, bt) * cdiv(d->N, bt);
    return n;
  };
  const int min_units = g_tune.min_units;
  bool big = ntiles_for(128) >= 2048;
  if (g_tune.tile == 128) big = true;
  if (g_tune.tile == 64 || g_tune.tile == 12864) big = false;
  // 128 x 64 tiles (each wave two 32x32 MFMA tiles down the rows): 1.33x the products per operand byte and per barrier
  // of 64^2 -- for plain products on the buffer-load path whose rows fill the chip anyway (experiment: MMNAS_GEMM_TILE=12864)
  bool wide = !big && g_tune.tile == 12864 && !accumulate && fast && g_tune.split != 2 && g_tune.split != 1;
  if (wide) {
    long n = 0;
    for (int g = 0; g < d->ngroups; ++g) n += (long)cdiv(d->g[g].M, 128) * cdiv(d->N, 64);
    if (n < g_tune.wide_min) wide = false;
  }
  const int bt = big ? 128 : 64;            // tile height (rows) unless wide
  const int btm = wide ? 128 : bt, btn = bt;
  k.tiles_n = cdiv(d->N, btn);
  long t0 = 0;
  for (int g = 0; g < MAXG; ++g) k.gtile0[g] = INT_MAX;
  for (int g = 0; g < d->ngroups; ++g) { k.g[g].tile0 = k.gtile0[g] = (int)t0; t0 += (long)cdiv(d->g[g].M, btm) * k.tiles_n; }
  MMNAS_REQUIRE(t0 < (1l << 30), MMNAS_E_SHAPE, "mmnas_gemm: too many output tiles");
  k.ntiles = (int)t0;
  k.gm = g_tune.gm > 0 ? g_tune.gm : 8;
  k.xcd_remap = g_tune.xcd;
  // co-resident workgroups: 256 CUs x 2 (128^2 tiles: 72 KB LDS each) or x 4
  const int slots = g_tune.wgs > 0 ? (g_tune.wgs < MAX_WGS ? g_tune.wgs : MAX_WGS) : ((big || wide) ? 512 : 1024);
  // how evenly whole tiles load the 256 CUs: mean / max tiles per CU (the dispatcher balances dynamically)
  const double per_cu = (double)k.ntiles / 256.0;
  const double dp_eff = per_cu / (double)(long)(per_cu + 0.999999);
  const long long U = (long long)k.ntiles * k.T;
  int sk = g_tune.sk;  // 0 never, 1 auto, 2 always
  k.mode = MODE_TILE;
  k.P = k.T;
  int nwg = k.ntiles;
  if (accumulate && sk != 0) {
    // split-K with atomics.  Pieces of ~split_p K-tiles (long enough to amortise a piece's operand prologue and its
    // 16 KB of atomic adds, short enough to balance), but at least ~2 workgroups per CU in total, >= min_units K-tiles each.
    // (MMNAS_GEMM_SPLIT_SLOTS=n restores the older rule: as many pieces as fit n co-resident slots.)
    int want;
    if (g_tune.split_slots > 0) want = g_tune.split_slots / k.ntiles;
    else {
      want = (k.T + g_tune.split_p / 2) / g_tune.split_p;
      if ((long)want * k.ntiles < g_tune.split_minwg) want = (g_tune.split_minwg + k.ntiles - 1) / k.ntiles;
    }
    if (want < 1) want = 1;
    if (want > k.T / min_units) want = k.T / min_units;
    if (want > 1) {
      k.P = (k.T + want - 1) / want;
      const int S = (k.T + k.P - 1) / k.P;
      MMNAS_REQUIRE((long long)S * k.ntiles < (1ll << 30), MMNAS_E_SHAPE, "mmnas_gemm: too many split-K pieces");
      k.mode = MODE_SPLIT;
      nwg = S * k.ntiles;
    }
  } else if (!accumulate && sk != 0 && !big && !wide && U < (1ll << 30) && k.ntiles <= MAX_CNT_TILES && g_tune.wgs == 0 &&
             sk == 1 && k.ntiles > 256 && k.ntiles < 8192 && k.ntiles % 256 != 0 && k.T >= g_tune.hyb_t) {
    // Single round (every tile resident at once, 3-4 workgroups per CU) with a ragged last "layer": the first
    // floor(ntiles / 256) * 256 tiles are computed whole; the R tail tiles are streamed by ~one extra SHORT workgroup
    // per CU (R * T units cut into <= 256 runs), launched first.  Their hand-over through the workspace ends long
    // before the whole tiles do, so -- unlike streaming everything -- it costs nothing, the whole tiles keep
    // their K-phases aligned (L2 reuse of the A-panels), and every CU carries the same number of K-tiles.
    const int n_full = (k.ntiles / 256) * 256, R = k.ntiles - n_full;
    const long long Ut = (long long)R * k.T;
    long long P = (Ut + 255) / 256;
    if (P < 2) P = 2;
    int n_sk = (int)((Ut + P - 1) / P);
    n_sk = (n_sk + 7) / 8 * 8;
    k.mode = MODE_STREAM;
    k.n_full = n_full; k.full_per = n_full / 8; k.sk_per = n_sk / 8;
    k.P = (int)P;
    k.U = (int)Ut;
    nwg = n_full + n_sk;
  } else if (!accumulate && sk != 0 && U < (1ll << 30) && k.ntiles <= MAX_CNT_TILES && k.T >= 2 * min_units &&
             (sk == 2 || (!big && !wide && ((dp_eff < 0.6 && k.T >= 16) || (dp_eff < 0.9 && k.T >= 48))))) {
    // (runs of >= 8 K-tiles when the choice is automatic: fewer contributors per tile in the hand-over, measured
    //  5-20 % faster than 4 on the M = 896 products)
    const int smu = sk == 2 ? min_units : 2 * min_units;
    long long G = U / smu;
    if (G > slots) G = slots;
    if (G < 1) G = 1;
    const long long P = (U + G - 1) / G;
    if (P < k.T) {  // (a whole tile or more each: launch whole tiles instead)
      k.mode = MODE_STREAM;
      k.P = (int)P;
      k.U = (int)U;
      nwg = (int)((U + P - 1) / P);
    }
  }
  if (k.mode == MODE_STREAM) {
    SkWorkspace w;
    const int rc = get_workspace(st, &w);
    if (rc) return rc;
    k.ws = w.ws; k.cnt = w.cnt;
  }
  // algorithmic work: 2*M*N*K flops per product; minimum traffic = operands once + result once
  out.tag[0] = 0;
  if (prof_enabled())
    snprintf(out.tag, sizeof(out.tag), "%s M=%d/%d/%d N=%d K=%d seg=%d t%d wg=%d P=%d/%d %s%s", tn ? "TN" : (bkc ? "NT" : "NN"),
             d->g[0].M, d->ngroups > 1 ? d->g[1].M : 0, d->ngroups > 2 ? d->g[2].M : 0, d->N, d->K, d->nseg, wide ? 12864 : bt, nwg, k.P, k.T,
             k.mode == MODE_TILE ? "tile" : (k.mode == MODE_SPLIT ? "split" : (k.n_full ? "hybrid" : "stream")),
             fast ? (g_tune.split == 2 ? " bf16x3" : (g_tune.split == 3 ? " bf16x6" : (g_tune.split == 1 ? " bf16x1" : ""))) : " generic");
  out.flops = 2.0 * sumM * d->N * d->K * d->nseg;
  out.bytes = 4.0 * (sumM * d->K * d->nseg + (double)d->N * d->K * d->nseg * d->ngroups + sumM * d->N);
  out.nwg = nwg; out.layout = d->layout; out.big = big; out.fast = fast; out.wide = wide;
  out.bdma = d->b_planes != 0;
  if (out.bdma) {
    // B[i] of every group = mmnas_split_planes output ([3][N][ldb] bf16).  One kernel shape exists for it; anything else
    // would read the planes as fp32: refuse.
    MMNAS_REQUIRE(d->layout == MMNAS_GEMM_NT && fast && !big && !wide && !accumulate && g_tune.split == 3 && d->N % 64 == 0 &&
                  (d->ldb % 8) == 0, MMNAS_E_ARG,
                  "mmnas_gemm: b_planes needs layout NT, 64^2 tiles on the buffer-load path, N %% 64 == 0, ldb %% 8 == 0, MMNAS_GEMM_SPLIT=6 "
                  "(N=%d K=%d ldb=%d)", d->N, d->K, d->ldb);
    if (out.tag[0]) strncat(out.tag, " Bdma", sizeof(out.tag) - strlen(out.tag) - 1);
  }
  // ---- the lean kernel: whole tiles of an NT / NN product on the default split-operand path whose epilogue operands can be
  //      moved as 16-byte rows and whose B matrix an XCD's L2 holds beside the A panels in flight ----
  out.lean = 0;
  const bool lean_base = fast && !big && !wide && !out.bdma && g_tune.split == 3 && g_tune.pf == 2 &&
                         (k.tiles_n & (k.tiles_n - 1)) == 0 && d->ldc % 4 == 0;
  if (lean_base && (g_tune.lean & 1) && !tn && k.mode != MODE_SPLIT && d->N % 4 == 0 && 4.0 * d->N * d->K * d->nseg <= (double)g_tune.lean_maxb) {
    bool ok = true, any_colsum = false;
    for (int g = 0; g < d->ngroups && ok; ++g) {
      const mmnas_gemm_group& s = d->g[g];
      const double rows = (double)s.M + 1.0;
      if (s.colsum) any_colsum = true;
      ok = ((uintptr_t)s.C & 15) == 0 && rows * d->ldc * 4.0 < 4.0e9 &&
           (!s.bias || ((uintptr_t)s.bias & 15) == 0) &&
           (!s.residual || (((uintptr_t)s.residual & 15) == 0 && d->ldres % 4 == 0 && rows * d->ldres * 4.0 < 4.0e9)) &&
           (!s.gate || (((uintptr_t)s.gate & 15) == 0 && d->ldgate % 4 == 0 && rows * d->ldgate * 4.0 < 4.0e9));
    }
    out.lean = ok ? (k.mode == MODE_TILE ? 1 : 2) : 0;
    // column sums: the element-wise lean form (NN, whole tiles) or the general kernel
    if (any_colsum) out.lean = (ok && k.mode == MODE_TILE && d->layout == MMNAS_GEMM_NN) ? 3 : 0;
  } else if (lean_base && tn && k.mode == MODE_SPLIT && d->nseg == 1 && (g_tune.lean & 2) && k.ntiles >= 2 && k.ntiles < 65536 && nwg < 65536) {
    // weight gradients: split-K pieces added by buffer atomics; nothing else rides on them
    bool ok = true;
    for (int g = 0; g < d->ngroups && ok; ++g) {
      const mmnas_gemm_group& s = d->g[g];
      ok = !s.bias && !s.residual && !s.gate && !s.colsum && ((double)s.M + 1.0) * d->ldc * 4.0 < 4.0e9;
    }
    out.lean = ok ? 1 : 0;
    if (ok) k.nt_magic = (unsigned)(((1ull << 32) + (unsigned)k.ntiles - 1) / (unsigned)k.ntiles);
  }
  if (out.lean) {
    int sh = 0;
    while ((1 << sh) < k.tiles_n) ++sh;
    k.tn_shift = sh;
    if (out.tag[0]) strncat(out.tag, " lean", sizeof(out.tag) - strlen(out.tag) - 1);
  }
  return MMNAS_OK;
}

static int launch_plan(GemmPlan& pl, hipStream_t st) {
  GemmK& k = pl.k;
  ProfScope ps(MMNAS_K_GEMM, pl.flops, pl.bytes, st, pl.tag);
  const int ns = pl.fast ? g_tune.split : 0;   // (odd shapes on the guarded-load path stay on the fp32 MFMA)
  if (pl.bdma) {
    MMNAS_LAUNCH((gemm_kernel<64, 64, true, true, true, 3, 0, 1, true>), dim3(pl.nwg), dim3(256), 0, st, k);
    return check_launch("gemm");
  }
  if (pl.lean) {
    const dim3 grid(pl.nwg), block(256);
    if (pl.layout == MMNAS_GEMM_NT) {
      if (pl.lean == 1) MMNAS_LAUNCH((gemm_kernel<64, 64, true, true, true, 3, 0, 2, false, 1>), grid, block, 0, st, k);
      else MMNAS_LAUNCH((gemm_kernel<64, 64, true, true, true, 3, 0, 2, false, 2>), grid, block, 0, st, k);
    } else if (pl.layout == MMNAS_GEMM_NN) {
      if (pl.lean == 3) MMNAS_LAUNCH((gemm_kernel<64, 64, true, false, true, 3, 0, 2, false, 3>), grid, block, 0, st, k);
      else if (pl.lean == 1) MMNAS_LAUNCH((gemm_kernel<64, 64, true, false, true, 3, 0, 2, false, 1>), grid, block, 0, st, k);
      else MMNAS_LAUNCH((gemm_kernel<64, 64, true, false, true, 3, 0, 2, false, 2>), grid, block, 0, st, k);
    } else MMNAS_LAUNCH((gemm_kernel<64, 64, false, false, true, 3, 0, 2, false, 1>), grid, block, 0, st, k);
    return check_launch("gemm");
  }
  if (pl.big) {
    if (ns == 1) return launch<128, 128, true, 1>(k, pl.layout, pl.nwg, st);
    if (ns == 2) return launch<128, 128, true, 2>(k, pl.layout, pl.nwg, st);
    if (ns == 3) return launch<128, 128, true, 3>(k, pl.layout, pl.nwg, st);
    return pl.fast ? launch<128, 128, true, 0>(k, pl.layout, pl.nwg, st) : launch<128, 128, false, 0>(k, pl.layout, pl.nwg, st);
  }
  if (pl.wide) return ns == 3 ? launch<128, 64, true, 3>(k, pl.layout, pl.nwg, st) : launch<128, 64, true, 0, 2>(k, pl.layout, pl.nwg, st);
  if (ns == 1) return launch<64, 64, true, 1, 2>(k, pl.layout, pl.nwg, st);
  if (ns == 2) return launch<64, 64, true, 2>(k, pl.layout, pl.nwg, st);
  if (ns == 3) return g_tune.pf == 2 ? launch<64, 64, true, 3, 2>(k, pl.layout, pl.nwg, st) : launch<64, 64, true, 3>(k, pl.layout, pl.nwg, st);
  if (pl.fast && g_tune.pf == 2) return launch<64, 64, true, 0, 2>(k, pl.layout, pl.nwg, st);
  return pl.fast ? launch<64, 64, true, 0>(k, pl.layout, pl.nwg, st) : launch<64, 64, false, 0>(k, pl.layout, pl.nwg, st);
}

}  // namespace mmnas

extern "C" int mmnas_gemm(const mmnas_gemm_desc* d, void* stream) {
  GemmPlan pl;
  const int rc = plan_gemm(d, (hipStream_t)stream, pl);
  if (rc) return rc;
  return launch_plan(pl, (hipStream_t)stream);
}

namespace mmnas {
int gemm_pair_aux(const mmnas_gemm_desc* dgrad, const mmnas_gemm_desc* wgrad, const AuxReduce* aux, hipStream_t st) {
  GemmPlan p0, p1;
  int rc;
  if ((rc = plan_gemm(dgrad, st, p0))) return rc;
  if ((rc = plan_gemm(wgrad, st, p1))) return rc;
  // one launch when both run on the 64^2 buffer-load kernel and the second one needs no workspace of its own
  const bool pair = g_tune.pair && p0.fast && p1.fast && !p0.big && !p1.big && !p0.wide && !p1.wide && p0.layout == MMNAS_GEMM_NN &&
                    p1.layout == MMNAS_GEMM_TN && p1.k.mode != MODE_STREAM;
  if (!pair) {
    if (aux && (rc = launch_aux_reduce(*aux, st))) return rc;
    if ((rc = launch_plan(p0, st))) return rc;
    return launch_plan(p1, st);
  }
  AuxReduceK ak;
  memset(&ak, 0, sizeof(ak));
  if (aux && aux->part && g_tune.pair != 3) {   // (MMNAS_GEMM_PAIR=3: pending reductions get their own launch)
    ak.part = aux->part; ak.nrows = aux->nrows; ak.d = aux->d;
    for (int i = 0; i < 3; ++i) ak.out[i] = aux->out[i];
    ak.ncb = cdiv(aux->d, 16);
    ak.njobs = 3 * ak.ncb;
  } else if (aux && (rc = launch_aux_reduce(*aux, st))) return rc;
  const int naux8 = (ak.njobs + 7) / 8 * 8;
  char tag[96] = "";
  if (prof_enabled()) snprintf(tag, sizeof(tag), "PAIR %.44s | %.40s", p0.tag, p1.tag);
  ProfScope ps(MMNAS_K_GEMM, p0.flops + p1.flops, p0.bytes + p1.bytes, st, tag);
  const int nwg0p = (p0.nwg + 7) / 8 * 8;
  dim3 grid(naux8 + nwg0p + p1.nwg), block(256);
  switch (g_tune.split) {
    case 1: MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 1, 2>), grid, block, 0, st, p0.k, p1.k, p0.nwg, nwg0p, ak, naux8); break;
    case 2: MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 2>), grid, block, 0, st, p0.k, p1.k, p0.nwg, nwg0p, ak, naux8); break;
    case 3:
      if (p0.lean == 3 && p1.lean) MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 3, 2, 3, 1>), grid, block, 0, st, p0.k, p1.k, p0.nwg, nwg0p, ak, naux8);
      else if (p0.lean == 3) MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 3, 2, 3, 0>), grid, block, 0, st, p0.k, p1.k, p0.nwg, nwg0p, ak, naux8);
      else if (p0.lean == 1 && p1.lean) MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 3, 2, 1, 1>), grid, block, 0, st, p0.k, p1.k, p0.nwg, nwg0p, ak, naux8);
      else if (p0.lean == 2 && p1.lean) MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 3, 2, 2, 1>), grid, block, 0, st, p0.k, p1.k, p0.nwg, nwg0p, ak, naux8);
      else if (p0.lean) MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 3, 2, 2, 0>), grid, block, 0, st, p0.k, p1.k, p0.nwg, nwg0p, ak, naux8);
      else if (p1.lean) MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 3, 2, 0, 1>), grid, block, 0, st, p0.k, p1.k, p0.nwg, nwg0p, ak, naux8);
      else if (g_tune.pf == 2) MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 3, 2>), grid, block, 0, st, p0.k, p1.k, p0.nwg, nwg0p, ak, naux8);
      else MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 3>), grid, block, 0, st, p0.k, p1.k, p0.nwg, nwg0p, ak, naux8);
      break;
    default:
      if (g_tune.pf == 2) MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 0, 2>), grid, block, 0, st, p0.k, p1.k, p0.nwg, nwg0p, ak, naux8);
      else MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 0>), grid, block, 0, st, p0.k, p1.k, p0.nwg, nwg0p, ak, naux8);
      break;
  }
  return check_launch("gemm_pair");
}
}  // namespace mmnas

namespace mmnas {
// A weight-gradient product (TN, split-K) with a pending column reduction riding on its launch as a few extra workgroups:
// gemm_pair_aux without the data-gradient half (the short-sequence backward of small.hip leaves no NN product to pair with).
int gemm_wgrad_aux(const mmnas_gemm_desc* wgrad, const AuxReduce* aux, hipStream_t st) {
  GemmPlan p1;
  int rc;
  if ((rc = plan_gemm(wgrad, st, p1))) return rc;
  const bool ride = aux && aux->part && g_tune.pair && g_tune.pair != 3 && g_tune.split == 3 && g_tune.pf == 2 && p1.fast && !p1.big && !p1.wide &&
                    p1.layout == MMNAS_GEMM_TN && p1.k.mode != MODE_STREAM;
  if (!ride) {
    if (aux && (rc = launch_aux_reduce(*aux, st))) return rc;
    return launch_plan(p1, st);
  }
  AuxReduceK ak;
  memset(&ak, 0, sizeof(ak));
  ak.part = aux->part; ak.nrows = aux->nrows; ak.d = aux->d;
  for (int i = 0; i < 3; ++i) ak.out[i] = aux->out[i];
  ak.ncb = cdiv(aux->d, 16);
  ak.njobs = 3 * ak.ncb;
  const int naux8 = (ak.njobs + 7) / 8 * 8;
  char tag[96] = "";
  if (prof_enabled()) snprintf(tag, sizeof(tag), "WGRAD+AUX %.60s", p1.tag);
  ProfScope ps(MMNAS_K_GEMM, p1.flops, p1.bytes, st, tag);
  dim3 grid(naux8 + p1.nwg), block(256);
  // (the first descriptor is not run -- 0 workgroups -- but keeps the second one at its kernel-argument offset)
  if (p1.lean) MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 3, 2, 0, 1>), grid, block, 0, st, p1.k, p1.k, 0, 0, ak, naux8);
  else MMNAS_LAUNCH((gemm_pair_kernel<64, 64, 3, 2>), grid, block, 0, st, p1.k, p1.k, 0, 0, ak, naux8);
  return check_launch("gemm_wgrad_aux");
}
}  // namespace mmnas

extern "C" int mmnas_gemm_pair(const mmnas_gemm_desc* dgrad, const mmnas_gemm_desc* wgrad, void* stream) {
  return gemm_pair_aux(dgrad, wgrad, nullptr, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// Weight matrices as three bf16 planes (h, m, l of split_pair: x = h + m + l exactly): what gemm_kernel<..., BDMA> streams
// into LDS by LDS-DMA.  Weights change once per optimizer step; the split is the same instruction sequence as the
// in-kernel one, so a product on the planes equals the product on the fp32 matrix bit for bit.
// ------------------------------------------------------------------------------------------
namespace mmnas {
__global__ void __launch_bounds__(256) split_planes_kernel(const float4* __restrict__ w, uint2* __restrict__ p0, uint2* __restrict__ p1,
                                                           uint2* __restrict__ p2, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = w[i];
    unsigned a0, a1, a2, b0, b1, b2;
    split_pair<3>(v.x, v.y, a0, a1, a2);
    split_pair<3>(v.z, v.w, b0, b1, b2);
    p0[i] = make_uint2(a0, b0);
    p1[i] = make_uint2(a1, b1);
    p2[i] = make_uint2(a2, b2);
  }
}
}  // namespace mmnas

extern "C" int mmnas_split_planes(const float* w, void* planes, size_t n, void* stream) {
  MMNAS_REQUIRE(w && planes, MMNAS_E_ARG, "split_planes: null pointer");
  MMNAS_REQUIRE(n > 0 && n % 8 == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)planes & 15) == 0, MMNAS_E_SHAPE,
                "split_planes: n = %zu must be a multiple of 8 and both buffers 16-byte aligned", n);
  const size_t n4 = n / 4;
  unsigned short* pl = (unsigned short*)planes;
  const int blocks = (int)std::min<size_t>((n4 + 255) / 256, 2048);
  MMNAS_LAUNCH(split_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float4*)w, (uint2*)pl, (uint2*)(pl + n),
               (uint2*)(pl + 2 * n), n4);
  return check_launch("split_planes");
}

// ------------------------------------------------------------------------------------------
// LSTM (single layer, zero initial state) on the step-fused GEMM epilogues above.  Replaces the MIOpen path of
// nn.LSTM (hygr_vqa.py:106-107 `self.lstm(lang_feat)`), which issues a GEMM + a pointwise kernel per time step and
// direction plus weight-buffer copies: ~110 launches per training step for 14 steps of a 64-row problem.
// Layouts: time-major buffers; gate columns interleaved (column 4j+g = gate g of unit j, g in i,f,g,o) -- the host
// side permutes nn.LSTM's [i|f|g|o] row blocks accordingly.
// ------------------------------------------------------------------------------------------
namespace mmnas {
// units_per_wg: 0 = whole tiles, one per workgroup; > 0 = stream-K runs of that many K-tiles
static int lstm_step_plan(GemmPlan& pl, const mmnas_gemm_desc& d, hipStream_t st, const char* who, int units_per_wg) {
  int rc = plan_gemm(&d, st, pl);
  if (rc) return rc;
  MMNAS_REQUIRE(pl.fast && !pl.big && !pl.wide, MMNAS_E_SHAPE, "%s: shape outside the step kernel's range (hidden size %% 32 == 0, aligned buffers)", who);
  GemmK& k = pl.k;
  k.n_full = k.full_per = k.sk_per = 0;
  if (units_per_wg <= 0 || units_per_wg >= k.T) {
    k.mode = MODE_TILE; k.P = k.T; pl.nwg = k.ntiles;
  } else {
    SkWorkspace w;
    if ((rc = get_workspace(st, &w))) return rc;
    k.ws = w.ws; k.cnt = w.cnt;
    k.mode = MODE_STREAM; k.P = units_per_wg; k.U = k.ntiles * k.T;
    pl.nwg = (k.U + k.P - 1) / k.P;
  }
  return MMNAS_OK;
}
}  // namespace mmnas

extern "C" int mmnas_lstm_supported(int E, int H) { return E > 0 && H >= 32 && H % 32 == 0 && H <= 4096; }

extern "C" int mmnas_lstm_fwd(const float* x_tm, const float* Wih, const float* Whh, const float* bias, float* xp, float* Hall,
                              float* Call, float* Gall, float* out, int T, int B, int E, int H, void* stream) {
  MMNAS_REQUIRE(x_tm && Wih && Whh && bias && xp && Hall && Call && Gall && out, MMNAS_E_ARG, "lstm_fwd: null pointer");
  MMNAS_REQUIRE(T > 0 && B > 0 && mmnas_lstm_supported(E, H), MMNAS_E_SHAPE, "lstm_fwd: T=%d B=%d E=%d H=%d", T, B, E, H);
  hipStream_t st = (hipStream_t)stream;
  mmnas_gemm_desc d;
  // input projection of all time steps: xp[T*B, 4H] = x W_ih^T + (b_ih + b_hh)
  memset(&d, 0, sizeof(d));
  d.layout = MMNAS_GEMM_NT; d.ngroups = 1; d.nseg = 1; d.N = 4 * H; d.K = E; d.lda = E; d.ldb = E; d.ldc = 4 * H;
  d.alpha = 1.f; d.gate_scale = 1.f; d.split_k = 1;
  d.g[0].M = T * B; d.g[0].A[0] = x_tm; d.g[0].B[0] = Wih; d.g[0].C = xp; d.g[0].bias = bias;
  int rc = mmnas_gemm(&d, stream);
  if (rc) return rc;
  const size_t bh = (size_t)B * H, bg = (size_t)B * 4 * H;
  for (int t = 0; t < T; ++t) {
    memset(&d, 0, sizeof(d));
    d.layout = MMNAS_GEMM_NT; d.ngroups = 1; d.nseg = 1; d.N = 4 * H; d.K = H; d.lda = H; d.ldb = H; d.ldc = H;
    d.ldres = 4 * H; d.alpha = 1.f; d.gate_scale = 1.f; d.split_k = 1;
    d.g[0].M = B; d.g[0].A[0] = Hall + t * bh; d.g[0].B[0] = Whh; d.g[0].C = Hall + (t + 1) * bh; d.g[0].residual = xp + t * bg;
    GemmPlan pl;
    if ((rc = lstm_step_plan(pl, d, st, "lstm_fwd", env_int("MMNAS_LSTM_FWD_P", 0)))) return rc;
    pl.k.lg_cprev = Call + t * bh; pl.k.lg_cout = Call + (t + 1) * bh; pl.k.lg_gates = Gall + t * bg;
    pl.k.lg_h2 = out + (size_t)t * H; pl.k.lg_ldh2 = T * H;
    ProfScope ps(MMNAS_K_GEMM, pl.flops, pl.bytes, st, pl.tag);
    MMNAS_LAUNCH((gemm_kernel<64, 64, true, true, true, 0, 1>), dim3(pl.nwg), dim3(256), 0, st, pl.k);
  }
  return check_launch("lstm_fwd");
}

extern "C" int mmnas_lstm_bwd(const float* dout, const float* Whh, const float* Call, const float* Gall, float* DG, float* dc,
                              float* scratch, int T, int B, int H, void* stream) {
  MMNAS_REQUIRE(dout && Whh && Call && Gall && DG && dc && scratch, MMNAS_E_ARG, "lstm_bwd: null pointer");
  MMNAS_REQUIRE(T > 0 && B > 0 && mmnas_lstm_supported(1, H), MMNAS_E_SHAPE, "lstm_bwd: T=%d B=%d H=%d", T, B, H);
  hipStream_t st = (hipStream_t)stream;
  const size_t bh = (size_t)B * H, bg = (size_t)B * 4 * H;
  for (int t = T - 1; t >= 0; --t) {
    mmnas_gemm_desc d;
    memset(&d, 0, sizeof(d));
    // dh_t = dout[:, t] + dG_{t+1} W_hh     [B, 4H] x [4H, H]
    d.layout = MMNAS_GEMM_NN; d.ngroups = 1; d.nseg = 1; d.N = H; d.K = 4 * H; d.lda = 4 * H; d.ldb = H; d.ldc = H;
    d.ldres = T * H; d.alpha = 1.f; d.gate_scale = 1.f; d.split_k = 1;
    d.g[0].M = B; d.g[0].A[0] = DG + (t + 1) * bg; d.g[0].B[0] = Whh; d.g[0].C = scratch; d.g[0].residual = dout + (size_t)t * H;
    GemmPlan pl;
    int rc;
    if ((rc = lstm_step_plan(pl, d, st, "lstm_bwd", env_int("MMNAS_LSTM_BWD_P", 8)))) return rc;
    pl.k.lg_act = Gall + t * bg; pl.k.lg_c = Call + (t + 1) * bh; pl.k.lg_cprev = Call + t * bh;
    pl.k.lg_dc = dc; pl.k.lg_gates = DG + t * bg;
    ProfScope ps(MMNAS_K_GEMM, pl.flops, pl.bytes, st, pl.tag);
    MMNAS_LAUNCH((gemm_kernel<64, 64, true, false, true, 0, 2>), dim3(pl.nwg), dim3(256), 0, st, pl.k);
  }
  return check_launch("lstm_bwd");
}

#ifdef MMNAS_DBG_STAMP
extern "C" int mmnas_dbg_stamps(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, reset & 2 ? HIP_SYMBOL(mmnas::g_life_acc) : HIP_SYMBOL(mmnas::g_stamp_acc), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset & 1) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(reset & 2 ? HIP_SYMBOL(mmnas::g_life_acc) : HIP_SYMBOL(mmnas::g_stamp_acc), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#endif

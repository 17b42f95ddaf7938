// Attention-core backward (MHAtt.att / RelMHAtt.forward, modules.py:191-199, 231-241, and their autograd) for d_h = 64 and
// at most 128 queries / keys -- the image stream's attention of every workload -- with all FIVE products of a tile on the
// bf16 matrix pipe: each fp32 product as six v_mfma_f32_32x32x16_bf16 products of exactly split operands (x = h + m + l, three
// bf16 parts hold all 24 mantissa bits; fp32 accumulate; dropped cross terms <= 2^-23 |a||b|), as gemm.hip multiplies.
// Round 5's fused kernel (attention.hip, mha_bwd_fused_kernel) runs the same five products on v_mfma_f32_32x32x2_f32, 1/16
// of the bf16 instruction rate: 640 MFMAs x 64 cycles = 17 us of its 45 us are matrix-pipe time at one wave per SIMD.
//
// Decomposition (one workgroup = one (batch, head), 4 waves, one per SIMD):
//   * a wave OWNS a block of 32 KEYS: its K / V rows live in registers as pre-split MFMA fragments for the whole kernel,
//     and dK^T / dV^T of those keys accumulate in its registers over the query blocks -- no LDS accumulators, no cross-wave
//     sum for them (round 5: waves own query blocks and meet in 64 KB of LDS accumulators, which left no room for split
//     operand images).  At step s wave w works on query block (w + s) mod 4: the waves of a step touch different blocks.
//   * Q and dO of the head sit in LDS ONCE, as ONE split image each (row = three runs of 64 bf16): read by rows
//     (ds_read_b128) as the B operands of S^T = K Q^T and dA^T = V dO^T, and by columns through the hardware transpose read
//     (ds_read_b64_tr_b16) as the A operands of dK^T += Q^T dS and dV^T += dO^T A.
//   * tile orientation as in round 5: key = accumulator row, QUERY = lane -- the relation bias and its gradient are
//     key-major [B,H,Sk,Sq], so lanes read / write them coalesced -- softmax backward in registers; dS^T is directly the A
//     operand of dQ += dS K (its B operand: the wave's own K block, transposed fragments in registers); the two tiles cross
//     a wave-private 4 KB LDS image once to put the key on the lane for the dK^T / dV^T products.
//   * dQ of a query block collects the four waves' contributions in an LDS accumulator [128][64] fp32 (32 KB): a barrier
//     per step orders the read-modify-writes (different blocks inside a step).
// LDS: 2 x 51.2 KB images + 32 KB + 16.9 KB + row statistics = 152 KB: one workgroup per CU.
#include <string.h>
#include "common.h"
#include "gemm_split.h"

namespace mmnas {



typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int B16_RSB = 400;         // bytes per image row: 3 parts x 128 B + 16 B pad (100 words: ds_read_b128 rows conflict-free)
// The images of both kernels PERMUTE the rows inside every group of 16: logical row 8 a + 4 b + q (the four rows q of one
// transposed read share a, b) sits at physical row 4 q + 2 a + b.  The four rows of a transposed read are then 4 physical
// rows = 400 words = 16 banks apart -- conflict-free, where consecutive rows overlap by half (conflict fractions as first
// written: 0.147 backward / 0.24 forward; now 0.000, profiles/r06_attention_b16_lds.txt) -- and a row read still covers the
// 16 rows of a group.  No extra LDS, no arithmetic in the loops (a per-lane constant plus the block's base).
// (Tried first: 448-byte rows with the 16-byte chunk c of a part at position c ^ ((row >> 2) & 3) -- conflict-free as well.
//  In the FORWARD it passed every test; in the BACKWARD the same scheme gave run-to-run different results under some
//  instruction schedules -- B16_VPM 8 and 12, not 0 and 4; on 400-byte rows only with scheduling barriers around the
//  transposition image -- although the addresses replay correctly on the host, no LDS word is read before it is written
//  and every vmcnt / lgkmcnt wait of the listing checks out: docs/LAB_NOTES.md, "Round 6 notebook".  Not understood; the
//  XOR scheme is shipped in neither kernel.)
__device__ __forceinline__ int b16_prow(int row) { return (row & ~15) | ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int b16_off(int row, int part, int chunk) {   // byte offset of 16-byte chunk `chunk` (0..7) of a part
  return b16_prow(row) * B16_RSB + part * 128 + (chunk << 4);
}


// eight fp32 values (k-slots 0..7 of one MFMA) -> their three bf16 parts as MFMA fragments
__device__ __forceinline__ void split8(const float* x, bf16x8& p0, bf16x8& p1, bf16x8& p2) {
  u32x4 w0, w1, w2;
  unsigned a, b, c;
  split_pair<3>(x[0], x[1], a, b, c); w0.x = a; w1.x = b; w2.x = c;
  split_pair<3>(x[2], x[3], a, b, c); w0.y = a; w1.y = b; w2.y = c;
  split_pair<3>(x[4], x[5], a, b, c); w0.z = a; w1.z = b; w2.z = c;
  split_pair<3>(x[6], x[7], a, b, c); w0.w = a; w1.w = b; w2.w = c;
  p0 = __builtin_bit_cast(bf16x8, w0); p1 = __builtin_bit_cast(bf16x8, w1); p2 = __builtin_bit_cast(bf16x8, w2);
}

// acc += A B as six bf16 products, smallest cross terms first (part c of A with part e of B while c + e < 3)
__device__ __forceinline__ f32x16 mfma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc) {
#pragma unroll
  for (int o = 2; o >= 0; --o)
#pragma unroll
    for (int c = 0; c <= o; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c], b[o - c], acc, 0, 0, 0);
  return acc;
}

// A fragment (8 k-slots) of part c read by COLUMNS from a split image: element t of the result = image[row0 + rows(t)][col],
// rows(t) = t for t < 4, 8 + (t - 4) for t >= 4 (the accumulator rows (t & 3) + 8 (t >> 2) of a 32x32 tile), col = this lane's
// output row.  Two ds_read_b64_tr_b16: per group of 16 lanes a block of 4 rows x 16 columns arrives column-major -- lane
// 4 q + p of the group supplies the address of row q, columns 4 p .. 4 p + 3; lane i receives column i of the 4 rows.
// (row0 = 0 or 4 mod 16: logical row0 + 8 + q is two physical rows behind logical row0 + q)
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int row0, int colbase, int part, int lane) {
  const int q = (lane & 15) >> 2, p = lane & 3, g1 = (lane >> 4) & 1;
  const char* a0 = img + b16_prow(row0 + q) * B16_RSB + part * 128 + (colbase + 16 * g1 + 4 * p) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 2 * B16_RSB));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

struct MhaB16K {
  int B, H, Sq, Sk, ldq, ldk, ldv, ldo;
  const float* Q; const float* K; const float* V; const float* O; const float* dO;
  const uint8_t* mask; const float* biasT; const float* stats;
  float* dQ; float* dK; float* dV; float* dbiasT;
  DropCfg drop; float scale;
  const int* qoff; const int* koff;
};

#ifndef B16_VPM
#define B16_VPM 8    // vector instructions scheduled behind each MFMA of the overlapped phase (tuning)
#endif
#ifndef B16_DBG
#define B16_DBG 0   // timing experiments only (wrong results): bit mask of phases left out (1 S^T / dA^T products, 2 softmax backward,
#endif              // 4 dQ products + accumulation, 8 transposition + dK^T / dV^T products, 16 every step)
template <bool DB>
__global__ void __launch_bounds__(256, 1) mha_bwd_b16_kernel(const MhaB16K p) {
  __shared__ __attribute__((aligned(16))) char Qi[128 * B16_RSB];
  __shared__ __attribute__((aligned(16))) char Gi[128 * B16_RSB];
  __shared__ __attribute__((aligned(16))) float dQs[128 * 64];
  __shared__ float TrAll[4][32 * 33];
  __shared__ float sDel[128];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y, h = blockIdx.x;
  const int SqS = p.Sq, SkS = p.Sk;
  int Sq = p.Sq, Sk = p.Sk;
  size_t qrow0 = (size_t)b * p.Sq, krow0 = (size_t)b * p.Sk;
  if (p.qoff) { const int o = p.qoff[b]; Sq = p.qoff[b + 1] - o; qrow0 = (size_t)o; }
  if (p.koff) { const int o = p.koff[b]; Sk = p.koff[b + 1] - o; krow0 = (size_t)o; }
  if (Sq <= 0 || Sk <= 0) return;
  const size_t bh = (size_t)b * p.H + h;
  float* Tr = TrAll[w];
  const int hc = h * 64;   // first column of the head

  // ---- prologue: EVERY global load of the kernel is issued here, before the first conversion waits for one -- the wave's K / V
  //      rows (fragments), its transposed K block, the row statistics of all four query blocks, then the Q / dO / O tiles of
  //      the head.  (First written phase by phase: three dependent memory round trips, 9.8 us before the first tile.)
  const int key = 32 * w + l31;
  const bool kok = key < Sk;
  float4 kraw[8], vraw[8];
  {
    const size_t r = krow0 + (kok ? key : 0);
    const float* kr = p.K + r * p.ldk + hc + 8 * hh;
    const float* vr = p.V + r * p.ldv + hc + 8 * hh;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      kraw[2 * s4] = *reinterpret_cast<const float4*>(kr + 16 * s4); kraw[2 * s4 + 1] = *reinterpret_cast<const float4*>(kr + 16 * s4 + 4);
      vraw[2 * s4] = *reinterpret_cast<const float4*>(vr + 16 * s4); vraw[2 * s4 + 1] = *reinterpret_cast<const float4*>(vr + 16 * s4 + 4);
    }
  }
  float ktraw[2][2][8];
#pragma unroll
  for (int jc = 0; jc < 2; ++jc)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int kk = 32 * w + acc_row(8 * u + t, hh);
        ktraw[jc][u][t] = p.K[(krow0 + (kk < Sk ? kk : 0)) * p.ldk + hc + 32 * jc + l31];
      }
  float m_s[4], inv_s[4];      // row statistics of this lane's query in each of the four query blocks
#pragma unroll
  for (int qb = 0; qb < 4; ++qb) {
    const int qi = 32 * qb + l31;
    const size_t o = (bh * SqS + (qi < Sq ? qi : 0)) * 2;
    m_s[qb] = p.stats[o]; inv_s[qb] = p.stats[o + 1];
  }
  {
    float4 qv[8], gv[8], ov[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + 256 * i, row = f >> 4, c4 = f & 15;
      const bool ok = row < Sq;
      const size_t r = qrow0 + (ok ? row : 0);
      qv[i] = *reinterpret_cast<const float4*>(p.Q + r * p.ldq + hc + 4 * c4);
      gv[i] = *reinterpret_cast<const float4*>(p.dO + r * p.ldo + hc + 4 * c4);
      ov[i] = *reinterpret_cast<const float4*>(p.O + r * p.ldo + hc + 4 * c4);
      if (!ok) { qv[i] = make_float4(0.f, 0.f, 0.f, 0.f); gv[i] = qv[i]; }
    }
    for (int i = tid; i < 128 * 64 / 4; i += 256) reinterpret_cast<float4*>(dQs)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + 256 * i, row = f >> 4, c4 = f & 15;
      unsigned a0, a1, a2, b0, b1, b2;
      split_pair<3>(qv[i].x, qv[i].y, a0, a1, a2);
      split_pair<3>(qv[i].z, qv[i].w, b0, b1, b2);
      char* d = Qi + b16_prow(row) * B16_RSB + c4 * 8;
      *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
      *reinterpret_cast<uint2*>(d + 128) = make_uint2(a1, b1);
      *reinterpret_cast<uint2*>(d + 256) = make_uint2(a2, b2);
      split_pair<3>(gv[i].x, gv[i].y, a0, a1, a2);
      split_pair<3>(gv[i].z, gv[i].w, b0, b1, b2);
      d = Gi + b16_prow(row) * B16_RSB + c4 * 8;
      *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
      *reinterpret_cast<uint2*>(d + 128) = make_uint2(a1, b1);
      *reinterpret_cast<uint2*>(d + 256) = make_uint2(a2, b2);
      float part = (gv[i].x * ov[i].x + gv[i].y * ov[i].y) + (gv[i].z * ov[i].z + gv[i].w * ov[i].w);
      part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64); part += __shfl_xor(part, 8, 64);
      if (c4 == 0) sDel[row] = part;    // (rows behind Sq: dO was zeroed)
    }
  }
  // the wave's key block as MFMA fragments: A fragments of K and V (row = key l31, 8 consecutive head-dim indices per k-step
  // and lane half) and the transposed fragments of K (B operand of dQ += dS K: column = head-dim index l31 + 32 jc, k-slots =
  // the accumulator rows of a lane)
  bf16x8 kA[4][3], vA[4][3];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    float4 x0 = kraw[2 * s4], x1 = kraw[2 * s4 + 1], y0 = vraw[2 * s4], y1 = vraw[2 * s4 + 1];
    if (!kok) { x0 = make_float4(0.f, 0.f, 0.f, 0.f); x1 = x0; y0 = x0; y1 = x0; }
    const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    const float ys[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
    split8(xs, kA[s4][0], kA[s4][1], kA[s4][2]);
    split8(ys, vA[s4][0], vA[s4][1], vA[s4][2]);
  }
  bf16x8 kT[2][2][3];   // [jc][u][part]
#pragma unroll
  for (int jc = 0; jc < 2; ++jc)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float xs[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) xs[t] = (32 * w + acc_row(8 * u + t, hh) < Sk) ? ktraw[jc][u][t] : 0.f;
      split8(xs, kT[jc][u][0], kT[jc][u][1], kT[jc][u][2]);
    }
  const bool masked_any = p.mask != nullptr;
  float mk[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kk = 32 * w + acc_row(r, hh);
    mk[r] = (masked_any && kk < Sk) ? (float)p.mask[(size_t)b * SkS + kk] : 0.f;
  }
  f32x16 dkT[2], dvT[2];
#pragma unroll
  for (int jc = 0; jc < 2; ++jc)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkT[jc][r] = 0.f; dvT[jc][r] = 0.f; }

  const size_t bho = bh * (size_t)SkS * SqS;
  const unsigned plane = (unsigned)SkS * (unsigned)SqS * 4u;
  const __amdgpu_buffer_rsrc_t bias_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.biasT ? p.biasT + bho : p.Q), 0, p.biasT ? plane : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t dbias_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(DB ? p.dbiasT + bho : p.dQ), 0, DB ? plane : 0u, 0x00020000);
  __syncthreads();

  // A tile = (this wave's key block) x (one query block); three phases.  No branch anywhere in a tile (a query block behind
  // the sequence or a wave without keys multiplies zeros: its contributions are exact zeros), so the whole step is ONE
  // scheduling region and the products of the NEXT tile's first phase are issued between the vector instructions of this
  // tile's softmax backward (one wave per SIMD: nothing else hides either).
  // phase 1: S^T = K Q^T, dA^T = V dO^T -- A = the wave's K / V fragments, B = row reads of the images (this lane's query)
  auto P1 = [&](const int qb, f32x16& acc, f32x16& dacc, float (&bias)[16]) __attribute__((always_inline)) {
    const int qi = 32 * qb + l31;
    const bool qok = qi < Sq;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = 32 * w + acc_row(r, hh);
      const bool okk = kk < Sk && qok;
      bias[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bias_rs, okk ? (unsigned)(kk * SqS + qi) * 4u : ~0u, 0, 0));
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; dacc[r] = 0.f; }
#if !(B16_DBG & 1)
    const char* qrow = Qi + b16_prow(qi) * B16_RSB + 16 * hh;
    const char* grow = Gi + b16_prow(qi) * B16_RSB + 16 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 qf[3], gf[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        qf[c] = *reinterpret_cast<const bf16x8*>(qrow + c * 128 + 32 * ks);
        gf[c] = *reinterpret_cast<const bf16x8*>(grow + c * 128 + 32 * ks);
      }
      acc = mfma6(kA[ks], qf, acc);
      dacc = mfma6(vA[ks], gf, dacc);
    }
#endif
  };
  // phase 2: softmax backward in registers: acc <- dZ^T / sqrt(d_h), dacc <- A^T = (P o D)^T    [key][query]
  auto P2 = [&](const int qb, const float m, const float inv, f32x16& acc, f32x16& dacc, const float (&bias)[16]) __attribute__((always_inline)) {
#if !(B16_DBG & 2)
    const int qi = 32 * qb + l31;
    const bool qok = qi < Sq;
    const float del = sDel[qi];
    const uint32_t dpre = drop_pre(p.drop, (uint32_t)((bh * SqS + qi) * SkS + 32 * w + 4 * hh));
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = 32 * w + acc_row(r, hh);
      const bool ok = kk < Sk && qok;
      const bool masked = mk[r] != 0.f;
      float v = acc[r] * p.scale + bias[r];
      v = masked ? -1e9f : v;
      const float pr = __expf(v - m) * inv;
      const float dm = drop_mult_pre(p.drop, dpre + (uint32_t)((r & 3) + 8 * (r >> 2)) * DROP_G);   // (no dropout: multiplier 1)
      const float dz = (ok && !masked) ? pr * (dacc[r] * dm - del) : 0.f;
      if (DB) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dz), dbias_rs, ok ? (unsigned)(kk * SqS + qi) * 4u : ~0u, 0, 0);
      acc[r] = dz * p.scale;
      dacc[r] = ok ? pr * dm : 0.f;
    }
#endif
  };
  // phases 3 + 4: dQ += dS K (the tile as A operand); the two tiles through the wave's LDS image; dK^T / dV^T
  auto P34 = [&](const int qb, f32x16& acc, f32x16& dacc) __attribute__((always_inline)) {
    const int q0 = 32 * qb;
#if !(B16_DBG & 4)
    {
      float xs[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) xs[r] = acc[r];
      bf16x8 aS[2][3];
      split8(xs, aS[0][0], aS[0][1], aS[0][2]);
      split8(xs + 8, aS[1][0], aS[1][1], aS[1][2]);
#pragma unroll
      for (int jc = 0; jc < 2; ++jc) {
        f32x16 dq;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[r] = 0.f;
        dq = mfma6(aS[0], kT[jc][0], dq);
        dq = mfma6(aS[1], kT[jc][1], dq);
        // (plain read-modify-write: the waves of a step own different rows.  ds_add_f32 instead -- one LDS instruction per
        //  element -- took the kernel from 38 to 74 us: LDS float atomics serialise, as round 2 found for the fp32 kernel)
#pragma unroll
        for (int r = 0; r < 16; ++r) dQs[(q0 + acc_row(r, hh)) * 64 + 32 * jc + l31] += dq[r];
      }
    }
#endif
#if B16_DBG & 8
    dkT[0][0] += acc[3]; dvT[1][2] += dacc[5];
#else
    float tz[16], tp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) Tr[acc_row(r, hh) * 33 + l31] = acc[r];
#pragma unroll
    for (int r = 0; r < 16; ++r) tz[r] = Tr[l31 * 33 + acc_row(r, hh)];
#pragma unroll
    for (int r = 0; r < 16; ++r) Tr[acc_row(r, hh) * 33 + l31] = dacc[r];
#pragma unroll
    for (int r = 0; r < 16; ++r) tp[r] = Tr[l31 * 33 + acc_row(r, hh)];
    bf16x8 zB[2][3], pB[2][3];
    split8(tz, zB[0][0], zB[0][1], zB[0][2]);
    split8(tz + 8, zB[1][0], zB[1][1], zB[1][2]);
    split8(tp, pB[0][0], pB[0][1], pB[0][2]);
    split8(tp + 8, pB[1][0], pB[1][1], pB[1][2]);
#pragma unroll
    for (int jc = 0; jc < 2; ++jc)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bf16x8 qT[3], gT[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          qT[c] = tr_frag(Qi, q0 + 16 * u + 4 * hh, 32 * jc, c, lane);
          gT[c] = tr_frag(Gi, q0 + 16 * u + 4 * hh, 32 * jc, c, lane);
        }
        dkT[jc] = mfma6(qT, zB[u], dkT[jc]);
        dvT[jc] = mfma6(gT, pB[u], dvT[jc]);
      }
#endif
  };

  f32x16 tA[2], tD[2];
  float tb[2][16];
  constexpr int NSTEP = (B16_DBG & 16) ? 0 : 4;
  if (NSTEP > 0) P1(w & 3, tA[0], tD[0], tb[0]);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int cur = s & 1, nxt = cur ^ 1;
    const int qb = (w + s) & 3;
    if (s + 1 < NSTEP) P1((w + s + 1) & 3, tA[nxt], tD[nxt], tb[nxt]);
    const float m = qb == 0 ? m_s[0] : (qb == 1 ? m_s[1] : (qb == 2 ? m_s[2] : m_s[3]));
    const float inv = qb == 0 ? inv_s[0] : (qb == 1 ? inv_s[1] : (qb == 2 ? inv_s[2] : inv_s[3]));
    P2(qb, m, inv, tA[cur], tD[cur], tb[cur]);
    if (s + 1 < NSTEP) {   // one MFMA of the next tile, then a slice of this tile's vector work, 48 times
#pragma unroll
      for (int i = 0; i < 48; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, B16_VPM, 0);
      }
    }
    P34(qb, tA[cur], tD[cur]);
    __syncthreads();
  }

  // ---- epilogue: dK / dV of this wave's keys from registers (lane = key, registers = 4 runs of 4 consecutive head-dim
  //      indices: 16-byte stores); dQ from the LDS accumulator ----
  if (kok) {
    float* dkr = p.dK + (krow0 + key) * p.ldk + hc;
    float* dvr = p.dV + (krow0 + key) * p.ldv + hc;
#pragma unroll
    for (int jc = 0; jc < 2; ++jc)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = 32 * jc + 8 * g + 4 * hh;
        *reinterpret_cast<float4*>(dkr + j) = make_float4(dkT[jc][4 * g], dkT[jc][4 * g + 1], dkT[jc][4 * g + 2], dkT[jc][4 * g + 3]);
        *reinterpret_cast<float4*>(dvr + j) = make_float4(dvT[jc][4 * g], dvT[jc][4 * g + 1], dvT[jc][4 * g + 2], dvT[jc][4 * g + 3]);
      }
  }
  for (int i = tid; i < Sq * 16; i += 256) {
    const int row = i >> 4, c4 = i & 15;
    *reinterpret_cast<float4*>(p.dQ + (qrow0 + row) * p.ldq + hc + 4 * c4) = reinterpret_cast<const float4*>(dQs)[row * 16 + c4];
  }
}

// ------------------------------------------------------------------------------------------
// Forward on the same scheme (d_h = 64, 65..128 keys, <= 128 queries): a wave owns 32 QUERIES (its Q rows are pre-split B
// fragments in registers); K and V of the head sit in LDS as split images.  S^T = K Q^T reads K by rows (ds_read_b128);
// the softmax runs in registers with the query on the lane (attention.hip's forward, unchanged arithmetic); the probability
// tile is split in registers and is the A operand of O = A V, whose B operand is V read by COLUMNS (transpose reads).
// 192 bf16 MFMAs of 32 cycles per wave where the fp32 kernel issues 256 fp32 MFMAs of 64.
// ------------------------------------------------------------------------------------------
struct MhaF16K {
  int B, H, Sq, Sk, ldq, ldk, ldv, ldo;
  const float* Q; const float* K; const float* V; const uint8_t* mask; const float* biasT;
  float* O; float* stats;
  DropCfg drop; float scale;
  const int* qoff; const int* koff;
};

// ONEBUF: ONE image buffer -- K first, V stored over it behind the score products (its rows wait in registers) -- 51 KB of LDS
// and <= 256 registers: TWO workgroups per CU.  For launches of more than one round of workgroups (H = 8 at B = 64) and for the
// architecture step's two-cores-in-one-launch form, where the second workgroup of a CU hides the first one's latencies; the
// two-buffer form (one barrier less, the relation bias fetched up front) stays for launches that are one round anyway.
template <bool ONEBUF>
__device__ __forceinline__ void mha_fwd_b16_body(const MhaF16K& p, const int h, char* Ki, char* Vi) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l31 = lane & 31, hh = lane >> 5;
  const int b = blockIdx.y;
  const int SqS = p.Sq, SkS = p.Sk;
  int Sq = p.Sq, Sk = p.Sk;
  size_t qrow0 = (size_t)b * p.Sq, krow0 = (size_t)b * p.Sk;
  if (p.qoff) { const int o = p.qoff[b]; Sq = p.qoff[b + 1] - o; qrow0 = (size_t)o; }
  if (p.koff) { const int o = p.koff[b]; Sk = p.koff[b + 1] - o; krow0 = (size_t)o; }
  if (Sq <= 0 || Sk <= 0) return;
  const size_t bh = (size_t)b * p.H + h;
  const int hc = h * 64;
  const int qi = 32 * w + l31;
  const bool qok = qi < Sq;

  // ---- every global load first: this lane's query row (fragments), the K / V tiles of the head, the bias ----
  float4 qraw[8];
  {
    const float* qr = p.Q + (qrow0 + (qok ? qi : 0)) * p.ldq + hc + 8 * hh;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) { qraw[2 * s4] = *reinterpret_cast<const float4*>(qr + 16 * s4); qraw[2 * s4 + 1] = *reinterpret_cast<const float4*>(qr + 16 * s4 + 4); }
  }
  float4 kv[8], vv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = tid + 256 * i, row = f >> 4, c4 = f & 15;
    const bool ok = row < Sk;
    const size_t r = krow0 + (ok ? row : 0);
    kv[i] = *reinterpret_cast<const float4*>(p.K + r * p.ldk + hc + 4 * c4);
    vv[i] = *reinterpret_cast<const float4*>(p.V + r * p.ldv + hc + 4 * c4);
    if (!ok) { kv[i] = make_float4(0.f, 0.f, 0.f, 0.f); vv[i] = kv[i]; }
  }
  // the relation bias of this lane's query against all 128 keys, fetched with everything else (64 values per lane; the
  // kernel has the whole register file: one wave per SIMD)
  const bool has_bias = p.biasT != nullptr;
  const int qic = qok ? qi : Sq - 1;
  float biasv[ONEBUF ? 1 : 4][16];
  if (!ONEBUF && has_bias) {
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = min(32 * kc + acc_row(r, hh), Sk - 1);
        biasv[ONEBUF ? 0 : kc][r] = p.biasT[(bh * SkS + key) * SqS + qic];
      }
  }
  // key mask / key range as bit masks (one ballot per 64 keys), pre-shifted per lane half: element (kc, r) tests bit
  // 32 (kc & 1) + (r & 3) + 8 (r >> 2) of word kc >> 1 (attention.hip's forward)
  unsigned long long mbits[2], vbits[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int key = 64 * j + lane;
    const bool inr = key < Sk;
    const bool mk = p.mask && inr && p.mask[(size_t)b * SkS + (inr ? key : 0)];
    mbits[j] = __ballot(mk) >> (4 * hh);
    vbits[j] = __ballot(inr) >> (4 * hh);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = tid + 256 * i, row = f >> 4, c4 = f & 15;
    unsigned a0, a1, a2, b0, b1, b2;
    split_pair<3>(kv[i].x, kv[i].y, a0, a1, a2);
    split_pair<3>(kv[i].z, kv[i].w, b0, b1, b2);
    char* d = Ki + b16_off(row, 0, c4 >> 1) + (c4 & 1) * 8;
    *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
    *reinterpret_cast<uint2*>(d + 128) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(d + 256) = make_uint2(a2, b2);
    if (!ONEBUF) {
      split_pair<3>(vv[i].x, vv[i].y, a0, a1, a2);
      split_pair<3>(vv[i].z, vv[i].w, b0, b1, b2);
      d = Vi + b16_off(row, 0, c4 >> 1) + (c4 & 1) * 8;
      *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
      *reinterpret_cast<uint2*>(d + 128) = make_uint2(a1, b1);
      *reinterpret_cast<uint2*>(d + 256) = make_uint2(a2, b2);
    }
  }
  bf16x8 qB[4][3];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    float4 x0 = qraw[2 * s4], x1 = qraw[2 * s4 + 1];
    if (!qok) { x0 = make_float4(0.f, 0.f, 0.f, 0.f); x1 = x0; }
    const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    split8(xs, qB[s4][0], qB[s4][1], qB[s4][2]);
  }
  __syncthreads();

  // ---- S^T = K Q^T: A = row reads of the K image (key block kc), B = this lane's query fragments ----
  f32x16 acc[4];
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[kc][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 kf[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) kf[c] = *reinterpret_cast<const bf16x8*>(Ki + b16_off(32 * kc + l31, c, 2 * ks + hh));
      acc[kc] = mfma6(kf, qB[ks], acc[kc]);
    }
  }
  if (ONEBUF) {   // every wave has its scores: the V image takes the K image's place
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = tid + 256 * i, row = f >> 4, c4 = f & 15;
      unsigned a0, a1, a2, b0, b1, b2;
      split_pair<3>(vv[i].x, vv[i].y, a0, a1, a2);
      split_pair<3>(vv[i].z, vv[i].w, b0, b1, b2);
      char* d = Vi + b16_off(row, 0, c4 >> 1) + (c4 & 1) * 8;
      *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
      *reinterpret_cast<uint2*>(d + 128) = make_uint2(a1, b1);
      *reinterpret_cast<uint2*>(d + 256) = make_uint2(a2, b2);
    }
  }
  // ---- softmax over the keys of this lane's query (registers + the other half-wave) ----
  float m = -INFINITY;
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    if (ONEBUF && has_bias) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = min(32 * kc + acc_row(r, hh), Sk - 1);
        biasv[0][r] = p.biasT[(bh * SkS + key) * SqS + qic];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      constexpr unsigned long long one = 1ull;
      const unsigned long long bit = one << (32 * (kc & 1) + (r & 3) + 8 * (r >> 2));
      float v = acc[kc][r] * p.scale;
      if (has_bias) v += biasv[ONEBUF ? 0 : kc][r];
      v = (mbits[kc >> 1] & bit) ? -1e9f : v;
      v = (vbits[kc >> 1] & bit) ? v : -INFINITY;
      acc[kc][r] = v;
      m = fmaxf(m, v);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int kc = 0; kc < 4; ++kc)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(acc[kc][r] - m);
      acc[kc][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  if (qok && hh == 0) {
    p.stats[(bh * SqS + qi) * 2] = m;
    p.stats[(bh * SqS + qi) * 2 + 1] = inv;
  }
  const uint32_t dpre = drop_pre(p.drop, (uint32_t)((bh * SqS + qi) * SkS + 4 * hh));
  if (ONEBUF) __syncthreads();   // the V image is complete
  // ---- O = A V: the probability tile (split in registers) as A operand, V by columns ----
  f32x16 o[2];
#pragma unroll
  for (int jc = 0; jc < 2; ++jc)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[jc][r] = 0.f;
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    float xs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a = acc[kc][r] * inv;
      if (p.drop.thresh) a *= drop_mult_pre(p.drop, dpre + (uint32_t)(32 * kc + (r & 3) + 8 * (r >> 2)) * DROP_G);
      xs[r] = a;
    }
    bf16x8 aP[2][3];
    split8(xs, aP[0][0], aP[0][1], aP[0][2]);
    split8(xs + 8, aP[1][0], aP[1][1], aP[1][2]);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int jc = 0; jc < 2; ++jc) {
        bf16x8 vT[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) vT[c] = tr_frag(Vi, 32 * kc + 16 * u + 4 * hh, 32 * jc, c, lane);
        o[jc] = mfma6(aP[u], vT, o[jc]);
      }
  }
  // the output tile leaves as whole 256-byte rows: through the K image's LDS (free once every wave has its scores: barrier),
  // [query][64 + 4 pad] fp32 per wave, read back 16 bytes per lane
  __syncthreads();
  float* Ot = reinterpret_cast<float*>(Ki) + w * (32 * 68);
#pragma unroll
  for (int jc = 0; jc < 2; ++jc)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ot[acc_row(r, hh) * 68 + 32 * jc + l31] = o[jc][r];
  // (the wave reads back what it wrote: same-wave LDS accesses stay in order)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = lane + 64 * i, row = f >> 4, c4 = f & 15;
    const int q = 32 * w + row;
    const float4 v4 = *reinterpret_cast<const float4*>(Ot + row * 68 + 4 * c4);
    if (q < Sq) *reinterpret_cast<float4*>(p.O + (qrow0 + q) * p.ldo + hc + 4 * c4) = v4;
  }
}

__global__ void __launch_bounds__(256, 1) mha_fwd_b16_kernel(const MhaF16K p) {
  __shared__ __attribute__((aligned(16))) char Ki[128 * B16_RSB];
  __shared__ __attribute__((aligned(16))) char Vi[128 * B16_RSB];
  mha_fwd_b16_body<false>(p, blockIdx.x, Ki, Vi);
}
// one image buffer, two workgroups per CU; blockIdx.x = head of problem 0, then head of problem 1 (nh0 = p0.H; p1 unused when
// the grid has only p0.H columns)
__global__ void __launch_bounds__(256, 2) mha_fwd_b16_two_kernel(const MhaF16K p0, const MhaF16K p1, const int nh0) {
  __shared__ __attribute__((aligned(16))) char KVi[128 * B16_RSB];
  if ((int)blockIdx.x < nh0) mha_fwd_b16_body<true>(p0, blockIdx.x, KVi, KVi);
  else mha_fwd_b16_body<true>(p1, (int)blockIdx.x - nh0, KVi, KVi);
}

static bool fwd16_fill(MhaF16K& k, int B, int H, int Sq, int Sk, int ldq, int ldk, int ldv, int ldo, const float* Q, const float* K, const float* V,
                       const uint8_t* mask, const float* biasT, float* O, float* stats, DropCfg drop, float scale, const int* qoff, const int* koff) {
  if (Sq > 128 || Sk > 128 || Sk <= 64) return false;
  if ((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) != 0 || (ldq | ldk | ldv | ldo) % 4 != 0) return false;
  memset(&k, 0, sizeof(k));
  k.B = B; k.H = H; k.Sq = Sq; k.Sk = Sk; k.ldq = ldq; k.ldk = ldk; k.ldv = ldv; k.ldo = ldo;
  k.Q = Q; k.K = K; k.V = V; k.mask = mask; k.biasT = biasT; k.O = O; k.stats = stats; k.drop = drop; k.scale = scale;
  k.qoff = qoff; k.koff = koff;
  return true;
}
static bool fwd16_on() {
  static const int on = [] { const char* e = getenv("MMNAS_MHA_FWD_B16"); return (e && e[0] == '0') ? 0 : 1; }();
  return on != 0;
}

bool mha_fwd_b16_launch(int B, int H, int Sq, int Sk, int ldq, int ldk, int ldv, int ldo, const float* Q, const float* K, const float* V,
                        const uint8_t* mask, const float* biasT, float* O, float* stats, DropCfg drop, float scale, const int* qoff,
                        const int* koff, hipStream_t st) {
  MhaF16K k;
  if (!fwd16_on() || !fwd16_fill(k, B, H, Sq, Sk, ldq, ldk, ldv, ldo, Q, K, V, mask, biasT, O, stats, drop, scale, qoff, koff)) return false;
  static const int two_from = [] { const char* e = getenv("MMNAS_MHA_FWD_B16_TWO"); return e && e[0] ? atoi(e) : 320; }();
  if ((long)B * H > two_from) MMNAS_LAUNCH(mha_fwd_b16_two_kernel, dim3(H, B), dim3(256), 0, st, k, k, H);   // more than one round: two per CU
  else MMNAS_LAUNCH(mha_fwd_b16_kernel, dim3(H, B), dim3(256), 0, st, k);
  return true;
}

// attention.hip: two cores of one geometry in one launch (the mixed chain); false = not this kernel's shapes
bool mha_fwd_b16_pair(int B, int H, int Sq, int Sk, int ld, const float* const* Q, const float* const* K, const float* const* V,
                      const uint8_t* const* mask, const float* const* biasT, float* const* O, float* const* stats, const DropCfg* drop,
                      float scale, const int* qoff, const int* koff, hipStream_t st) {
  MhaF16K k0, k1;
  if (!fwd16_on()) return false;
  if (!fwd16_fill(k0, B, H, Sq, Sk, ld, ld, ld, ld, Q[0], K[0], V[0], mask[0], biasT[0], O[0], stats[0], drop[0], scale, qoff, koff)) return false;
  if (!fwd16_fill(k1, B, H, Sq, Sk, ld, ld, ld, ld, Q[1], K[1], V[1], mask[1], biasT[1], O[1], stats[1], drop[1], scale, qoff, koff)) return false;
  MMNAS_LAUNCH(mha_fwd_b16_two_kernel, dim3(2 * H, B), dim3(256), 0, st, k0, k1, H);
  return true;
}

// attention.hip hands over the (validated) problem; returns false when the shape is outside this kernel's range
bool mha_bwd_b16_launch(int B, int H, int Sq, int Sk, int ldq, int ldk, int ldv, int ldo, const float* Q, const float* K, const float* V,
                        const float* O, const float* dO, const uint8_t* mask, const float* biasT, const float* stats, float* dQ, float* dK,
                        float* dV, float* dbiasT, DropCfg drop, float scale, const int* qoff, const int* koff, hipStream_t st) {
  static const int on = [] { const char* e = getenv("MMNAS_MHA_BWD_B16"); return (e && e[0] == '0') ? 0 : 1; }();
  if (!on || Sq > 128 || Sk > 128 || Sk <= 64) return false;   // (few keys: waves without keys would idle -- the fp32 kernel's small instantiations stay)
  if ((((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)dO | (uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) & 15) != 0) return false;
  if ((ldq | ldk | ldv | ldo) % 4 != 0) return false;
  MhaB16K k;
  memset(&k, 0, sizeof(k));
  k.B = B; k.H = H; k.Sq = Sq; k.Sk = Sk; k.ldq = ldq; k.ldk = ldk; k.ldv = ldv; k.ldo = ldo;
  k.Q = Q; k.K = K; k.V = V; k.O = O; k.dO = dO; k.mask = mask; k.biasT = biasT; k.stats = stats;
  k.dQ = dQ; k.dK = dK; k.dV = dV; k.dbiasT = dbiasT; k.drop = drop; k.scale = scale; k.qoff = qoff; k.koff = koff;
  if (dbiasT) MMNAS_LAUNCH((mha_bwd_b16_kernel<true>), dim3(H, B), dim3(256), 0, st, k);
  else MMNAS_LAUNCH((mha_bwd_b16_kernel<false>), dim3(H, B), dim3(256), 0, st, k);
  return true;
}

}  // namespace mmnas

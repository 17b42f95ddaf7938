// Split-operand helpers shared by the matrix-product kernels (gemm.hip, gemmln.hip): an fp32 operand as three bf16 parts
// (x = h + m + l exactly), their LDS image and the buffer loads that feed it.  Moved out of gemm.hip in round 6, unchanged.
#pragma once
#include "common.h"

#ifndef MMNAS_SPLIT_NOPK
#define MMNAS_SPLIT_NOPK 1
#endif

namespace mmnas {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff = 0) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
  float4 f;
  f.x = __uint_as_float(v.x); f.y = __uint_as_float(v.y); f.z = __uint_as_float(v.z); f.w = __uint_as_float(v.w);
  return f;
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// (x0, x1) -> NS packed bf16 pairs: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); the subtractions are exact
// (x0, x1) -> NS packed bf16 pairs written to dst[c * 16], c = 0..NS-1: h = bf16(x), m = bf16(x - h),
// l = bf16(x - h - m); the subtractions are exact
template <int NS>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& w0, unsigned& w1, unsigned& w2) {
#ifdef MMNAS_DBG_NOCONV   // timing experiment only (wrong results): what the conversion VALU costs
  w0 = __float_as_uint(x0); w1 = __float_as_uint(x1); w2 = w0 ^ w1;
  return;
#endif
#if MMNAS_SPLIT_NOPK
  // the two subtractions of a pair as two v_sub_f32: left to itself the compiler packs them into one v_pk_add_f32, which
  // costs more issue time than the two it replaces (MI355X_MICROARCH.md, per-instruction constants)
  f32x2 r = {x0, x1};
  const bf16x2 h = __builtin_convertvector(r, bf16x2);
  w0 = __builtin_bit_cast(unsigned, h);
  if (NS == 1) return;   // single-pass bf16: the rounded operand is all there is
  float r0, r1;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(__uint_as_float(w0 << 16)));
  asm("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(__uint_as_float(w0 & 0xffff0000u)));
  r = f32x2{r0, r1};
  const bf16x2 m = __builtin_convertvector(r, bf16x2);
  w1 = __builtin_bit_cast(unsigned, m);
  if (NS > 2) {
    float q0, q1;
    asm("v_sub_f32 %0, %1, %2" : "=v"(q0) : "v"(r0), "v"(__uint_as_float(w1 << 16)));
    asm("v_sub_f32 %0, %1, %2" : "=v"(q1) : "v"(r1), "v"(__uint_as_float(w1 & 0xffff0000u)));
    r = f32x2{q0, q1};
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  }
#else
  f32x2 r = {x0, x1};
  const bf16x2 h = __builtin_convertvector(r, bf16x2);
  w0 = __builtin_bit_cast(unsigned, h);
  if (NS == 1) return;
  r -= __builtin_convertvector(h, f32x2);
  const bf16x2 m = __builtin_convertvector(r, bf16x2);
  w1 = __builtin_bit_cast(unsigned, m);
  if (NS > 2) {
    r -= __builtin_convertvector(m, f32x2);
    w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  }
#endif
}

// LDS image of the split operands (checked lane group by lane group against the bank rules of MI355X_MICROARCH.md by
// tools/lds_bank_model.py, confirmed by SQ_LDS_BANK_CONFLICT: every access class of a K-tile is conflict-free).
// Row r of a tile: NS runs of 32 bf16 + 16 B pad (RSW = 16 NS + 4 words: an odd number of 16-B groups).  Word w
// (0..15: a pair of K indices) of a part sits at word w ^ 8 p(r), p = parity of row bits 2-4:
//   * fragment reads (ds_read_b128, 64 banks, four 16-lane groups {0-3,12-15,20-27}, ...: 16 rows, all residues mod 16):
//     13 r + (g ^ 2 p(r)) mod 16 is a bijection on every such group;
//   * K-contiguous stores (ds_write_b64, 32 banks, 16 consecutive lanes): the lanes of a group hold rows {r, r + 4} x 8
//     chunks (kc_row) -- 208-B rows put r and r + 4 sixteen banks apart;
//   * transposing stores (ds_write_b32, 32 banks, 32 consecutive lanes): a group holds 8 k-pairs x 4 row quads (t_map) --
//     the k-pairs fill an aligned run of 8 banks, the quad's bit 0 moves it by 16 banks (4 x 52 = 16 mod 32), bit 1 by 8
//     through p.
// (Round 2 XORed the 16-B group with row bits 4-5 and kept lane-linear store maps: exactly 2-way on all three classes.)
__device__ __forceinline__ int swz(int w, int row) { return w ^ ((((row >> 2) ^ (row >> 3) ^ (row >> 4)) & 1) << 3); }
// load / store f (0 .. rows * 8) of a K-contiguous tile -> its row; the chunk is f & 7
__device__ __forceinline__ int kc_row(int f) { const int r = f >> 3; return (r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3); }
// One K-tile of one operand, registers -> LDS in split form: row r of the tile is NS runs of 32 bf16 (part c at
// word c*16), K index j at position j of each run.  KC: a thread holds 4 consecutive K of one row per load;
// otherwise loads 2j / 2j+1 hold rows k = 2kp / 2kp+1 of 4 consecutive tile rows.
template <int BR, bool KC, int NS>
__device__ __forceinline__ void split_store_kc(unsigned* dst, const float4 v, int f) {
  constexpr int RSW = NS * 16 + 4;
  const int row = kc_row(f), kq = f & 7;
  unsigned a0, a1 = 0, a2 = 0, b0, b1 = 0, b2 = 0;
  split_pair<NS>(v.x, v.y, a0, a1, a2);
  split_pair<NS>(v.z, v.w, b0, b1, b2);
  unsigned* d = dst + row * RSW + swz(2 * kq, row);
  *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
  if (NS > 1) *reinterpret_cast<uint2*>(d + 16) = make_uint2(a1, b1);
  if (NS > 2) *reinterpret_cast<uint2*>(d + 32) = make_uint2(a2, b2);
}
}  // namespace mmnas

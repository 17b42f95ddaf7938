// Does one SIMD of an MI355X CU run a wave's dependent v_mfma_f32_32x32x16_bf16 chain beside another wave's vector
// instructions of the kind the split-operand GEMM's conversion uses (v_cvt_pk_bf16_f32, v_and, v_lshl, v_sub_f32)?
// Tuning probe: hipcc --offload-arch=gfx950 -O3 -o overlap_probe overlap_probe.hip ; ./overlap_probe
//   mode 0: 4 waves per workgroup (one per SIMD), all run the MFMA chain
//   mode 1: 4 waves, all run the conversion chain
//   mode 2: 8 waves: waves 0-3 the MFMA chain, waves 4-7 the conversion chain (each SIMD hosts one of each)
//   mode 3: 8 waves, all MFMA;  mode 4: 8 waves, all conversion
//   mode 5: 8 waves: 0-3 MFMA, 4-7 LDS traffic (ds_read_b128 + ds_write_b64 of the GEMM's per-tile volume)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NACC>
__device__ __forceinline__ void mfma_chain_n(int iters, float* out) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k % NACC], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  if (s == 123.456f) out[threadIdx.x] = s;
}
__device__ __forceinline__ void mfma_chain(int iters, float* out) { mfma_chain_n<1>(iters, out); }

// the same chain with the wave stepping back from the issue port between MFMAs: NOPS cycles of s_nop after each one
template <int NOPS, int SLEEP>
__device__ __forceinline__ void mfma_chain_yield(int iters, float* out) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      if (SLEEP) __builtin_amdgcn_s_sleep(1);
      else {
        if (NOPS >= 16) asm volatile("s_nop 15");
        if (NOPS % 16) asm volatile("s_nop %0" ::"n"((NOPS % 16) - 1));   // (one s_nop unit = 4 clocks)
      }
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// one wave doing both: per MFMA one slice of the conversion (8 pairs over 12 MFMAs), NACC accumulators
template <int NACC>
__device__ __forceinline__ void both_chain(int iters, float* out) {
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f); }
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = 1.0f + 0.001f * (threadIdx.x + i);
  unsigned sink = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k % NACC], 0, 0, 0);
      if (k % 3 != 2) {
        const int pr = (k / 3) * 2 + (k % 3);
        const float x0 = x[2 * pr], x1 = x[2 * pr + 1];
        f32x2 r = {x0, x1};
        const bf16x2 h = __builtin_convertvector(r, bf16x2);
        const unsigned w0 = __builtin_bit_cast(unsigned, h);
        float r0, r1;
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(__uint_as_float(w0 << 16)));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(__uint_as_float(w0 & 0xffff0000u)));
        r = f32x2{r0, r1};
        const bf16x2 m = __builtin_convertvector(r, bf16x2);
        const unsigned w1 = __builtin_bit_cast(unsigned, m);
        float q0, q1;
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(q0) : "v"(r0), "v"(__uint_as_float(w1 << 16)));
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(q1) : "v"(r1), "v"(__uint_as_float(w1 & 0xffff0000u)));
        r = f32x2{q0, q1};
        const unsigned w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
        sink ^= w0 ^ w1 ^ w2;
        x[2 * pr] = x0 + 1e-7f; x[2 * pr + 1] = x1 + 1e-7f;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n)
    for (int r = 0; r < 16; ++r) s += acc[n][r];
  if (s == 123.456f || sink == 0x12345678u) out[threadIdx.x] = s;
}

__device__ __forceinline__ void conv_chain(int iters, float* out) {
  // per "K-tile": 4 float4 = 8 pairs, each pair -> 3 packed bf16 words (the GEMM's split_pair with scalar subtractions)
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = 1.0f + 0.001f * (threadIdx.x + i);
  unsigned sink = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int pr = 0; pr < 8; ++pr) {
      const float x0 = x[2 * pr], x1 = x[2 * pr + 1];
      f32x2 r = {x0, x1};
      const bf16x2 h = __builtin_convertvector(r, bf16x2);
      const unsigned w0 = __builtin_bit_cast(unsigned, h);
      float r0, r1;
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(__uint_as_float(w0 << 16)));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(__uint_as_float(w0 & 0xffff0000u)));
      r = f32x2{r0, r1};
      const bf16x2 m = __builtin_convertvector(r, bf16x2);
      const unsigned w1 = __builtin_bit_cast(unsigned, m);
      float q0, q1;
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(q0) : "v"(r0), "v"(__uint_as_float(w1 << 16)));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(q1) : "v"(r1), "v"(__uint_as_float(w1 & 0xffff0000u)));
      r = f32x2{q0, q1};
      const unsigned w2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
      sink ^= w0 ^ w1 ^ w2;
      x[2 * pr] = x0 + 1e-7f; x[2 * pr + 1] = x1 + 1e-7f;   // keeps the chain from being hoisted
    }
  }
  if (sink == 0x12345678u) out[threadIdx.x] = 1.f;
}

// KIND 0: v_fma_f32, 1: v_cvt_pk_bf16_f32, 2: v_and / v_lshl, 3: v_sub_f32 -- 88 instructions per "tile", 8 independent chains
template <int KIND>
__device__ __forceinline__ void valu_chain(int iters, float* out) {
  float x[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = 1.0f + 0.001f * (threadIdx.x + i); u[i] = threadIdx.x * 2654435761u + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 11; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
        if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[(i + 1) & 7]));
        if (KIND == 2) asm volatile("v_and_b32 %0, 0xffff0000, %0\n\tv_lshlrev_b32 %1, 16, %1" : "+v"(u[i]), "+v"(u[(i + 3) & 7]));
        if (KIND == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
      }
  }
  float s = 0.f; unsigned t = 0;
  for (int i = 0; i < 8; ++i) { s += x[i]; t ^= u[i]; }
  if (s == 123.456f || t == 0x12345678u) out[threadIdx.x] = s;
}

__device__ __forceinline__ void lds_chain(int iters, float* out, float* lds) {
  // per "K-tile" and wave: 12 ds_read_b128 + 8 ds_write_b64 (wave-private region, conflict-free linear addresses)
  const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 3;
  float* base = lds + w * 2048;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      const f4 v = *reinterpret_cast<volatile f4*>(base + ((lane * 4 + k * 256) & 2047));
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<volatile f2*>(base + ((lane * 2 + k * 128) & 2047)) = f2{s.x, s.y};
    }
  }
  if (s.x == 123.456f) out[threadIdx.x] = s.x;
}

// LDS / global traffic of one wave per "tile" without dependent chains: KIND 0: 8 ds_write_b64, 1: 12 ds_read_b128 (one wait per
// tile), 2: 4 buffer-style global_load_dwordx4 from a 64 KB L2-resident region (one wait per tile)
template <int KIND>
__device__ __forceinline__ void mem_chain(int iters, float* out, float* lds, const float* gsrc) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, w = (threadIdx.x >> 6) & 3;
  float* base = lds + w * 2048;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) *reinterpret_cast<volatile f2*>(base + ((lane * 2 + k * 128) & 2047)) = f2{(float)it, (float)k};
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (KIND == 1) {
      f4 v[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) v[k] = *reinterpret_cast<volatile f4*>(base + ((lane * 4 + k * 256) & 2047));
#pragma unroll
      for (int k = 0; k < 12; ++k) acc += v[k];
    } else {
      f4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const volatile f4*>(gsrc + ((size_t)blockIdx.x * 4096 + (size_t)((it * 4 + k) & 3) * 1024 + w * 256 + lane * 4));
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += v[k];
    }
  }
  if (acc.x == 123.456f) out[threadIdx.x] = acc.x + acc.y;
}

__global__ void __launch_bounds__(512) probe(int mode, int iters, float* out, const float* gsrc) {
  __shared__ float lds[4 * 2048];
  if (threadIdx.x < 4 * 2048 / 8) for (int i = 0; i < 8; ++i) lds[threadIdx.x * 8 + i] = 1.0f;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  if (mode == 0 || mode == 3) mfma_chain(iters, out);
  else if (mode == 1 || mode == 4) conv_chain(iters, out);
  else if (mode == 2) { if (wave < 4) mfma_chain(iters, out); else conv_chain(iters, out); }
  else if (mode == 5) { if (wave < 4) mfma_chain(iters, out); else lds_chain(iters, out, lds); }
  else if (mode == 6) lds_chain(iters, out, lds);
  else if (mode == 7) { if (wave < 4) conv_chain(iters, out); else lds_chain(iters, out, lds); }
  else if (mode == 8) { if (wave < 4) mfma_chain_n<2>(iters, out); else conv_chain(iters, out); }
  else if (mode == 9) { if (wave < 4) mfma_chain_n<4>(iters, out); else conv_chain(iters, out); }
  else if (mode == 10) mfma_chain_n<4>(iters, out);
  else if (mode == 11) both_chain<1>(iters, out);
  else if (mode == 12) both_chain<2>(iters, out);
  else if (mode == 13) both_chain<4>(iters, out);
  else if (mode == 14) valu_chain<0>(iters, out);
  else if (mode == 15) { if (wave < 4) mfma_chain(iters, out); else valu_chain<0>(iters, out); }
  else if (mode == 16) valu_chain<1>(iters, out);
  else if (mode == 17) { if (wave < 4) mfma_chain(iters, out); else valu_chain<1>(iters, out); }
  else if (mode == 18) valu_chain<2>(iters, out);
  else if (mode == 19) { if (wave < 4) mfma_chain(iters, out); else valu_chain<2>(iters, out); }
  else if (mode == 20) valu_chain<3>(iters, out);
  else if (mode == 21) { if (wave < 4) mfma_chain(iters, out); else valu_chain<3>(iters, out); }
  else if (mode == 22) { if ((wave & 2) == 0) mfma_chain(iters, out); else conv_chain(iters, out); }   // waves 0,1,4,5 MFMA (SIMD 0,1 if waves go round-robin); 2,3,6,7 conversion
  else if (mode == 24) mem_chain<0>(iters, out, lds, gsrc);
  else if (mode == 25) { if (wave < 4) mfma_chain(iters, out); else mem_chain<0>(iters, out, lds, gsrc); }
  else if (mode == 26) mem_chain<1>(iters, out, lds, gsrc);
  else if (mode == 27) { if (wave < 4) mfma_chain(iters, out); else mem_chain<1>(iters, out, lds, gsrc); }
  else if (mode == 28) mem_chain<2>(iters, out, lds, gsrc);
  else if (mode == 29) { if (wave < 4) mfma_chain(iters, out); else mem_chain<2>(iters, out, lds, gsrc); }
  else if (mode == 30) { if (wave >= 4) mfma_chain(iters, out); else conv_chain(iters, out); }            // the MFMA waves are the YOUNGER ones
  else if (mode == 31) { if (wave >= 4) mfma_chain(iters, out); else mem_chain<1>(iters, out, lds, gsrc); }
  else if (mode == 32) { if (wave >= 4) mfma_chain(iters, out); else mem_chain<2>(iters, out, lds, gsrc); }
  else if (mode == 33) { if (wave < 4) { __builtin_amdgcn_s_setprio(0); mfma_chain(iters, out); } else { __builtin_amdgcn_s_setprio(3); conv_chain(iters, out); } }   // conversion waves at high priority
  else if (mode == 34) mfma_chain_yield<24, 0>(iters, out);
  else if (mode == 35) { if (wave < 4) mfma_chain_yield<24, 0>(iters, out); else conv_chain(iters, out); }
  else if (mode == 36) mfma_chain_yield<16, 0>(iters, out);
  else if (mode == 37) { if (wave < 4) mfma_chain_yield<16, 0>(iters, out); else conv_chain(iters, out); }
  else if (mode == 38) mfma_chain_yield<0, 1>(iters, out);
  else if (mode == 39) { if (wave < 4) mfma_chain_yield<0, 1>(iters, out); else conv_chain(iters, out); }
  else if (mode == 40) mfma_chain_yield<6, 0>(iters, out);
  else if (mode == 41) { if (wave < 4) mfma_chain_yield<6, 0>(iters, out); else conv_chain(iters, out); }
  else if (mode == 42) mfma_chain_yield<4, 0>(iters, out);
  else if (mode == 43) { if (wave < 4) mfma_chain_yield<4, 0>(iters, out); else conv_chain(iters, out); }
  else if (mode == 44) mfma_chain_yield<2, 0>(iters, out);
  else if (mode == 45) { if (wave < 4) mfma_chain_yield<2, 0>(iters, out); else conv_chain(iters, out); }
  else if (mode == 46) { if (wave < 4) mfma_chain_yield<6, 0>(iters, out); else mem_chain<1>(iters, out, lds, gsrc); }
  else if (mode == 47) { if (wave < 4) mfma_chain_yield<6, 0>(iters, out); else mem_chain<2>(iters, out, lds, gsrc); }
  else if (mode == 23) {   // roles from the hardware SIMD id
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const unsigned simd = (hw >> 4) & 3;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[512 + wave] = (float)simd;
    if (simd < 2) mfma_chain(iters, out); else conv_chain(iters, out);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  const int iters = 2000;
  const char* names[] = {"4 waves: MFMA chain (12 x 32x32x16 bf16 per tile)", "4 waves: conversion chain (8 pairs per tile)",
                         "8 waves: 4 MFMA + 4 conversion", "8 waves: all MFMA", "8 waves: all conversion",
                         "8 waves: 4 MFMA + 4 LDS (12 ds_read_b128 + 8 ds_write_b64 per tile)", "4 waves: LDS only", "8 waves: 4 conversion + 4 LDS",
                         "8 waves: 4 MFMA (2 accumulators) + 4 conversion", "8 waves: 4 MFMA (4 accumulators) + 4 conversion",
                         "4 waves: MFMA, 4 accumulators", "4 waves: each MFMA followed by a conversion slice, 1 accumulator",
                         "4 waves: each MFMA followed by a conversion slice, 2 accumulators", "4 waves: each MFMA followed by a conversion slice, 4 accumulators",
                         "4 waves: 88 v_fma_f32", "8 waves: 4 MFMA + 4 x 88 v_fma_f32", "4 waves: 88 v_cvt_pk_bf16_f32", "8 waves: 4 MFMA + 4 x 88 v_cvt_pk_bf16_f32",
                         "4 waves: 88 x (v_and + v_lshl)", "8 waves: 4 MFMA + 4 x 88 x (v_and + v_lshl)", "4 waves: 88 v_sub_f32", "8 waves: 4 MFMA + 4 x 88 v_sub_f32",
                         "8 waves: waves 0,1,4,5 MFMA, waves 2,3,6,7 conversion", "8 waves: MFMA on hardware SIMD 0-1, conversion on SIMD 2-3",
                         "4 waves: 8 ds_write_b64 per tile", "8 waves: 4 MFMA + 4 x 8 ds_write_b64", "4 waves: 12 ds_read_b128 per tile", "8 waves: 4 MFMA + 4 x 12 ds_read_b128",
                         "4 waves: 4 global_load_dwordx4 per tile (L2-resident)", "8 waves: 4 MFMA + 4 x 4 global_load_dwordx4",
                         "8 waves: 4 conversion (older waves) + 4 MFMA (younger)", "8 waves: 4 x 12 ds_read_b128 (older) + 4 MFMA (younger)",
                         "8 waves: 4 x 4 global loads (older) + 4 MFMA (younger)", "8 waves: 4 MFMA at priority 0 + 4 conversion at priority 3",
                         "4 waves: MFMA chain, s_nop 15 + s_nop 7 (96 clocks) after each", "8 waves: 4 MFMA (96 clocks of s_nop after each) + 4 conversion",
                         "4 waves: MFMA chain, s_nop 15 (64 clocks) after each", "8 waves: 4 MFMA (64 clocks of s_nop after each) + 4 conversion",
                         "4 waves: MFMA chain, s_sleep 1 after each", "8 waves: 4 MFMA (s_sleep 1 after each) + 4 conversion",
                         "4 waves: MFMA chain, s_nop 5 (24 clocks) after each", "8 waves: 4 MFMA (s_nop 5 after each) + 4 conversion",
                         "4 waves: MFMA chain, s_nop 3 (16 clocks) after each", "8 waves: 4 MFMA (s_nop 3 after each) + 4 conversion",
                         "4 waves: MFMA chain, s_nop 1 (8 clocks) after each", "8 waves: 4 MFMA (s_nop 1 after each) + 4 conversion",
                         "8 waves: 4 MFMA (s_nop 5 after each) + 4 x 12 ds_read_b128", "8 waves: 4 MFMA (s_nop 5 after each) + 4 x 4 global loads"};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float* gsrc; hipMalloc(&gsrc, 256 * 4096 * 4); hipMemset(gsrc, 0, 256 * 4096 * 4);
  for (int mode = 0; mode < 48; ++mode) {
    const int threads = (mode == 0 || mode == 1 || mode == 6 || (mode >= 10 && mode <= 14) || mode == 16 || mode == 18 || mode == 20 || mode == 24 || mode == 26 || mode == 28 || mode == 34 || mode == 36 || mode == 38 || mode == 40 || mode == 42 || mode == 44) ? 256 : 512;
    probe<<<256, threads>>>(mode, 10, out, gsrc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<<<256, threads>>>(mode, iters, out, gsrc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d  %-75s %8.1f us  = %6.0f cycles per tile at 2.4 GHz\n", mode, names[mode], ms * 1e3, ms * 1e-3 * 2.4e9 / iters);
    if (mode == 23) {
      float hs[8];
      hipMemcpy(hs, out + 512, sizeof(hs), hipMemcpyDeviceToHost);
      printf("         SIMD of waves 0..7 of workgroup 0:");
      for (int i = 0; i < 8; ++i) printf(" %d", (int)hs[i]);
      printf("\n");
    }
  }
  return 0;
}

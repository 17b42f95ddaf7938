// Counter-based dropout generator shared by every kernel that applies or replays a dropout mask.
// keep(idx) depends only on (seed, site, idx): forward and backward kernels with different thread
// layouts agree without storing the mask.  CPU restatement for tests: oracle/dropout_rng.py.
#pragma once
#include <stdint.h>

namespace mmnas {

__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

struct DropCfg {
  uint32_t seed_lo;   // low 32 bits of the call seed
  uint32_t site_key;  // site * 0x85EBCA77 + seed_hi
  uint32_t thresh;    // floor(p * 2^24); 0 disables dropout
  float scale;        // 1 / (1 - p)
};

__host__ inline DropCfg make_drop(float p, uint64_t seed, uint32_t site) {
  DropCfg c;
  c.seed_lo = (uint32_t)(seed & 0xFFFFFFFFull);
  c.site_key = site * 0x85EBCA77u + (uint32_t)(seed >> 32);
  if (p > 0.f) {
    c.thresh = (uint32_t)((double)p * 16777216.0);
    c.scale = 1.0f / (1.0f - p);
  } else {
    c.thresh = 0; c.scale = 1.0f;
  }
  return c;
}

// multiplier for element idx: 0 or 1/(1-p)
__device__ __forceinline__ float drop_mult(const DropCfg& c, uint32_t idx) {
  // one finaliser round over (golden-ratio-spread index + seed) ^ site key.  (A second round, as first written, bought
  // nothing measurable in the keep-rate / independence checks and doubled the integer work of every mask replay.)
  uint32_t h = fmix32((idx * 0x9E3779B1u + c.seed_lo) ^ c.site_key);
  return ((h >> 8) >= c.thresh) ? c.scale : 0.0f;
}

// The same decision from a PRE-MULTIPLIED index: pre = idx * DROP_G + seed_lo.  A kernel whose lane walks indices
// base + (compile-time constant) keeps base * DROP_G + seed_lo in a register and adds constant * DROP_G per element -- one add
// instead of a quarter-rate 32-bit multiply per element (the attention cores replay 64-256 decisions per lane).
constexpr uint32_t DROP_G = 0x9E3779B1u;
__device__ __forceinline__ uint32_t drop_pre(const DropCfg& c, uint32_t idx) { return idx * DROP_G + c.seed_lo; }
__device__ __forceinline__ float drop_mult_pre(const DropCfg& c, uint32_t pre) {
  const uint32_t h = fmix32(pre ^ c.site_key);
  return ((h >> 8) >= c.thresh) ? c.scale : 0.0f;
}

}  // namespace mmnas

// Shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mmnas_hip.h"
#include "rng.h"

namespace mmnas {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define MMNAS_REQUIRE(cond, code, ...) \
  do { if (!(cond)) { ::mmnas::set_error(__VA_ARGS__); return (code); } } while (0)

// Sum / maximum over the 64 lanes of a wave (all lanes active), the same value returned to every lane.  DPP lane
// permutations inside the vector ALU: quad swaps, half-row and row mirrors give every lane its 16-lane row's result, two row
// broadcasts carry it across the four rows into lane 63, a readlane hands it out.  (The __shfl_xor butterfly this replaces is six
// ds_bpermute_b32 round trips through the LDS crossbar: LayerNorm backward spent 24 of them per row.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xB1, 0xf>(0.f, v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xf>(0.f, v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x141, 0xf>(0.f, v);   // row_half_mirror
  v += dpp_move<0x140, 0xf>(0.f, v);   // row_mirror
  v += dpp_move<0x142, 0xa>(0.f, v);   // row_bcast:15 into rows 1 and 3
  v += dpp_move<0x143, 0xc>(0.f, v);   // row_bcast:31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  constexpr float NI = -__builtin_inff();
  v = fmaxf(v, dpp_move<0xB1, 0xf>(NI, v));
  v = fmaxf(v, dpp_move<0x4E, 0xf>(NI, v));
  v = fmaxf(v, dpp_move<0x141, 0xf>(NI, v));
  v = fmaxf(v, dpp_move<0x140, 0xf>(NI, v));
  v = fmaxf(v, dpp_move<0x142, 0xa>(NI, v));
  v = fmaxf(v, dpp_move<0x143, 0xc>(NI, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Row of the 32x32 MFMA accumulator held in register r of a lane in half `hh` (= lane >> 5):
// C/D layout of v_mfma_f32_32x32x2_f32: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * hh.
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// RAII timing scope: when profiling is enabled (mmnas_prof_enable) the kernel launches enclosed by the scope
// carry a start / stop HIP event in their own dispatch packets (hipExtLaunchKernelGGL through MMNAS_LAUNCH:
// device timestamps of the first kernel's begin and the last kernel's end on the launch stream, without
// the barrier packets a separate hipEventRecord costs -- bracketing with hipEventRecord slowed the
// training step by 12 %), tagged with the algorithmic work.
struct ProfScope {
  ProfScope(int kind, double flops, double bytes, hipStream_t st, const char* tag = nullptr);
  ~ProfScope();
  long idx_;
  hipStream_t st_;
};

struct ProfEvents { hipEvent_t start, stop; };
ProfEvents prof_launch_events();  // events for the next launch of the active scope ({null, null}: none)

// Every kernel launch of the library goes through this.
#define MMNAS_LAUNCH(kernel, grid, block, shmem, stream, ...)                                            \
  do {                                                                                                   \
    const ::mmnas::ProfEvents pe_ = ::mmnas::prof_launch_events();                                       \
    if (pe_.stop) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, pe_.start, pe_.stop, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                            \
  } while (0)

bool prof_enabled();

// Column reduction of per-workgroup partial rows that a kernel left behind (LayerNorm backward: part[nrows][3][d] ->
// out[w][c] += sum_rows part[row][w][c]).  Instead of its own launch it can ride on the next mmnas_gemm_pair launch as
// a few extra workgroups (internal: gemm_pair_aux); launch_aux_reduce() runs it stand-alone.
struct AuxReduce {
  const float* part;
  int nrows, d;
  float* out[3];   // NULL entries are skipped
};
int launch_aux_reduce(const AuxReduce& a, hipStream_t st);
int gemm_pair_aux(const mmnas_gemm_desc* dgrad, const mmnas_gemm_desc* wgrad, const AuxReduce* aux, hipStream_t st);
int gemm_wgrad_aux(const mmnas_gemm_desc* wgrad, const AuxReduce* aux, hipStream_t st);   // the same without a data-gradient half
// mmnas_layernorm_bwd without the final reduction: fills *aux (part == NULL when nothing is pending)
int layernorm_bwd_deferred(const float* x, const float* a, const float* dy, float* dx, float* da, float* db, float* ddrop,
                           float* dcol, float* ws, float drop_p, uint64_t seed, uint32_t site, int M, int d, float eps,
                           hipStream_t st, AuxReduce* aux);


// gemmln.hip: the merge / last FFN projection with its dropout + residual epilogue AND the LayerNorm behind it as one launch
// (row panels of 32 x 256; N = 256 only).  gemm_ln_applies() says whether *d qualifies (and the switch MMNAS_GEMM_LN is on);
// z = d->g[0].C (may be NULL: not stored), y = LayerNorm(z).
bool gemm_ln_applies(const mmnas_gemm_desc* d);
int gemm_ln(const mmnas_gemm_desc* d, const float* ln_a, const float* ln_b, float* y, int ldy, float eps, hipStream_t st);

// AttFlat with one glimpse (head.hip): the glimpse-logit layer as a matrix-vector product / an outer product + reductions
bool glimpse1_supported(int MID);
int glimpse1_fwd(const float* h0, const float* w0, const float* b0, float* l0, long rows0, const float* h1, const float* w1,
                 const float* b1, float* l1, long rows1, int MID, hipStream_t st);
// mixed.hip: mmnas_node_mix_bwd with the gate-gradient reduction left pending (partials in ws), and the reduction of many nodes
// lnb (round 6, architecture step): the sampled candidate's LayerNorm backward inside the same launch -- the kernel writes dz /
// dt / the parameter partials [nwg][3][d] (exactly ln_bwd_kernel's, <= 512 workgroups) instead of d_active
struct NodeLnBwd {
  float* dz; float* dt;    // gradient wrt the candidate's pre-LayerNorm sum; the same behind its output dropout (NULL: no dropout)
  float* part;             // partial rows (mmnas_layernorm_bwd_ws_floats(M, d) floats)
  DropCfg drop;            // the candidate's output dropout (site 1)
};
int node_mix_bwd_impl(const float* const* z, const float* const* ln_a, const float* const* ln_b, int n, const float* gate,
                      const float* dout, float* d_active, int active, float* dgate, float* ws, int M, int d, float eps,
                      hipStream_t stream, bool reduce, int* nwg_out, const NodeLnBwd* lnb = nullptr);
int mixed_reduce_many(const float* const* parts, float* const* dgates, const int* nwg, const int* n, int count, hipStream_t st);
int mha_core_fwd_pair(const mmnas_mha_desc* d0, const mmnas_mha_desc* d1, hipStream_t st);   // attention.hip: two cores, one launch
int transpose2d(const float* in, float* out, int R, int C, hipStream_t st);   // head.hip: out[c][r] = in[r][c]
int glimpse1_bwd_blocks(long rows, int MID);
int glimpse1_bwd(const float* dlog, const float* h, const float* w2, float gate_scale, int gated, float* dh, float* db1, float* dW2,
                 float* part, long rows, int MID, hipStream_t st, AuxReduce* aux);

// head.hip: AttFlat pooling over packed rows (ragged batches)
int attflat_pool_fwd_packed(const float* logits, const float* x, float* probs, float* pooled, int B, int S, int d, int G, const int* off, hipStream_t st);
int attflat_pool_bwd_packed(const float* probs, const float* x, const float* dpooled, float* dlogits, float* dx, int B, int S, int d, int G,
                            const int* off, hipStream_t st);

// rowops.hip: y = srcs[0] + ... + srcs[n-1] (count % 4 == 0, 16-byte aligned), n <= ADD_MANY_MAX
constexpr int ADD_MANY_MAX = 24;
int add_many(const float* const* srcs, int n, float* y, size_t count, hipStream_t st);

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace mmnas

/*
 * mmnas_hip.h -- C ABI of libmmnas_hip.so, the MI355X (gfx950) implementation of the MMNas
 * candidate-operator hot path.
 *
 * The reference (MILVLG/mmnas) is pure Python on ATen and has no FFI of its own; each entry
 * point below replaces the ATen kernel group behind one reference operator and cites it
 * (paths relative to the reference root; modules.py = mmnas/model/modules.py).  The binding a
 * maintainer adds on the reference side is the ctypes stub in INTEGRATION.md; ours is
 * mmnas_amd/_lib.py.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory (HBM) unless marked host.
 *   - all tensors are dense row-major fp32; masks are uint8 (non-zero = padded key).
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Every call only enqueues
 *     work on that stream: no allocation (one exception: mmnas_gemm's workspace, see there), no
 *     synchronisation, graph-capturable.
 *   - return value: 0 on success, negative MMNAS_E_* otherwise; mmnas_last_error() gives the text.
 *   - scratch/saved buffers are caller-owned; sizes come from the *_plan() functions.
 */
#ifndef MMNAS_HIP_H
#define MMNAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMNAS_ABI_VERSION 1

enum {
  MMNAS_OK = 0,
  MMNAS_E_SHAPE = -1,    /* unsupported / inconsistent shape (message says which) */
  MMNAS_E_ARG = -2,      /* null pointer or bad flag combination */
  MMNAS_E_LAUNCH = -3    /* hipGetLastError() after a launch */
};

/* operator flags */
enum {
  MMNAS_F_NORM = 1,      /* trailing LayerNorm (modules.py:268-269) */
  MMNAS_F_RESIDUAL = 2,  /* x + core(x) (modules.py:263-264) */
  MMNAS_F_MASK = 4,      /* key-padding mask present */
  MMNAS_F_REL = 8,       /* relation bias (RelMHAtt, modules.py:231-235) */
  MMNAS_F_SELF = 16,     /* query and key/value source are the same tensor */
  MMNAS_F_TRAIN = 32,    /* dropout active */
  MMNAS_F_RELRAW = 64    /* with MMNAS_F_REL: `rel` is the RAW [B,Sq,Sk,C] tensor + (Wy, by) -- lazy handle */
};

int mmnas_abi_version(void);
const char* mmnas_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Dropout generator.  nn.Dropout sites of the reference (modules.py:22,135-137,176,256,347)
 * draw from ATen's Philox stream; ours is counter-based: keep(idx) is a pure function of
 * (seed, site, idx) so backward kernels replay the forward mask instead of storing it
 * (definition: mmnas_amd/csrc/rng.h; CPU restatement for the tests: oracle/dropout_rng.py).
 * mmnas_dropout_mask writes the float multiplier (0 or 1/(1-p)) for idx = 0..n-1.
 * ------------------------------------------------------------------------------------------ */
int mmnas_dropout_mask(float* out, size_t n, float p, uint64_t seed, uint32_t site, void* stream);

/* ------------------------------------------------------------------------------------------
 * Grouped fp32 GEMM with fused epilogue.  Default arithmetic: every fp32 product as 6 v_mfma_f32_32x32x16_bf16 products of
 * operands split EXACTLY into three bf16 parts, fp32 accumulation (fp32-grade error, tests/test_kernels_gpu.py::
 * test_default_products_are_fp32_grade); MMNAS_GEMM_SPLIT=0 selects v_mfma_f32_32x32x2_f32 (exact fp32 fma chain), which
 * shapes outside the buffer-load path (K % 32 != 0, unaligned operands) always take.
 * Replaces the mm/addmm/bmm calls behind nn.Linear in modules.py:18,38,172-175 and their
 * autograd backward.  For group g (independent problems launched together):
 *     C_g[M_g,N] = epilogue( alpha * sum_{s<nseg} op(A_{g,s}) * op(B_{g,s}) )
 *   (Leading dimensions are plain row strides: a row stride SMALLER than the row length -- overlapping rows, lda = d with
 *   K = k d -- is how a k-tap Conv1d over the sequence (modules.py:472,480) is one product on the zero-padded input with no
 *   im2col buffer; reads past the M * lda extent of such an operand return zero.)
 *   layout MMNAS_GEMM_NT : A[M,K] (lda), B[N,K] (ldb)        y = x W^T        (forward)
 *   layout MMNAS_GEMM_NN : A[M,K] (lda), B[K,N] (ldb)        dx = dy W        (data gradient)
 *   layout MMNAS_GEMM_TN : A[K,M] (lda), B[K,N] (ldb)        dW = dy^T x      (weight gradient)
 *   epilogue, in this order: + bias[N]; relu; * dropout(seed, site, idx = row*N+col);
 *                            * (gate[row*ldgate+col] > 0 ? gate_scale : 0); + residual[row*ldres+col]
 *   accumulate != 0: the result is ADDED onto the values C holds (weight gradients accumulating into
 *   a flat gradient buffer); not combinable with relu / dropout.
 *   Scheduling is internal.  Plain products: whole tiles, with ragged tails / under-filled launches cut
 *   along K (stream-K); partial tiles meet in a per-stream workspace the library allocates on first use
 *   (64 MiB -- the one allocation any call makes) and are summed in a fixed order: bitwise reproducible,
 *   no float atomics.  Accumulating products: split-K, the pieces added with float atomics (summation order
 *   not fixed).  split_k is kept for source compatibility: a value > 1 only implies accumulate.
 * ------------------------------------------------------------------------------------------ */
enum { MMNAS_GEMM_NT = 0, MMNAS_GEMM_NN = 1, MMNAS_GEMM_TN = 2 };

typedef struct mmnas_gemm_group {
  int M;
  const float* A[3];
  const float* B[3];
  float* C;
  const float* bias;      /* [N] or NULL */
  const float* residual;  /* [M, ldres] or NULL */
  const float* gate;      /* [M, ldgate] or NULL */
  float* colsum;          /* [N] or NULL: colsum[n] += sum_m C[m, n] of the values just stored (the bias gradient of the
                             layer below when C is its pre-activation gradient); not with accumulate */
  uint64_t drop_seed;     /* != 0: this group's own dropout seed (drop_p / drop_site of the descriptor): the merge products
                             of a supernet node's candidates share a launch but not a mask */
} mmnas_gemm_group;

#define MMNAS_GEMM_MAX_GROUPS 9   /* e.g. the nine N = K = d projections of a decoder node's three attention candidates */

typedef struct mmnas_gemm_desc {
  int layout, ngroups, nseg;
  int N, K;
  int lda, ldb, ldc, ldres, ldgate;
  int relu, split_k;
  int accumulate;
  int b_planes;           /* != 0: B[i] of every group points at the pre-split bf16 planes of the weight matrix
                             (mmnas_split_planes: [3][N][ldb] bf16) instead of the fp32 matrix; the planes go global -> LDS
                             by LDS-DMA.  Layout NT with N % 64 == 0 and ldb % 8 == 0 only; same result bit for bit. */
  float alpha, gate_scale;
  float drop_p;           /* 0 = no dropout */
  uint32_t drop_site;
  uint64_t drop_seed;
  mmnas_gemm_group g[MMNAS_GEMM_MAX_GROUPS];
} mmnas_gemm_desc;

int mmnas_gemm(const mmnas_gemm_desc* d, void* stream);

/* y = LayerNorm(z), z = dropout(A W^T + bias) + residual: the merge projection of an attention operator / the last layer of
 * a feed-forward operator (modules.py:186-187, 261-271, 351-362: `x = self.norm(x + self.dropout(self.linear_merge(att)))`)
 * with the hand-written LayerNorm behind it (modules.py:44-56) as ONE launch.  *d describes the product exactly as for
 * mmnas_gemm (layout NT, one group; z = d->g[0].C with leading dimension ldc, may be NULL when nobody needs z); y has
 * d->N columns.  N = 256, K % 32 == 0 and 16-byte aligned operands run the row-panel kernel (32 x 256 panels, statistics
 * in-kernel) when the switch is ON (MMNAS_GEMM_LN=1 / mmnas_set_gemm_ln(1); default OFF: measured neutral on the supernet
 * step, profiles/r06_ab.txt); every other shape -- or the switch off -- runs mmnas_gemm + mmnas_layernorm_fwd
 * (z must then be given, ldc = N).  mmnas_set_gemm_ln returns the previous setting. */
int mmnas_gemm_ln(const mmnas_gemm_desc* d, const float* ln_a, const float* ln_b, float* y, float eps, void* stream);
int mmnas_set_gemm_ln(int on);

/* A weight matrix (nn.Linear.weight, modules.py:18,172-175) as three bf16 planes, planes[c * n + i] = part c of w[i]
 * with w[i] = part0 + part1 + part2 exactly -- the split mmnas_gemm otherwise performs on every K-tile it stages; done
 * once per optimizer step instead.  n % 8 == 0; `planes` holds 3 n bf16 (6 n bytes), 16-byte aligned. */
int mmnas_split_planes(const float* w, void* planes, size_t n, void* stream);

/* The two backward products of one linear layer -- data gradient dx = dy W (layout NN) and weight gradient
 * dW += dy^T x (layout TN, accumulate) -- which autograd issues as two independent mm calls
 * (torch/csrc/autograd: AddmmBackward / MmBackward of modules.py:18,38,172-175).  Results are those of two mmnas_gemm
 * calls; when both take the 64x64 buffer-load kernel they are issued as ONE launch whose second section starts as
 * the first one drains (saves one launch's idle start/end phases, ~10 us).  MMNAS_GEMM_PAIR=0 forces two launches.
 * The outputs must not overlap each other or either product's inputs. */
int mmnas_gemm_pair(const mmnas_gemm_desc* dgrad, const mmnas_gemm_desc* wgrad, void* stream);

/* Single-layer LSTM with zero initial state: nn.LSTM(batch_first=True) of the nets' language stem
 * (hygr_vqa.py:86-92 construction, :106-107 call; torch/nn/modules/rnn.py for the gate equations), one launch per time
 * step (recurrent product + gate arithmetic) instead of MIOpen's GEMM + pointwise kernels + weight-buffer copies.
 * Time-major buffers; gate columns INTERLEAVED: row / column 4j+g of the weight / gate arrays is gate g (i, f, g, o) of
 * hidden unit j, i.e. nn.LSTM's [i|f|g|o] row blocks permuted by the caller; bias = b_ih + b_hh likewise.
 *   fwd: x_tm [T,B,E]; Wih [4H,E]; Whh [4H,H]; xp [T,B,4H] scratch; Hall, Call [T+1,B,H] with slice 0 ZEROED by the
 *        caller (h_t, c_t land in slice t+1); Gall [T,B,4H] activated gates (saved for backward); out [B,T,H] = h_t.
 *   bwd: dout [B,T,H]; DG [T+1,B,4H] with slice T ZEROED (receives d pre-activations, slice t); dc [B,H] ZEROED
 *        (running cell gradient); scratch [B,H].  The weight / input gradients are ordinary products of DG[0:T]
 *        (TN with Hall[0:T] and x_tm, NN with Wih) issued by the caller through mmnas_gemm.
 * mmnas_lstm_supported: hidden size a multiple of 32. */
int mmnas_lstm_supported(int E, int H);
int mmnas_lstm_fwd(const float* x_tm, const float* Wih, const float* Whh, const float* bias, float* xp, float* Hall,
                   float* Call, float* Gall, float* out, int T, int B, int E, int H, void* stream);
int mmnas_lstm_bwd(const float* dout, const float* Whh, const float* Call, const float* Gall, float* DG, float* dc,
                   float* scratch, int T, int B, int H, void* stream);

/* The same LSTM (nn.LSTM(batch_first=True), one layer, zero initial state; hygr_vqa.py:86-92,106-107) as ONE persistent
 * launch per pass: the recurrent matrix is split over workgroups by hidden unit and stays in registers for the whole
 * sequence, the state vector is handed between workgroups once per step (agent-scope counter + acquire, no host
 * involvement).  Batch-first buffers, nn.LSTM's NATIVE gate order (rows / columns g*H + u, g in i,f,g,o): no weight
 * permutation, and the weight gradients below come out in the parameters' own layout.
 *   fwd: xp [B,T,4H] = x W_ih^T + b_ih (caller: one mmnas_gemm over the B*T rows); bhh [4H] (nullable);
 *        outputs Hprev [B,T,H] (Hprev[b][t] = h_{t-1}, slice 0 zero), Cs [B,T,H] (c_t), Gall [B,T,4H] (activated gates),
 *        out [B,T,H] (h_t).  Nothing needs zeroing by the caller.
 *   bwd: dout [B,T,H] -> DG [B,T,4H] (gradients of the pre-activations).  The caller finishes with ordinary products:
 *        dW_ih += DG^T x, dW_hh += DG^T Hprev (mmnas_gemm TN over the B*T rows), db += column sums of DG,
 *        dx = DG W_ih (mmnas_gemm NN).
 * Supported: H in {64, 128, 256, 512}, any T >= 1, B <= 960 (blocks of 32 / 16 samples run independently); needs a device
 * (the query asks it how many workgroups of the two kernels it holds at once).  Residency: the step barriers spin on the
 * other unit workgroups of a sample block, so a pass is cut into launches of whole sample blocks whose grid is resident at
 * once (occupancy query x CU count, one block per CU taken off when the query reports more than one); every batch of the
 * VQA / VGD configurations is one launch, the ITM batch of 160 at H = 512 two.
 * A step hand-off that gives up waiting (a workgroup not resident after all -- e.g. a co-running kernel held its slot)
 * sets a flag AND poisons the pass: a NaN goes into the output sequence (forward) / the gate gradients (backward), so the
 * loss or the gradient norm shows it without anybody polling.  mmnas_lstm_seq_timed_out(stream): synchronises and returns
 * 1 if that happened in the last pass on the stream, 0 otherwise -- tests / debugging. */
int mmnas_lstm_seq_supported(int H, int B);
int mmnas_lstm_seq_fwd(const float* xp, const float* bhh, const float* Whh, float* Hprev, float* Cs, float* Gall,
                       float* out, int T, int B, int H, void* stream);
int mmnas_lstm_seq_bwd(const float* dout, const float* Whh, const float* Cs, const float* Gall, float* DG, int T,
                       int B, int H, void* stream);
int mmnas_lstm_seq_timed_out(void* stream);

/* Scheduling knobs of mmnas_gemm (MMNAS_GEMM_TILE, _SK, _WGS, _MIN_UNITS, _GENERIC, _GM, _XCD, _LEAN: tuning and tests only;
 * _LEAN=0 keeps every product on the general kernel instead of the lean instantiations, whose whole-tile results are
 * bit-equal) and MMNAS_GEMM_SPLIT (6, the default: each fp32 product as 6 bf16-MFMA products of exactly split operands,
 * fp32 accumulation; 0: fp32 MFMA; 3 / 1: reduced-precision experiments) are read from the environment on the first call;
 * this re-reads them. */
int mmnas_gemm_reload_tuning(void);

/* ------------------------------------------------------------------------------------------
 * LayerNorm with Bessel-corrected std and eps added to the std (modules.py:52-56).
 *   fwd : y = a*(x-mean)/(std+eps)+b, x,y [M,d]
 *   bwd : dx [M,d]; da,db [d] are ACCUMULATED (atomics) -- zero them for a plain gradient.
 *         If ddrop != NULL it receives dx * dropout(seed, site, idx=row*d+col) (the gradient
 *         that flows into the operator core through the output dropout, modules.py:261).
 *         If dcol != NULL, dcol[d] += column sums of ddrop (bias gradient of the last linear).
 *         ws: NULL -> the column reductions use float atomics; otherwise a scratch buffer of
 *         mmnas_layernorm_bwd_ws_floats(M, d) floats -> per-workgroup partial rows + a second
 *         pass (no atomics on 3*d hot addresses, bitwise reproducible).
 * d % 4 == 0, d <= 2048.
 * ------------------------------------------------------------------------------------------ */
int mmnas_layernorm_fwd(const float* x, const float* a, const float* b, float* y,
                        int M, int d, float eps, void* stream);
int mmnas_layernorm_bwd(const float* x, const float* a, const float* dy, float* dx,
                        float* da, float* db, float* ddrop, float* dcol, float* ws,
                        float drop_p, uint64_t seed, uint32_t site,
                        int M, int d, float eps, void* stream);

size_t mmnas_layernorm_bwd_ws_floats(int M, int d);  /* host only */

/* out[N] += column sums of x[M,N] (bias gradients). */
int mmnas_colsum(const float* x, float* out, int M, int N, int ldx, void* stream);

/* Element-wise helpers (modules.py:96-119 and the bare registry entries ops_adapter.py:25-29).
 * kind: 0 zero, 1 relu, 2 leaky-relu(0.01), 3 gelu-tanh (modules.py:109).
 * bwd computes dx = dy * f'(x). */
int mmnas_eltwise_fwd(int kind, const float* x, float* y, size_t n, void* stream);
int mmnas_eltwise_bwd(int kind, const float* x, const float* dy, float* dx, size_t n, void* stream);
/* y = dropout(x) * scale-free copy with optional residual: y = res + x*dropmask (n elements, idx = i) */
int mmnas_drop_add(const float* x, const float* res, float* y, size_t n, float drop_p, uint64_t seed,
                   uint32_t site, void* stream);
/* nn.GLU over the last dim (modules.py:116-119): h [M,2C] -> y[M,C] = h[:, :C]*sigmoid(h[:, C:]);
 * optional relu and dropout after it (GLU(layers=2) unit_0 path, modules.py:145). */
int mmnas_glu_fwd(const float* h, float* y, int M, int C, int relu, float drop_p, uint64_t seed,
                  uint32_t site, void* stream);
int mmnas_glu_bwd(const float* h, const float* dy, float* dh, int M, int C, int relu, float drop_p,
                  uint64_t seed, uint32_t site, void* stream);

/* ------------------------------------------------------------------------------------------
 * Stem / head helpers.
 *   mmnas_row_is_zero: make_mask of the nets (hygr_vqa.py:121-122): mask[r] = 1 iff sum_j |f[r,j]| == 0
 *     (one pass over f [rows, d]).
 *   mmnas_attflat_pool_*: the pooling stage of AttFlat (modules.py:78-84) for logits [B,S,G], features x [B,S,d],
 *     key mask [B,S] (uint8, non-zero = padded, may be NULL):
 *       probs[b,s,g] = softmax over s of (mask ? -1e9 : logits);  pooled[b, g*d + j] = sum_s probs[b,s,g] x[b,s,j]
 *     bwd: dlogits [B,S,G] (zero at masked positions) and dx [B,S,d] (the pooling path's share of the feature
 *     gradient) from dpooled [B,G*d].
 *   S <= 1024.
 * ------------------------------------------------------------------------------------------ */
int mmnas_row_is_zero(const float* f, uint8_t* mask, long rows, int d, void* stream);
/* nn.Linear with ONE output unit (AttFlat's glimpse logits at ATTFLAT_GLIMPSES = 1: MLP.linear, modules.py:34-41): a
 * matrix-vector product instead of a GEMM launch with one live column, and its backward as one pass over x:
 *   fwd: y[r] = x[r,:] . w + b[0]                                  x [rows,K], w [K], b [1] or NULL, y [rows]
 *   bwd: dx[r,:] = dy[r] w;  dw[:] += sum_r dy[r] x[r,:];  db[0] += sum_r dy[r]   (db may be NULL)
 *        ws: mmnas_glimpse1_bwd_ws_floats(rows, K) floats of scratch (partial column sums, reduced by a second launch)
 * K % 4 == 0, K <= 1024 (mmnas_glimpse1_supported); the native head (mmnas_head_*) uses the same kernels, there with the
 * relu' / dropout replay of the hidden layer folded in. */
int mmnas_glimpse1_supported(int K);
size_t mmnas_glimpse1_bwd_ws_floats(long rows, int K);
int mmnas_glimpse1_fwd(const float* x, const float* w, const float* b, float* y, long rows, int K, void* stream);
int mmnas_glimpse1_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, float* ws, long rows, int K,
                       void* stream);
int mmnas_attflat_pool_fwd(const float* logits, const float* x, const uint8_t* mask, float* probs, float* pooled,
                           int B, int S, int d, int G, void* stream);
int mmnas_attflat_pool_bwd(const float* probs, const float* x, const uint8_t* mask, const float* dpooled,
                           float* dlogits, float* dx, int B, int S, int d, int G, void* stream);

/* Data path: box-geometry relation features of the loaders (relation_embedding, load_data_vqa.py:7-33 =
 * load_data_vgd.py:7-33; pinned by tests/golden/loader.npz), batched: bbox [B,S,4] (x1,y1,x2,y2), nobj [B] valid boxes per sample (NULL = S)
 * -> out [B,S,S,4] = (log max(|dcx|/w_i,1e-3), log max(|dcy|/h_i,1e-3), log(w_i/w_j), log(h_i/h_j)), zero padded. */
int mmnas_relation_embedding(const float* bbox, const int* nobj, float* out, int B, int S, void* stream);

/* Data path: token-relation features of the loaders (semantic_embedding, load_data_vqa.py:36-58), batched:
 * ques_ix [B,S] int64 token indices, nwords [B] = min(#words, S) per question, emb [V,E] the GloVe table ->
 * out [B,S,S,3] = (|g_i - g_j|_2, <g_i,g_j> / (sqrt|g_i| sqrt|g_j| + 1e-6), |i-j| / nwords), zero for i or j >= nwords.
 * S <= 64, E <= 320.  (64 x 301 floats of LDS per workgroup exceed the 64 KB default only for S > 54: the launch asks
 * for what it needs.) */
int mmnas_semantic_embedding(const long* ques_ix, const int* nwords, const float* emb, float* out, int B, int S,
                             int E, long V, void* stream);

/* Supernet plumbing: out [rows, width] = one-hot rows, out[r, idx_host[r]] = 1 (idx_host is a HOST array, read at
 * call time and passed in the kernel arguments; rows <= 128).  Writes the alpha_gate block of all MixedOps after
 * sampling (mixed.py:131-158) without a host->device copy / stream synchronisation. */
int mmnas_onehot_rows(float* out, int rows, int width, const int* idx_host, void* stream);

/* ------------------------------------------------------------------------------------------
 * MixedOp plumbing of the architecture step (mmnas/model/mixed.py).
 *   mmnas_mixed_sum_fwd: out = sum_j gate[j] * o_j over a node's n <= MMNAS_MIXED_MAX candidate outputs
 *     (MixedOp.forward in modes 'full' / 'two', mixed.py:59-68; replaces select + mul + add per candidate).
 *     outs_host is a HOST array of n device pointers (read at call time, carried in the kernel arguments); a NULL
 *     entry is a candidate that takes no part (mode 'two' evaluates two of them): skipped, its gate gradient += 0;
 *     gate is the node's alpha_gate row (device); count = elements per output, a multiple of 4.
 *   mmnas_mixed_sum_bwd: dgate[j] += <dout, o_j> for every candidate (the detached ones included: the gate, not the
 *     output, carries their gradient), d_active = gate[active] * dout (the one candidate that is differentiated;
 *     NULL to skip).  ws: mmnas_mixed_sum_ws_floats() floats of scratch (per-workgroup partials, summed in a fixed
 *     order by a second tiny launch: bitwise reproducible).
 *   mmnas_alpha_full_step: for every node r (row of the [rows, width] alpha_prob block; unused columns hold -inf):
 *     p = softmax(alpha_r); dalpha_i = sum_j g_j p_j (delta_ij - p_i) (set_arch_param_grad, mixed.py:194-198, 'full'
 *     mode) and the torch.optim.Adam update of alpha_optim (search_vqa.py:194) with its moments m, v [rows, width] at
 *     step `step` (1-based).  prob_grad (nullable) receives dalpha.  One launch instead of ~6 ATen kernels per node
 *     plus the optimizer's.
 * ------------------------------------------------------------------------------------------ */
#define MMNAS_MIXED_MAX 8
size_t mmnas_mixed_sum_ws_floats(void);   /* host only */
/* Node epilogue of the architecture step in one pass: out[M,d] = sum_j gate[j] * LN_j(z_j), LN_j the candidate's own
 * LayerNorm (modules.py:52-56; ln_a[j] == NULL: z_j is taken as the candidate's output as it stands) -- replaces the n
 * LayerNorm launches at the end of the n candidate operators plus the gated sum (mixed.py:59-68); the candidates' outputs
 * themselves are never stored.  z / ln_a / ln_b: HOST arrays of n device pointers (n <= MMNAS_MIXED_MAX, d <= 1024);
 * z[j] == NULL: candidate j of the node was not evaluated (mode 'two') -- no term in the sum, no gate gradient.
 * bwd: dgate[j] += <dout, LN_j(z_j)> (recomputed from z_j), d_active = gate[active] * dout (NULL: skip);
 * ws: mmnas_mixed_sum_ws_floats() floats. */
int mmnas_node_mix_fwd(const float* const* z, const float* const* ln_a, const float* const* ln_b, int n, const float* gate,
                       float* out, int M, int d, float eps, void* stream);
int mmnas_node_mix_bwd(const float* const* z, const float* const* ln_a, const float* const* ln_b, int n, const float* gate,
                       const float* dout, float* d_active, int active, float* dgate, float* ws, int M, int d, float eps,
                       void* stream);
int mmnas_mixed_sum_fwd(const float* const* outs_host, int n, const float* gate, float* out, size_t count,
                        void* stream);
int mmnas_mixed_sum_bwd(const float* const* outs_host, int n, const float* gate, const float* dout,
                        float* d_active, int active, float* dgate, float* ws, size_t count, void* stream);
int mmnas_alpha_full_step(float* prob, const float* gate_grad, float* m, float* v, float* prob_grad, int rows,
                          int width, float lr, float beta1, float beta2, float eps, int step, void* stream);

/* nn.Embedding backward (hygr_vqa.py:85,105; aten embedding_dense_backward): dW[idx[t], :] += dy[t, :] for the n_tok
 * int64 token indices -- straight into the (already zeroed or accumulating) gradient buffer instead of a dense
 * [V, E] temporary.  Indices outside [0, V) are ignored. */
int mmnas_embedding_bwd(const long* idx, const float* dy, float* dW, long n_tok, int E, long V, void* stream);
/* The same with a fixed summation order (token order, no atomics) and a scale: dW[idx[t]] += scale * dy[t].  For the
 * data-parallel exchange of the embedding gradient (mmnas_amd/dp.py RowExchange: ranks all-gather (idx, dy) -- ~1 MB --
 * instead of all-reducing the dense [V, E] table -- 24 MB, what DDP does at search_vqa.py:292 -- and every rank applies all
 * of them itself): the ranks' tables stay bitwise identical.  E <= 1024.  ws: mmnas_embedding_bwd_det_ws_floats(n_tok, E)
 * floats of scratch (per-chunk partial rows). */
size_t mmnas_embedding_bwd_det_ws_floats(long n_tok, int E);   /* host only */
int mmnas_embedding_bwd_det(const long* idx, const float* dy, float* dW, float* ws, long n_tok, int E, long V, float scale,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * Relation bias of RelMHAtt (modules.py:231-235):
 *   biasT[b,h,k,q] = log(max(relu(rel[b,q,k,:] . Wr[h,:] + br[h]), 1e-6))
 * rel [B,Sq,Sk,R] is read exactly once (the HBM-bound kernel of the path); the bias is written
 * key-major ([B,H,Sk,Sq]) so the attention core reads it coalesced along the query lanes.
 * bwd: dpre = dbiasT / r where r > 1e-6, else 0;  drel[b,q,k,:] = sum_h dpre*Wr[h,:] (overwritten
 * or, with accumulate_drel != 0, added to); dWr [H,R], dbr [H] accumulated with atomics.
 * R in {16,32,64,128,256}, H <= 32.
 * ------------------------------------------------------------------------------------------ */
int mmnas_rel_bias_fwd(const float* rel, const float* Wr, const float* br, float* biasT,
                       int B, int Sq, int Sk, int R, int H, void* stream);
int mmnas_rel_bias_bwd(const float* rel, const float* Wr, const float* br, const float* dbiasT,
                       float* drel, float* dWr, float* dbr, int accumulate_drel,
                       int B, int Sq, int Sk, int R, int H, void* stream);

/* ------------------------------------------------------------------------------------------
 * Lazy relation handle: the same bias straight from the RAW relation tensor raw[B,Sq,Sk,C]
 * (C = 4 box-geometry channels, hygr_vqa.py:111 / full_vqa.py:103; C = 3 for token relations),
 * fusing the stem's  rel = relu(raw Wy^T + by)  (linear_y_rel, [R,C] / [R]) with linear_r:
 *   biasT[b,h,k,q] = log(max(relu(relu(raw Wy^T + by) . Wr[h,:] + br[h]), 1e-6))
 * The [B,S,S,R] tensor, its gradient and their accumulation never exist.  bwd ACCUMULATES
 * dWy [R,C], dby [R], dWr [H,R], dbr [H] (no input gradient: raw is data); ws holds
 * mmnas_rel_fused_bwd_ws_floats(B,Sq,Sk) floats.  Supported: R = 64, C in {3,4}, H <= 32
 * (mmnas_rel_fused_supported); otherwise materialise rel and use mmnas_rel_bias_*.
 * ------------------------------------------------------------------------------------------ */
int mmnas_rel_fused_supported(int C, int R, int H);
int mmnas_rel_fused_fwd(const float* raw, const float* Wy, const float* by, const float* Wr,
                        const float* br, float* biasT, int B, int Sq, int Sk, int C, int R, int H,
                        void* stream);
size_t mmnas_rel_fused_bwd_ws_floats(int B, int Sq, int Sk);  /* host only */
int mmnas_rel_fused_bwd(const float* raw, const float* Wy, const float* by, const float* Wr,
                        const float* br, const float* dbiasT, float* dWy, float* dby, float* dWr,
                        float* dbr, float* ws, int B, int Sq, int Sk, int C, int R, int H,
                        void* stream);
/* The same for RAGGED batches (self-attention over the first n_b = off[b+1] - off[b] of the S rows of sample b; off = [B+1]
 * device prefix sums of the lengths): the padding rows / columns are masked keys resp. rows nothing downstream reads
 * (hygr_vqa.py:121-122, modules.py:195-196), their bias gradient is exactly zero.  Forward writes biasT[b,h,k,q] for
 * k, q < n_b only (padded [B,H,S,S] layout); backward reads dbiasT there only and walks the n_b x n_b valid elements of
 * every sample: tile_off = [B+1] device prefix sums of ceil(n_b^2 / 32), ntiles = their total (host value). */
int mmnas_rel_fused_fwd_ragged(const float* raw, const float* Wy, const float* by, const float* Wr, const float* br,
                               float* biasT, int B, int S, int C, int R, int H, const int* off, void* stream);
int mmnas_rel_fused_bwd_ragged(const float* raw, const float* Wy, const float* by, const float* Wr, const float* br,
                               const float* dbiasT, float* dWy, float* dby, float* dWr, float* dbr, float* ws,
                               int B, int S, int C, int R, int H, const int* off, const int* tile_off, int ntiles, void* stream);

/* ------------------------------------------------------------------------------------------
 * The relation bias of SEVERAL RelSelfAtt operators in one launch per direction (relmulti.hip, round 5).
 * Every relation operator of a network reads the same embedding rel = relu(linear_y_rel(raw)) -- ONE stem layer
 * (hygr_vqa.py:111, full_vqa.py:103) -- and its bias log(max(relu(linear_r(rel)), 1e-6)) (modules.py:231-235) depends on
 * nothing the backbone computes.  One call computes the hidden layer once per (b, q, k) element and
 *   fwd: writes biasT[n] [B,H,S,S] (key-major, as mmnas_rel_fused_fwd) for each of the n_ops operators from its own Wr[n] / br[n];
 *   bwd: reads every operator's dbiasT[n], ACCUMULATES dWr[n] / dbr[n] per operator and dWy / dby ONCE for all of them.
 * Self-attention only (S x S); R = 64, C in {3,4}, H <= 32, n_ops <= MMNAS_REL_MULTI_MAX (the call loops launches of 32
 * head rows backward / 96 forward); ragged batches as mmnas_rel_fused_*_ragged (off / tile_off / ntiles, else NULL / 0).
 * ws: mmnas_rel_multi_bwd_ws_floats(B, S) floats (backward only).  Results equal the per-operator calls to round-off.
 * ------------------------------------------------------------------------------------------ */
#define MMNAS_REL_MULTI_MAX 32
typedef struct mmnas_rel_multi {
  int B, S, C, R, H, n_ops;
  const float* raw;                 /* [B,S,S,C] */
  const float* Wy; const float* by; /* linear_{x,y}_rel: [R,C], [R] */
  float* dWy; float* dby;           /* bwd: accumulated (+=) */
  const float* Wr[MMNAS_REL_MULTI_MAX];      /* mhatt.linear_r.weight [H,R] per operator */
  const float* br[MMNAS_REL_MULTI_MAX];      /* mhatt.linear_r.bias [H] */
  float* biasT[MMNAS_REL_MULTI_MAX];         /* fwd out: [B,H,S,S] per operator */
  const float* dbiasT[MMNAS_REL_MULTI_MAX];  /* bwd in */
  float* dWr[MMNAS_REL_MULTI_MAX]; float* dbr[MMNAS_REL_MULTI_MAX];   /* bwd: accumulated (+=) */
  const int* off; const int* tile_off; int ntiles, reserved;          /* ragged batches (device arrays [B+1]) or NULL */
  float* ws;
} mmnas_rel_multi;
int mmnas_rel_multi_supported(int C, int R, int H);
size_t mmnas_rel_multi_bwd_ws_floats(int B, int S);   /* host only */
int mmnas_rel_multi_fwd(const mmnas_rel_multi* m, void* stream);
int mmnas_rel_multi_bwd(const mmnas_rel_multi* m, void* stream);
/* Backbone chains (mmnas_chain_*) compute the bias of all their lazy-handle relation operators this way: forward at chain
 * entry, backward behind the last relation operator of a stream.  mmnas_set_rel_hoist(0) / MMNAS_REL_HOIST=0 restores one
 * mmnas_rel_fused_* launch per operator (A/B runs); returns the previous setting. */
int mmnas_set_rel_hoist(int on);
/* Likewise the key / value projections of a chain's guided operators (all read the final language state; GuidedAtt,
 * modules.py:313-325): forward as grouped launches behind the encoder, backward (key / value source gradient, dWk, dWv) as
 * grouped gradient-pair launches behind the last guided operator.  MMNAS_GUIDED_HOIST=0 / mmnas_set_guided_hoist(0): one
 * launch set per operator; returns the previous setting. */
int mmnas_set_guided_hoist(int on);
/* The image stream's relation launches of a chain on a second stream beside the language stream's operators (forward: joined in
 * front of the decoder's first relation operator; backward: joined at the end of mmnas_chain_bwd).  OFF by default (measured
 * slower: +2.5 % on the supernet step); MMNAS_REL_OVERLAP=1 / mmnas_set_rel_overlap(1) enables it; returns the previous setting. */
int mmnas_set_rel_overlap(int on);

/* ------------------------------------------------------------------------------------------
 * Attention core: MHAtt.att (modules.py:191-199) for all (batch, head) pairs.
 *   Z = Q K^T / sqrt(dh) (+ biasT) ; Z[mask] = -1e9 ; P = softmax(Z) ; A = dropout(P) ; O = A V
 * Q [B*Sq, ldq], K,V [B*Sk, ldk/ldv], head h occupies columns [h*dh, (h+1)*dh).
* O [B*Sq, ldo] same column convention; lse [B,H,Sq,2] = (row max, 1/row sum) of each score row
 * (saved for backward instead of the map).  dropout idx = ((b*H+h)*Sq+q)*Sk+k.
 * Limits: dh in {16,32,64,128,256}; Sk <= 256.
 * bwd recomputes P from (Q,K,lse): dQ,dK,dV (same layouts as Q,K,V) are overwritten;
 * dbiasT [B,H,Sk,Sq] (nullable) receives dZ.
 * ------------------------------------------------------------------------------------------ */
typedef struct mmnas_mha_desc {
  int B, H, Sq, Sk, dh;
  int ldq, ldk, ldv, ldo;
  const float* Q; const float* K; const float* V;
  const uint8_t* mask;     /* [B,Sk] or NULL */
  const float* biasT;      /* [B,H,Sk,Sq] or NULL */
  float* O;                /* fwd: out.  bwd: in (forward output, for delta = rowsum(dO*O)) */
  float* lse;              /* row statistics [B,H,Sq,2] = (row max, 1/row sum): fwd out, bwd in */
  float drop_p; uint32_t drop_site; uint64_t drop_seed;
  /* backward only */
  const float* dO;         /* [B*Sq, ldo] */
  float* dQ; float* dK; float* dV;
  float* dbiasT;           /* [B,H,Sk,Sq] or NULL */
  float* delta;            /* scratch [B,H,Sq] */
  /* PACKED rows (ragged batches without their padding rows; d_h = 64, Sq, Sk <= 128): q_off / k_off = [B+1] device prefix
   * offsets -- batch b owns rows q_off[b] .. q_off[b+1] of Q / O / dO / dQ (k_off: of K / V / dK / dV), Sq / Sk are the
   * maximum lengths (strides of lse / delta / biasT, which keep their padded [B,H,...] layout).  Packed keys carry no
   * padding, so no mask goes with k_off.  The reference masks padded keys with -1e9 (modules.py:195-196): their
   * probability is exactly 0 in fp32, so leaving them out changes nothing.  NULL = dense rows b * Sq + q. */
  const int* q_off; const int* k_off;
} mmnas_mha_desc;

int mmnas_mha_core_fwd(const mmnas_mha_desc* d, void* stream);
int mmnas_mha_core_bwd(const mmnas_mha_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Operator level: one call = one reference operator forward (or backward).
 *
 * Attention family -- SelfAtt (modules.py:260-271), RelSelfAtt (:286-298), GuidedAtt (:313-325),
 * UniimgAtt (:415-428, caller concatenates x and y into xkv):
 *   y = LN( xq + drop( merge( att( xq Wq^T, xkv Wk^T, xkv Wv^T ) ) ) )
 * dropout sites: 0 = attention map, 1 = operator output.
 * ------------------------------------------------------------------------------------------ */
typedef struct mmnas_att_op {
  int B, Sq, Sk, d, di, H, dh, R;
  int flags;
  float drop_p, eps;
  uint64_t seed;
  const float* xq;         /* [B*Sq, d] */
  const float* xkv;        /* [B*Sk, d] (== xq when MMNAS_F_SELF) */
  const uint8_t* mask;     /* [B,Sk] */
  const float* rel;        /* [B,Sq,Sk,R] */
  const float* Wq; const float* Wk; const float* Wv;   /* [di, d]  mhatt.linear_{q,k,v}.weight */
  const float* Wm;         /* [d, di]  mhatt.linear_merge.weight */
  const float* Wr; const float* br;                    /* [H,R],[H] mhatt.linear_r */
  const float* ln_a; const float* ln_b;                /* [d] ln.a_2, ln.b_2 */
  float* y;                /* [B*Sq, d] */
  void* save;              /* saved-for-backward block, mmnas_att_op_plan().save_bytes */
  void* ws;                /* scratch, max(ws_fwd_bytes, ws_bwd_bytes) */
  /* backward */
  const float* dy;         /* [B*Sq, d] */
  float* dxq;              /* [B*Sq, d] overwritten (total input gradient when MMNAS_F_SELF) */
  float* dxkv;             /* [B*Sk, d] overwritten (ignored when MMNAS_F_SELF) */
  float* drel;             /* [B,Sq,Sk,R] overwritten, or NULL to skip */
  float* dWq; float* dWk; float* dWv; float* dWm;      /* accumulated (+=) */
  float* dWr; float* dbr; float* dln_a; float* dln_b;  /* accumulated (+=) */
  /* lazy relation handle (MMNAS_F_RELRAW): rel is raw[B,Sq,Sk,C]; linear_y_rel parameters + grads */
  int C;
  const float* Wy; const float* by;                    /* [R,C], [R] */
  float* dWy; float* dby;                              /* accumulated (+=) */
  /* PACKED rows (ragged batches without their padding rows; see mmnas_mha_desc.q_off): q_off != NULL -- xq / y / dy / dxq
   * hold Mq = q_off[B] rows, sample b owning rows q_off[b] .. q_off[b+1]; Sq is the maximum length.  k_off / Mk likewise
   * for xkv / dxkv (MMNAS_F_SELF: k_off = q_off; a guided operator keeps padded keys + mask: k_off = NULL).  With packed
   * keys there is no mask.  MMNAS_F_REL then needs MMNAS_F_RELRAW + MMNAS_F_SELF and the relation tiles of
   * mmnas_rel_fused_bwd_ragged (rel_tile_off [B+1] device, rel_ntiles host).  Sequences are prefixes: sample b's valid
   * rows are 0 .. n_b - 1 of its S. */
  const int* q_off; const int* k_off;
  int Mq, Mk;
  const int* rel_tile_off;
  int rel_ntiles, reserved2;
} mmnas_att_op;

typedef struct mmnas_plan {
  size_t save_bytes, ws_fwd_bytes, ws_bwd_bytes;
} mmnas_plan;

/* Sequences of <= 16 rows (the language stream: 14 tokens) with heads of 64, d in {256, 512} and B*H <= 256 run SelfAtt as ONE launch
 * forward (small.hip) instead of projection / core / merge / LayerNorm launches; the saved block is the same.
 * mmnas_set_small_ops(0) forces the general path (returns the previous setting; default on, env MMNAS_SMALL_OPS). */
int mmnas_set_small_ops(int on);
/* The same operators' BACKWARD (single-stream order): LayerNorm backward, d(att), the attention core's backward and the input
 * gradient as ONE launch, then dWm / dWq / dWk / dWv as one grouped launch carrying the LayerNorm parameter reduction -- 2 launches
 * instead of 4 dependent ones.  mmnas_set_small_bwd(0): the general backward (returns the previous setting; default on, env
 * MMNAS_SMALL_BWD).  Either backward follows either forward. */
int mmnas_set_small_bwd(int on);
/* FeedForward (d = 256, hidden 1024, <= 1024 rows) forward as ONE launch (small.hip: ffn_small_fwd_kernel).  Opt-in: measured
 * neutral against the two products + LayerNorm it replaces (default off, env MMNAS_SMALL_FFN; returns the previous setting). */
int mmnas_set_small_ffn(int on);
int mmnas_att_op_plan(const mmnas_att_op* op, mmnas_plan* plan);   /* host only */
int mmnas_att_op_fwd(const mmnas_att_op* op, void* stream);
int mmnas_att_op_bwd(const mmnas_att_op* op, void* stream);

/* ------------------------------------------------------------------------------------------
 * MLP family: a chain of nl (1..3) Linear layers with ReLU+dropout between them, then the common
 * epilogue.  FeedForward (modules.py:351-362): nl = 2, dims {d, mid_k*d, d};
 * FeedForward_deep (:389-400): nl = 3, dims {d, 2d, 2d, d}.
 *   h_0 = x ; h_{i+1} = drop_i(relu(h_i W_i^T + b_i)) for i < nl-1 ; core = h_{nl-1} W^T + b
 *   y = LN( x + drop_out(core) )
 * dropout sites: hidden layer i -> site {0,2}[i], operator output -> site 1.
 * ------------------------------------------------------------------------------------------ */
typedef struct mmnas_mlp_op {
  int M, nl;
  int dims[4];             /* dims[0] = dims[nl] = d */
  int flags;
  float drop_p, eps;
  uint64_t seed;
  const float* x;          /* [M, d] */
  const float* W[3];       /* W[i]: [dims[i+1], dims[i]] */
  const float* b[3];
  const float* ln_a; const float* ln_b;
  float* y;
  void* save; void* ws;
  const float* dy;
  float* dx;               /* overwritten */
  float* dW[3]; float* db[3];                          /* accumulated (+=) */
  float* dln_a; float* dln_b;
} mmnas_mlp_op;

int mmnas_mlp_op_plan(const mmnas_mlp_op* op, mmnas_plan* plan);
int mmnas_mlp_op_fwd(const mmnas_mlp_op* op, void* stream);
int mmnas_mlp_op_bwd(const mmnas_mlp_op* op, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backbone chain: every cell operator of a backbone in ONE call per direction (Backbone_*.forward, hygr_vqa.py:45-52,
 * full_vqa.py:46-53; Cell_*.forward, hygr_vqa.py:23-27, for single-operator nodes of the attention / MLP families).
 *   ops[]: host array in evaluation order, encoder nodes (on_y = 0: language stream x, [B*Sx, d]) first, then decoder
 *   nodes (on_y = 1: image stream y, [B*Sy, d]; a guided operator reads the FINAL language state).  Each record holds
 *   an operator descriptor as for mmnas_att_op_* / mmnas_mlp_op_* with the parameters, flags (NORM, RESIDUAL, TRAIN, SELF,
 *   REL, RELRAW), dropout rate / seed, R, C, Wy / by and -- for backward -- the parameter-gradient pointers filled in;
 *   shapes, masks, relation tensors and every activation / scratch pointer are set by the chain.
 *   arena: mmnas_chain_plan() bytes, written by fwd, read by bwd; it must stay alive until the backward's side-stream
 *   work has been joined (mmnas_chain_join).
 *   fwd: x_out [B*Sx,d], y_out [B*Sy,d].  bwd: dx_out (nullable = 0), dy_out -> dx_in, dy_in; parameter gradients are
 *   ACCUMULATED as by the per-operator calls.
 *   use_side_stream != 0 (bwd): the call's stream carries only the data-gradient chain; weight-gradient products,
 *   LayerNorm parameter reductions and the relation-bias backward run on a library-owned side stream behind one
 *   event per operator (use_side_stream == 2: only the relation-bias backward moves).  mmnas_chain_join(main, waiting) makes `waiting` wait for everything issued to main's side
 *   stream so far -- call it (with waiting = main) before anything reads the parameter gradients.
 * ------------------------------------------------------------------------------------------ */
#define MMNAS_CHAIN_MAX_OPS 128
enum { MMNAS_CHAIN_ATT = 0, MMNAS_CHAIN_MLP = 1 };
typedef struct mmnas_chain_op {
  int kind, on_y;
  mmnas_att_op att;
  mmnas_mlp_op mlp;
  /* mixed chains (mmnas_chain.mixed = 1; MixedOp.forward in modes 'full' / 'two', mixed.py:59-68): consecutive operators
   * with the same `node` are the evaluated candidates of one supernet node -- all read the node's input, the node's
   * output is sum_j gate[node][cand_j] * output_j.  Exactly one of them has detached = 0: the sampled candidate, the only
   * one differentiated; the others contribute their output to the sum and to the gate gradients only. */
  int node, cand, detached, reserved;
} mmnas_chain_op;
typedef struct mmnas_chain {
  int n_ops;
  const mmnas_chain_op* ops;
  int B, Sx, Sy, d;
  const float* x_in; const float* y_in;
  const uint8_t* x_mask; const uint8_t* y_mask;   /* [B,Sx], [B,Sy] or NULL */
  const float* x_rel; const float* y_rel;         /* relation tensors of the REL operators (raw with RELRAW) or NULL */
  void* arena;
  float* x_out; float* y_out;
  const float* dx_out; const float* dy_out;
  float* dx_in; float* dy_in;
  int use_side_stream, reserved;
  /* bwd, optional: marks[i] != NULL is a hipEvent_t recorded on the issuing stream right after operator i's backward
   * launches -- a data-parallel reducer waits on it to start a bucket's all-reduce while the operators in front of i
   * (issued later) still run.  NULL: no marks. */
  void* const* marks;
  /* architecture step: gate / dgate = the [n_nodes, gate_width] blocks of the nodes' binary gates and of their gradients
   * (row = node, column = candidate; dgate[node][cand] += <d node output, candidate output>).  Candidates' LayerNorms
   * and the gated sum run as one kernel per node and direction (mmnas_node_mix_fwd/bwd). */
  int mixed, gate_width;
  const float* gate;
  float* dgate;
  /* RAGGED decoder stream: y_off != NULL -- y_in / y_out / dy_out / dy_in hold Ny = y_off[B] PACKED rows (sample b: rows
   * y_off[b] .. y_off[b+1], its first n_b regions; the caller packs and unpacks), every decoder operator runs on Ny rows,
   * self-attention over a sample's own rows without a mask, guided attention from packed queries to the padded language
   * keys (x_mask).  The padding rows of the reference's decoder stream are masked as keys everywhere and dropped by
   * AttFlat's mask (hygr_vqa.py:113-122, modules.py:78-84,195-196): no logit and no parameter gradient depends on
   * them.  y_mask is ignored; y_rel stays the padded raw [B,Sy,Sy,C] tensor; y_tile_off / y_ntiles: see
   * mmnas_rel_fused_bwd_ragged. */
  const int* y_off;
  const int* y_tile_off;
  int Ny, y_ntiles;
} mmnas_chain;
int mmnas_chain_plan(const mmnas_chain* c, size_t* arena_bytes);   /* host only */
int mmnas_chain_fwd(const mmnas_chain* c, void* stream);
int mmnas_chain_bwd(const mmnas_chain* c, void* stream);
int mmnas_chain_join(void* main_stream, void* waiting_stream);
/* Encoder / decoder overlap inside mmnas_chain_fwd/bwd: the language-stream operators on their own (high-priority)
 * stream beside the image-stream operators that precede the first guided operator; joined before the call returns its
 * last launch.  Measured: no gain on MI355X, so off by default (env MMNAS_CHAIN_OVERLAP=1 or this
 * call enable it; returns the previous setting).  Results are identical either way. */
int mmnas_set_chain_overlap(int on);

/* ------------------------------------------------------------------------------------------
 * Answer head: AttFlat over the language state + AttFlat over the image state, their sum, LayerNorm and the answer
 * projection (Net_*.forward, hygr_vqa.py:113-119 / full_vqa.py:105-114; AttFlat modules.py:59-85; MLP / FC
 * modules.py:13-41) in one call per direction.
 *   side: x [B*S, d], key mask [B,S] (nullable), MLP fc (W1 [MID,d], b1), glimpse linear (W2 [G,MID], b2), merge
 *   (Wm [OUT, G*d], bm); seed = the dropout seed of the side's FC (site 0).
 *   fwd: logits [B, ANS] = proj(LN(attflat_x + attflat_y)).   bwd: dlogits -> sx.dx, sy.dx (overwritten); every
 *   parameter gradient is ACCUMULATED (+=).  arena: mmnas_head_plan() bytes, written by fwd, read by bwd.
 * Also the loss of the VQA scripts (BCEWithLogitsLoss(reduction='sum'), search_vqa.py:211 / train_vqa.py:237) as two
 * kernels: mmnas_bce_logits_sum_fwd ADDS sum(max(x,0) - x t + log1p(exp(-|x|))) to loss[0] (zero it first);
 * mmnas_bce_logits_bwd writes dlogits = go[0] * (sigmoid(x) - t) (go: device scalar, the loss's upstream gradient).
 * ------------------------------------------------------------------------------------------ */
typedef struct mmnas_attflat_side {
  int S, reserved;
  const float* x; const uint8_t* mask;
  const float* W1; const float* b1; const float* W2; const float* b2; const float* Wm; const float* bm;
  float* dW1; float* db1; float* dW2; float* db2; float* dWm; float* dbm;
  uint64_t seed;
  float* dx;
  /* PACKED rows (ragged batches without their padding rows, the image side of a ragged decoder stream): off != NULL -- x / dx
   * hold M = off[B] rows, sample b owning rows off[b] .. off[b+1]; S is the longest sample; `mask` is ignored (every packed
   * row is valid -- the reference masks the padding rows out of AttFlat's softmax, modules.py:78-81: same result). */
  const int* off;
  int M, reserved2;
} mmnas_attflat_side;
typedef struct mmnas_head {
  int B, d, MID, G, OUT, ANS, flags, reserved;
  float drop_p, eps;
  mmnas_attflat_side sx, sy;
  const float* ln_a; const float* ln_b; float* dln_a; float* dln_b;
  const float* Wp; const float* bp; float* dWp; float* dbp;
  float* logits;
  const float* dlogits;
  void* arena;
} mmnas_head;
int mmnas_head_plan(const mmnas_head* hd, size_t* arena_bytes);   /* host only */
int mmnas_head_fwd(const mmnas_head* hd, void* stream);
int mmnas_head_bwd(const mmnas_head* hd, void* stream);
int mmnas_bce_logits_sum_fwd(const float* logits, const float* target, float* loss, size_t n, void* stream);
int mmnas_bce_logits_bwd(const float* logits, const float* target, const float* go, float* dlogits, size_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * 1-D convolutions over the sequence axis of x[B,S,d] (channels last, zero "same" padding, odd
 * kernel size k <= 11): building blocks of StdConv (modules.py:465-491: im2col -> mmnas_gemm) and
 * SepConv (modules.py:431-462: depthwise stencil -> pointwise mmnas_gemm).
 *   im2col : col[m, t*d + c] = x[b, s+t-k/2, c] (0 outside the sequence), m = b*S+s
 *   col2im : dx[m, c] = sum_t dcol[(b, s-t+k/2), t*d + c]            (its adjoint)
 *   dwconv : y[m,c] = bias[c] + sum_t w[c,t] x[b, s+t-k/2, c]   (w = depthwise_conv.weight [d,1,k])
 *            bwd overwrites dx and ACCUMULATES dw [d,k], db [d].
 * ------------------------------------------------------------------------------------------ */
int mmnas_im2col_seq(const float* x, float* col, int B, int S, int d, int k, void* stream);
/* The zero-padded row grid of the window-buffer-free dense convolution (modules.py:472,480; see mmnas_gemm on overlapping
 * rows): xp[rows_total, d]; row b * Sp + j holds x[b, j - front, :] for 0 <= j - front < S, zero otherwise (also every row
 * from B * Sp on: slack the overlapping rows of the last sequence read).  d % 4 == 0. */
int mmnas_pad_seq(const float* x, float* xp, int B, int S, int d, int front, int Sp, long rows_total, void* stream);
int mmnas_col2im_seq(const float* dcol, float* dx, int B, int S, int d, int k, void* stream);
int mmnas_dwconv_seq_fwd(const float* x, const float* w, const float* bias, float* y,
                         int B, int S, int d, int k, void* stream);
int mmnas_dwconv_seq_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw,
                         float* db, int B, int S, int d, int k, void* stream);

/* ------------------------------------------------------------------------------------------
 * Data-parallel helper: gather/scatter a list of gradient segments into/from one contiguous
 * staging buffer so that a sampled sub-network's gradients travel in a single RCCL all-reduce
 * (replaces DDP's bucket copies, search_vqa.py:210,292).  `segs` is a device array of
 * nseg {ptr, offset(floats), n(floats)} records; direction 0 = pack (src -> staging*scale),
 * 1 = unpack (staging*scale -> dst).
 * ------------------------------------------------------------------------------------------ */
typedef struct mmnas_segment { float* ptr; uint64_t offset; uint64_t n; } mmnas_segment;
int mmnas_pack_segments(const mmnas_segment* segs, int nseg, float* staging, float scale,
                        int direction, void* stream);
/* Same, with `segs` a HOST array read at call time and carried in the kernel arguments (chunks of 96 records):
 * no host->device copy of the table, hence no stream synchronisation per step. */
/* Ragged batches: padded [B, S, d] <-> packed [off[B], d] rows; sample b's valid rows are its first off[b+1] - off[b] (the
 * loaders pad region features behind the detected boxes, load_data_vqa.py:221-246).  unpack writes zeros into the padding
 * rows.  d % 4 == 0.  Used around mmnas_chain with y_off. */
int mmnas_pack_rows(const float* x, const int* off, float* packed, int B, int S, int d, void* stream);
int mmnas_unpack_rows(const float* packed, const int* off, float* x, int B, int S, int d, void* stream);
int mmnas_pack_segments_host(const mmnas_segment* segs_host, int nseg, float* staging, float scale,
                             int direction, void* stream);

/* Fused Adam over a flat fp32 parameter buffer (net_optim.step(), search_vqa.py:300 through
 * mmnas/utils/optimizer.py): torch.optim.Adam arithmetic with bias correction at `step`.
 * If sumsq != NULL (device scalar holding the squared global gradient norm) the gradient is first
 * scaled by min(1, max_norm / (sqrt(*sumsq) + 1e-6)) -- clip_grad_norm_, search_vqa.py:296-298 --
 * without a host round trip. */
int mmnas_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, const float* sumsq, float max_norm,
                    int step, void* stream);
/* out[0] += sum of squares of g[0..n) (for clip_grad_norm_). */
int mmnas_sumsq(const float* g, size_t n, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement aid (bench.py roofline): while enabled, every kernel launch of the classes below is
 * bracketed by HIP events recorded on the stream it is launched on and tagged with its ALGORITHMIC
 * flops / bytes; mmnas_prof_collect() synchronises the events, sums per class and resets.
 * ------------------------------------------------------------------------------------------ */
enum { MMNAS_K_GEMM = 0, MMNAS_K_MHA_FWD = 1, MMNAS_K_MHA_BWD = 2, MMNAS_K_REL_FWD = 3, MMNAS_K_REL_BWD = 4,
       MMNAS_K_ROWOPS = 5, MMNAS_K_LSTM = 6, MMNAS_K_SMALL = 7 /* fused short-sequence operators */, MMNAS_K_COUNT = 8 };
typedef struct mmnas_prof_stat { double ms, flops, bytes; long launches; } mmnas_prof_stat;
int mmnas_prof_enable(int on);
int mmnas_prof_collect(mmnas_prof_stat* stats /* [MMNAS_K_COUNT] */);

#ifdef __cplusplus
}
#endif
#endif /* MMNAS_HIP_H */

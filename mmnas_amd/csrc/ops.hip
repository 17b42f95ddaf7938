// Operator-level entry points: one call enqueues every kernel of one reference operator forward
// (or backward) on the caller's stream.  Host-side orchestration only -- the arithmetic lives in
// gemm.hip / attention.hip / relbias.hip / rowops.hip.
//
//   attention family (SelfAtt modules.py:260-271, RelSelfAtt :286-298, GuidedAtt :313-325,
//   UniimgAtt :415-428):
//     fwd:  [Q|K|V] = grouped GEMM -> (rel bias) -> attention core -> merge GEMM with fused
//           dropout+residual epilogue -> LayerNorm                            (4-5 launches)
//     bwd:  LN bwd (+dropout replay) -> d(att) GEMM, dWm GEMM -> core bwd (3 launches) ->
//           dW{q,k,v} grouped GEMM -> dx GEMM(s) with fused residual add -> (rel bias bwd)
//   MLP family (FeedForward modules.py:351-362, FeedForward_deep :389-400).
#include <string.h>
#include <functional>
#include <vector>
#include "common.h"

namespace mmnas {

static inline size_t al(size_t n) { return (n + 255) & ~(size_t)255; }

struct Carver {
  char* base; size_t off;
  explicit Carver(void* p) : base((char*)p), off(0) {}
  float* take(size_t nfloats) { float* r = (float*)(base + off); off += al(nfloats * sizeof(float)); return r; }
};

static void gemm_init(mmnas_gemm_desc& g, int layout, int N, int K, int lda, int ldb, int ldc) {
  memset(&g, 0, sizeof(g));
  g.layout = layout; g.ngroups = 1; g.nseg = 1; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = 1.f; g.gate_scale = 1.f; g.split_k = 1;
}

struct AttLayout {
  size_t Mq, Mk;
  float *Q, *K, *V, *att, *stats, *z, *biasT;       // save
  float *dz, *dt, *datt, *dQ, *dK, *dV, *delta, *dbiasT, *lnws, *relws;  // backward scratch
  size_t save_bytes, ws_bwd_bytes;
};

static AttLayout att_layout(const mmnas_att_op* op) {
  AttLayout L;
  L.Mq = op->q_off ? (size_t)op->Mq : (size_t)op->B * op->Sq;     // (packed rows: the sum of the sequences' lengths)
  L.Mk = op->k_off ? (size_t)op->Mk : (size_t)op->B * op->Sk;
  const bool norm = op->flags & MMNAS_F_NORM, rel = op->flags & MMNAS_F_REL;
  const bool drop = (op->flags & MMNAS_F_TRAIN) && op->drop_p > 0.f;
  Carver s(op->save);
  L.Q = s.take(L.Mq * op->di); L.K = s.take(L.Mk * op->di); L.V = s.take(L.Mk * op->di);
  L.att = s.take(L.Mq * op->di);
  L.stats = s.take((size_t)op->B * op->H * op->Sq * 2);
  L.z = norm ? s.take(L.Mq * op->d) : nullptr;
  L.biasT = rel ? s.take((size_t)op->B * op->H * op->Sk * op->Sq) : nullptr;
  L.save_bytes = s.off;
  Carver w(op->ws);
  L.dz = norm ? w.take(L.Mq * op->d) : nullptr;
  L.dt = drop ? w.take(L.Mq * op->d) : nullptr;
  L.datt = w.take(L.Mq * op->di);
  L.dQ = w.take(L.Mq * op->di); L.dK = w.take(L.Mk * op->di); L.dV = w.take(L.Mk * op->di);
  L.delta = w.take((size_t)op->B * op->H * op->Sq);
  L.dbiasT = rel ? w.take((size_t)op->B * op->H * op->Sk * op->Sq) : nullptr;
  L.lnws = norm ? w.take(mmnas_layernorm_bwd_ws_floats((int)L.Mq, op->d)) : nullptr;
  L.relws = (rel && (op->flags & MMNAS_F_RELRAW)) ? w.take(mmnas_rel_fused_bwd_ws_floats(op->B, op->Sq, op->Sk)) : nullptr;
  L.ws_bwd_bytes = w.off;
  return L;
}

// small.hip: one-launch forms for short sequences
bool sa_small_applies(const mmnas_att_op* op);
int sa_small_fwd(const mmnas_att_op* op, float* Q, float* K, float* V, float* att, float* stats, float* z, hipStream_t st);
bool ffn_small_applies(const mmnas_mlp_op* op);
int ffn_small_fwd(const mmnas_mlp_op* op, float* h, float* z, hipStream_t st);
bool sa_small_bwd_applies(const mmnas_att_op* op);
int sa_small_bwd(const mmnas_att_op* op, const float* Q, const float* K, const float* V, const float* stats, const float* z,
                 float* dt_out, float* dQ, float* dK, float* dV, float* lnpart, hipStream_t st);

static int att_check(const mmnas_att_op* op, const char* who) {
  MMNAS_REQUIRE(op, MMNAS_E_ARG, "%s: null descriptor", who);
  MMNAS_REQUIRE(op->B > 0 && op->Sq > 0 && op->Sk > 0 && op->d > 0 && op->di > 0, MMNAS_E_SHAPE,
                "%s: B=%d Sq=%d Sk=%d d=%d di=%d", who, op->B, op->Sq, op->Sk, op->d, op->di);
  MMNAS_REQUIRE(op->H * op->dh == op->di, MMNAS_E_SHAPE, "%s: H*dh=%d*%d != di=%d", who, op->H, op->dh, op->di);
  MMNAS_REQUIRE(op->d % 4 == 0 && op->di % 4 == 0, MMNAS_E_SHAPE, "%s: d=%d di=%d must be multiples of 4", who,
                op->d, op->di);
  if (op->flags & MMNAS_F_SELF) MMNAS_REQUIRE(op->Sq == op->Sk, MMNAS_E_SHAPE, "%s: SELF needs Sq == Sk", who);
  if (op->q_off || op->k_off) {   // packed rows
    MMNAS_REQUIRE(op->q_off && op->Mq > 0 && op->Mq <= op->B * op->Sq, MMNAS_E_ARG, "%s: packed rows: q_off / Mq=%d", who, op->Mq);
    MMNAS_REQUIRE(!op->k_off || (op->Mk > 0 && op->Mk <= op->B * op->Sk), MMNAS_E_ARG, "%s: packed rows: Mk=%d", who, op->Mk);
    if (op->flags & MMNAS_F_SELF) MMNAS_REQUIRE(op->k_off == op->q_off && op->Mk == op->Mq, MMNAS_E_ARG, "%s: packed SELF needs k_off == q_off", who);
    MMNAS_REQUIRE(!(op->k_off && (op->flags & MMNAS_F_MASK)), MMNAS_E_ARG, "%s: packed keys carry no mask", who);
    if (op->flags & MMNAS_F_REL)
      MMNAS_REQUIRE((op->flags & MMNAS_F_RELRAW) && (op->flags & MMNAS_F_SELF) && op->rel_tile_off && op->rel_ntiles >= 0, MMNAS_E_ARG,
                    "%s: a relation bias on packed rows needs the lazy handle (RELRAW), SELF and the relation tile offsets", who);
  }
  return MMNAS_OK;
}

}  // namespace mmnas

using namespace mmnas;

extern "C" int mmnas_att_op_plan(const mmnas_att_op* op, mmnas_plan* plan) {
  int rc = att_check(op, "att_op_plan");
  if (rc) return rc;
  MMNAS_REQUIRE(plan, MMNAS_E_ARG, "att_op_plan: null plan");
  mmnas_att_op tmp = *op;
  tmp.save = nullptr; tmp.ws = nullptr;
  AttLayout L = att_layout(&tmp);
  plan->save_bytes = L.save_bytes;
  plan->ws_fwd_bytes = 256;
  plan->ws_bwd_bytes = L.ws_bwd_bytes;
  return MMNAS_OK;
}

namespace mmnas {
// defer_ln (mixed chains): with NORM the operator stops behind its residual sum z (saved block) and writes no output --
// the node epilogue (mmnas_node_mix_fwd) normalises every candidate of the node and forms the gated sum in one pass.
// *ln_done tells the caller whether the output was normalised here after all (the one-launch short-sequence kernel).
// rel_ready (backbone chains): the relation bias already sits in the saved block (one mmnas_rel_multi_fwd launch at chain entry
// computed it for every relation operator of the stream): skip the per-operator bias launch.
// kv_ready (backbone chains, guided operators): the key / value projections of every guided operator of the chain were
// issued as one grouped launch behind the encoder (they all read the final language state): only Q is projected here.
static int att_fwd_impl(const mmnas_att_op* op, void* stream, bool defer_ln, bool* ln_done, bool rel_ready = false, bool kv_ready = false);
}
extern "C" int mmnas_att_op_fwd(const mmnas_att_op* op, void* stream) {
  bool done;
  return mmnas::att_fwd_impl(op, stream, false, &done);
}
namespace mmnas {
// The forward of an attention operator in three stages, so that a supernet node can run the same stage of all its
// attention candidates as ONE launch (mixed chains): the Q / K / V projections as groups of one grouped product, the
// cores one by one, the merge projections (+ output dropout + residual) as groups of one product with per-group seeds.
static void att_qkv_groups(const mmnas_att_op* op, const AttLayout& L, mmnas_gemm_group* g) {
  memset(g, 0, 3 * sizeof(*g));
  g[0].M = (int)L.Mq; g[0].A[0] = op->xq;  g[0].B[0] = op->Wq; g[0].C = L.Q;
  g[1].M = (int)L.Mk; g[1].A[0] = op->xkv; g[1].B[0] = op->Wk; g[1].C = L.K;
  g[2].M = (int)L.Mk; g[2].A[0] = op->xkv; g[2].B[0] = op->Wv; g[2].C = L.V;
}

static void att_core_desc(const mmnas_att_op* op, const AttLayout& L, mmnas_mha_desc* m) {
  const int fl = op->flags, di = op->di;
  const bool drop = (fl & MMNAS_F_TRAIN) && op->drop_p > 0.f;
  memset(m, 0, sizeof(*m));
  m->B = op->B; m->H = op->H; m->Sq = op->Sq; m->Sk = op->Sk; m->dh = op->dh;
  m->ldq = m->ldk = m->ldv = m->ldo = di;
  m->Q = L.Q; m->K = L.K; m->V = L.V; m->mask = (fl & MMNAS_F_MASK) ? op->mask : nullptr; m->biasT = L.biasT;
  m->O = L.att; m->lse = L.stats;
  m->q_off = op->q_off; m->k_off = op->k_off;
  m->drop_p = drop ? op->drop_p : 0.f; m->drop_site = 0; m->drop_seed = op->seed;
}

// desc_only != NULL: the relation bias is computed, the core itself is left to the caller (its descriptor in *desc_only)
static int att_core_fwd(const mmnas_att_op* op, const AttLayout& L, void* stream, mmnas_mha_desc* desc_only = nullptr, bool rel_ready = false) {
  const int fl = op->flags;
  const bool rel = (fl & MMNAS_F_REL) && !rel_ready;
  int rc;
  if (rel) {
    if (fl & MMNAS_F_RELRAW) {  // lazy handle: bias straight from the raw [B,Sq,Sk,C] relations
      MMNAS_REQUIRE(op->Wy && op->by, MMNAS_E_ARG, "att_op_fwd: RELRAW without Wy/by");
      // Self-attention (every RelSelfAtt): the one-operator form of the kernel the backbone chains run for ALL relation
      // operators of a stream (relmulti.hip).  A row's bias does not depend on which other rows share its launch, so the
      // per-operator path and the chains produce the same bias BIT FOR BIT -- and with it the same logits.
      if (op->Sq == op->Sk && mmnas_rel_multi_supported(op->C, op->R, op->H) && (long)op->B * op->H * op->Sq * op->Sq < (1l << 31) &&
          !(getenv("MMNAS_REL_FWD_VALU") && getenv("MMNAS_REL_FWD_VALU")[0] == '1')) {
        mmnas_rel_multi q;
        memset(&q, 0, sizeof(q));
        q.B = op->B; q.S = op->Sq; q.C = op->C; q.R = op->R; q.H = op->H; q.n_ops = 1;
        q.raw = op->rel; q.Wy = op->Wy; q.by = op->by;
        q.Wr[0] = op->Wr; q.br[0] = op->br; q.biasT[0] = L.biasT;
        if (op->q_off) {
          MMNAS_REQUIRE(op->rel_tile_off, MMNAS_E_ARG, "att_op_fwd: packed rows without the relation tile offsets");
          q.off = op->q_off; q.tile_off = op->rel_tile_off; q.ntiles = op->rel_ntiles;
        }
        rc = mmnas_rel_multi_fwd(&q, stream);
      } else
      rc = op->q_off ? mmnas_rel_fused_fwd_ragged(op->rel, op->Wy, op->by, op->Wr, op->br, L.biasT, op->B, op->Sq, op->C, op->R,
                                                  op->H, op->q_off, stream)
                     : mmnas_rel_fused_fwd(op->rel, op->Wy, op->by, op->Wr, op->br, L.biasT, op->B, op->Sq, op->Sk, op->C, op->R,
                                           op->H, stream);
    } else {
      rc = mmnas_rel_bias_fwd(op->rel, op->Wr, op->br, L.biasT, op->B, op->Sq, op->Sk, op->R, op->H, stream);
    }
    if (rc) return rc;
  }
  if (desc_only) { att_core_desc(op, L, desc_only); return MMNAS_OK; }
  mmnas_mha_desc m;
  att_core_desc(op, L, &m);
  return mmnas_mha_core_fwd(&m, stream);
}

// the merge projection as one group: C = z (NORM: the LayerNorm follows) or y; residual = the operator's input
static void att_merge_group(const mmnas_att_op* op, const AttLayout& L, mmnas_gemm_group* g) {
  const bool norm = op->flags & MMNAS_F_NORM;
  memset(g, 0, sizeof(*g));
  g->M = (int)L.Mq; g->A[0] = L.att; g->B[0] = op->Wm; g->C = norm ? L.z : op->y;
  if (op->flags & MMNAS_F_RESIDUAL) g->residual = op->xq;
  g->drop_seed = op->seed;
}

static int att_fwd_args(const mmnas_att_op* op) {
  int rc = att_check(op, "att_op_fwd");
  if (rc) return rc;
  MMNAS_REQUIRE(op->xq && op->xkv && op->Wq && op->Wk && op->Wv && op->Wm && op->y && op->save, MMNAS_E_ARG,
                "att_op_fwd: null pointer");
  const int fl = op->flags;
  if (fl & MMNAS_F_NORM) MMNAS_REQUIRE(op->ln_a && op->ln_b, MMNAS_E_ARG, "att_op_fwd: NORM without ln parameters");
  if (fl & MMNAS_F_REL) MMNAS_REQUIRE(op->rel && op->Wr && op->br, MMNAS_E_ARG, "att_op_fwd: REL without rel/Wr/br");
  if (fl & MMNAS_F_MASK) MMNAS_REQUIRE(op->mask, MMNAS_E_ARG, "att_op_fwd: MASK without mask");
  return MMNAS_OK;
}
}  // namespace mmnas

static int mmnas::att_fwd_impl(const mmnas_att_op* op, void* stream, bool defer_ln, bool* ln_done, bool rel_ready, bool kv_ready) {
  *ln_done = true;
  int rc = att_fwd_args(op);
  if (rc) return rc;
  const int fl = op->flags;
  const bool norm = fl & MMNAS_F_NORM;
  const bool drop = (fl & MMNAS_F_TRAIN) && op->drop_p > 0.f;
  AttLayout L = att_layout(op);
  const int d = op->d, di = op->di;
  if (sa_small_applies(op)) return sa_small_fwd(op, L.Q, L.K, L.V, L.att, L.stats, L.z, (hipStream_t)stream);

  mmnas_gemm_desc g;
  gemm_init(g, MMNAS_GEMM_NT, di, d, d, d, di);
  g.ngroups = kv_ready ? 1 : 3;
  att_qkv_groups(op, L, g.g);
  if ((rc = mmnas_gemm(&g, stream))) return rc;
  if ((rc = att_core_fwd(op, L, stream, nullptr, rel_ready))) return rc;
  gemm_init(g, MMNAS_GEMM_NT, d, di, di, di, d);
  att_merge_group(op, L, &g.g[0]);
  if (fl & MMNAS_F_RESIDUAL) g.ldres = d;
  if (drop) { g.drop_p = op->drop_p; g.drop_site = 1; g.drop_seed = op->seed; }
  // d = 256: the projection, its dropout + residual epilogue and the LayerNorm as ONE launch (gemmln.hip)
  if (norm && !defer_ln && gemm_ln_applies(&g)) return gemm_ln(&g, op->ln_a, op->ln_b, op->y, d, op->eps, (hipStream_t)stream);
  if ((rc = mmnas_gemm(&g, stream))) return rc;

  if (norm && defer_ln) { *ln_done = false; return MMNAS_OK; }
  if (norm) return mmnas_layernorm_fwd(L.z, op->ln_a, op->ln_b, op->y, (int)L.Mq, d, op->eps, stream);
  return MMNAS_OK;
}

namespace mmnas {
// Parameter-gradient work set aside by the backward functions below when a queue is given: launches that nothing on
// the data-gradient chain waits for (weight-gradient products, LayerNorm parameter reductions, the relation-bias
// backward).  The caller decides when and on which stream they run (SideQueue::flush).
struct SideQueue {
  std::vector<std::function<int(hipStream_t)>> work;
  bool rel_only = false;   // queue only the relation-bias backward; the weight gradients stay paired on the main stream
  // everything `main` has been given so far finishes first; then the queued launches run on `side`
  int flush(hipStream_t main, hipStream_t side, hipEvent_t ev) {
    if (work.empty()) return MMNAS_OK;
    hipStream_t s = side;
    if (hipEventRecord(ev, main) != hipSuccess || hipStreamWaitEvent(side, ev, 0) != hipSuccess)
      s = main;   // (cannot order the streams: stay on the main one, which is always correct)
    for (auto& f : work) {
      const int rc = f(s);
      if (rc) return rc;
    }
    work.clear();
    return MMNAS_OK;
  }
};

// sq == nullptr: the single-stream order (data- and weight-gradient products paired in one launch).
// With a queue: `stream` carries only the chain the NEXT operator's backward waits for -- LayerNorm backward, the
// data-gradient products, the attention core; the weight-gradient products, the LayerNorm parameter reduction and the
// whole relation-bias backward (~40 % of an operator's backward time, read by nothing before the optimizer / the
// gradient exchange) are queued.
// acc_kv (guided operators inside the backbone chain): the key / value source's gradient is ADDED to *dxkv (the chain's
// running sum over the guided operators) instead of overwriting it -- saves the chain a buffer and an add launch each.
// rel_defer (backbone chains): the bias gradient dbiasT stays in the operator's scratch block; ONE mmnas_rel_multi_bwd launch
// behind the stream's last relation operator turns every operator's dbiasT into its dWr / dbr and the shared dWy / dby.
// kv_defer (backbone chains, guided operators): dK / dV stay in the scratch block; the chain turns every guided operator's
// pair into its key / value source gradient and dWk / dWv by grouped launches behind the last guided operator's backward.
// ln_pre (mixed chains, round 6): the LayerNorm backward already ran inside the node's mix kernel -- dz / dt of the scratch block
// are filled and *ln_pre is the pending parameter reduction; op->dy is not read.
static int att_bwd_impl(const mmnas_att_op* op, hipStream_t stream, SideQueue* sq, bool acc_kv = false, bool rel_defer = false, bool kv_defer = false,
                        const AuxReduce* ln_pre = nullptr) {
  const bool side = sq != nullptr && !sq->rel_only;
  const bool side_rel = sq != nullptr;
  int rc = att_check(op, "att_op_bwd");
  if (rc) return rc;
  MMNAS_REQUIRE(op->xq && op->xkv && op->Wq && op->Wk && op->Wv && op->Wm && op->save && op->ws && op->dy &&
                    op->dxq && op->dWq && op->dWk && op->dWv && op->dWm,
                MMNAS_E_ARG, "att_op_bwd: null pointer");
  const int fl = op->flags;
  const bool norm = fl & MMNAS_F_NORM, rel = fl & MMNAS_F_REL, self = fl & MMNAS_F_SELF;
  const bool drop = (fl & MMNAS_F_TRAIN) && op->drop_p > 0.f;
  if (!self) MMNAS_REQUIRE(op->dxkv, MMNAS_E_ARG, "att_op_bwd: dxkv required unless SELF");
  if (norm) MMNAS_REQUIRE(op->ln_a && op->dln_a && op->dln_b, MMNAS_E_ARG, "att_op_bwd: NORM gradients missing");
  if (rel) MMNAS_REQUIRE(op->rel && op->Wr && op->br && op->dWr && op->dbr, MMNAS_E_ARG, "att_op_bwd: REL gradients missing");
  AttLayout L = att_layout(op);
  const int d = op->d, di = op->di, Mq = (int)L.Mq, Mk = (int)L.Mk;

  // Short sequences (the language stream): steps 1, 2, 4 and the data-gradient half of 5 + 6 as ONE launch (small.hip), then
  // the four weight gradients as one grouped launch that carries the LayerNorm parameter reduction.
  if (!ln_pre && !sq && self && !rel && sa_small_bwd_applies(op)) {
    float* const dt_out = drop ? L.dt : (norm ? L.dz : nullptr);
    if ((rc = sa_small_bwd(op, L.Q, L.K, L.V, L.stats, L.z, dt_out, L.dQ, L.dK, L.dV, norm ? L.lnws : nullptr, stream))) return rc;
    mmnas_gemm_desc w4;
    gemm_init(w4, MMNAS_GEMM_TN, d, Mq, d, d, d);
    w4.ngroups = 4;
    w4.g[0].M = d; w4.g[0].A[0] = dt_out ? dt_out : op->dy; w4.g[0].B[0] = L.att; w4.g[0].C = op->dWm;
    w4.g[1].M = d; w4.g[1].A[0] = L.dQ; w4.g[1].B[0] = op->xq; w4.g[1].C = op->dWq;
    w4.g[2].M = d; w4.g[2].A[0] = L.dK; w4.g[2].B[0] = op->xq; w4.g[2].C = op->dWk;
    w4.g[3].M = d; w4.g[3].A[0] = L.dV; w4.g[3].B[0] = op->xq; w4.g[3].C = op->dWv;
    w4.accumulate = 1;
    AuxReduce red;
    red.part = norm ? L.lnws : nullptr; red.nrows = op->B; red.d = d;
    red.out[0] = op->dln_a; red.out[1] = op->dln_b; red.out[2] = nullptr;
    return gemm_wgrad_aux(&w4, &red, stream);
  }

  // 1. through LayerNorm and the output dropout
  const float* dz = op->dy;   // gradient wrt z = x + drop(core)
  const float* dt = op->dy;   // gradient wrt core
  AuxReduce lnred;   // the LayerNorm parameter-gradient reduction rides on the first gradient-pair launch below
  lnred.part = nullptr;
  if (norm && ln_pre) {
    lnred = *ln_pre;
    dz = L.dz; dt = drop ? L.dt : L.dz;
  } else if (norm) {
    if ((rc = layernorm_bwd_deferred(L.z, op->ln_a, op->dy, L.dz, op->dln_a, op->dln_b, drop ? L.dt : nullptr, nullptr,
                                     L.lnws, drop ? op->drop_p : 0.f, op->seed, 1, Mq, d, op->eps, (hipStream_t)stream, &lnred)))
      return rc;
    dz = L.dz; dt = drop ? L.dt : L.dz;
  } else if (drop) {
    if ((rc = mmnas_drop_add(op->dy, nullptr, L.dt, (size_t)Mq * d, op->drop_p, op->seed, 1, stream))) return rc;
    dt = L.dt;
  }

  mmnas_gemm_desc g, w, wm, w2;
  // 2. d(att) = dt Wm            [Mq,d] x [d,di]
  gemm_init(g, MMNAS_GEMM_NN, di, d, d, di, di);
  g.g[0].M = Mq; g.g[0].A[0] = dt; g.g[0].B[0] = op->Wm; g.g[0].C = L.datt;
  // 3. dWm += dt^T att           [d,di], reduction over the Mq rows (same launch: mmnas_gemm_pair)
  gemm_init(wm, MMNAS_GEMM_TN, di, Mq, d, di, di);
  wm.g[0].M = d; wm.g[0].A[0] = dt; wm.g[0].B[0] = L.att; wm.g[0].C = op->dWm;
  wm.accumulate = 1;
  if (side) { if ((rc = mmnas_gemm(&g, stream))) return rc; }
  else if ((rc = gemm_pair_aux(&g, &wm, &lnred, (hipStream_t)stream))) return rc;

  // 4. attention core backward
  mmnas_mha_desc m;
  memset(&m, 0, sizeof(m));
  m.B = op->B; m.H = op->H; m.Sq = op->Sq; m.Sk = op->Sk; m.dh = op->dh;
  m.ldq = m.ldk = m.ldv = m.ldo = di;
  m.Q = L.Q; m.K = L.K; m.V = L.V; m.mask = (fl & MMNAS_F_MASK) ? op->mask : nullptr; m.biasT = L.biasT;
  m.O = L.att; m.lse = L.stats;
  m.drop_p = drop ? op->drop_p : 0.f; m.drop_site = 0; m.drop_seed = op->seed;
  m.dO = L.datt; m.dQ = L.dQ; m.dK = L.dK; m.dV = L.dV; m.dbiasT = L.dbiasT; m.delta = L.delta;
  m.q_off = op->q_off; m.k_off = op->k_off;
  if ((rc = mmnas_mha_core_bwd(&m, stream))) return rc;

  // 5. + 6. projection weight gradients and input gradients (+ the residual branch dz), pairwise in one launch
  if (self) {
    gemm_init(w, MMNAS_GEMM_TN, d, Mq, di, d, d);
    w.ngroups = 3;
    w.g[0].M = di; w.g[0].A[0] = L.dQ; w.g[0].B[0] = op->xq;  w.g[0].C = op->dWq;
    w.g[1].M = di; w.g[1].A[0] = L.dK; w.g[1].B[0] = op->xkv; w.g[1].C = op->dWk;
    w.g[2].M = di; w.g[2].A[0] = L.dV; w.g[2].B[0] = op->xkv; w.g[2].C = op->dWv;
    w.accumulate = 1;
    gemm_init(g, MMNAS_GEMM_NN, d, di, di, d, d);
    g.nseg = 3;
    g.g[0].M = Mq; g.g[0].C = op->dxq;
    g.g[0].A[0] = L.dQ; g.g[0].B[0] = op->Wq;
    g.g[0].A[1] = L.dK; g.g[0].B[1] = op->Wk;
    g.g[0].A[2] = L.dV; g.g[0].B[2] = op->Wv;
    if (fl & MMNAS_F_RESIDUAL) { g.g[0].residual = dz; g.ldres = d; }
    if (side) {
      if ((rc = mmnas_gemm(&g, stream))) return rc;
    } else if ((rc = mmnas_gemm_pair(&g, &w, stream))) return rc;
  } else {
    gemm_init(w, MMNAS_GEMM_TN, d, Mq, di, d, d);
    w.g[0].M = di; w.g[0].A[0] = L.dQ; w.g[0].B[0] = op->xq; w.g[0].C = op->dWq;
    w.accumulate = 1;
    gemm_init(g, MMNAS_GEMM_NN, d, di, di, d, d);
    g.g[0].M = Mq; g.g[0].C = op->dxq; g.g[0].A[0] = L.dQ; g.g[0].B[0] = op->Wq;
    if (fl & MMNAS_F_RESIDUAL) { g.g[0].residual = dz; g.ldres = d; }
    if (side) {
      if ((rc = mmnas_gemm(&g, stream))) return rc;
    } else if ((rc = mmnas_gemm_pair(&g, &w, stream))) return rc;
    if (kv_defer) goto params;
    gemm_init(w2, MMNAS_GEMM_TN, d, Mk, di, d, d);
    w2.ngroups = 2;
    w2.g[0].M = di; w2.g[0].A[0] = L.dK; w2.g[0].B[0] = op->xkv; w2.g[0].C = op->dWk;
    w2.g[1].M = di; w2.g[1].A[0] = L.dV; w2.g[1].B[0] = op->xkv; w2.g[1].C = op->dWv;
    w2.accumulate = 1;
    gemm_init(g, MMNAS_GEMM_NN, d, di, di, d, d);
    g.nseg = 2;
    g.g[0].M = Mk; g.g[0].C = op->dxkv;
    g.g[0].A[0] = L.dK; g.g[0].B[0] = op->Wk;
    g.g[0].A[1] = L.dV; g.g[0].B[1] = op->Wv;
    if (acc_kv) g.accumulate = 1;
    if (side) {
      if ((rc = mmnas_gemm(&g, stream))) return rc;
    } else if ((rc = mmnas_gemm_pair(&g, &w2, stream))) return rc;
  }

params:
  // Parameter-gradient work.  Single stream: the weight gradients went out paired above, the relation-bias backward
  // follows.  Side stream: all of it goes there, behind ONE event recorded at this point of `stream`.
  if (side) {
    sq->work.push_back([lnred](hipStream_t s) { return launch_aux_reduce(lnred, s); });
    sq->work.push_back([wm](hipStream_t s) { return mmnas_gemm(&wm, s); });
    sq->work.push_back([w](hipStream_t s) { return mmnas_gemm(&w, s); });
    if (!self) sq->work.push_back([w2](hipStream_t s) { return mmnas_gemm(&w2, s); });
  }
  // 7. relation bias
  if (rel && rel_defer) return MMNAS_OK;
  if (rel && (fl & MMNAS_F_RELRAW)) {
    MMNAS_REQUIRE(op->Wy && op->by && op->dWy && op->dby, MMNAS_E_ARG, "att_op_bwd: RELRAW gradients missing");
    const mmnas_att_op o = *op;
    float* const dbiasT = L.dbiasT;
    float* const relws = L.relws;
    auto relb = [o, dbiasT, relws](hipStream_t s) {
      if (o.q_off)
        return mmnas_rel_fused_bwd_ragged(o.rel, o.Wy, o.by, o.Wr, o.br, dbiasT, o.dWy, o.dby, o.dWr, o.dbr, relws, o.B, o.Sq,
                                          o.C, o.R, o.H, o.q_off, o.rel_tile_off, o.rel_ntiles, s);
      return mmnas_rel_fused_bwd(o.rel, o.Wy, o.by, o.Wr, o.br, dbiasT, o.dWy, o.dby, o.dWr, o.dbr, relws, o.B, o.Sq, o.Sk,
                                 o.C, o.R, o.H, s);
    };
    if (side_rel) { sq->work.push_back(relb); return MMNAS_OK; }
    return relb(stream);
  }
  if (rel)   // (a materialised relation tensor's own gradient may feed the caller: main stream)
    return mmnas_rel_bias_bwd(op->rel, op->Wr, op->br, L.dbiasT, op->drel, op->dWr, op->dbr, 0, op->B, op->Sq,
                              op->Sk, op->R, op->H, stream);
  return MMNAS_OK;
}
}  // namespace mmnas

extern "C" int mmnas_att_op_bwd(const mmnas_att_op* op, void* stream) {
  return att_bwd_impl(op, (hipStream_t)stream, nullptr);
}

// ------------------------------------------------------------------------------------------ MLP
namespace mmnas {

struct MlpLayout {
  float* h[3];      // h[i] = input of layer i (h[0] = x, not stored); saved for i >= 1
  float* z;
  float *dz, *dt, *dp[3], *lnws;
  size_t save_bytes, ws_bwd_bytes;
};

static MlpLayout mlp_layout(const mmnas_mlp_op* op) {
  MlpLayout L;
  const bool norm = op->flags & MMNAS_F_NORM;
  const bool drop = (op->flags & MMNAS_F_TRAIN) && op->drop_p > 0.f;
  const size_t M = (size_t)op->M;
  Carver s(op->save);
  L.h[0] = nullptr;
  for (int i = 1; i < 3; ++i) L.h[i] = (i < op->nl) ? s.take(M * op->dims[i]) : nullptr;
  L.z = norm ? s.take(M * op->dims[0]) : nullptr;
  L.save_bytes = s.off;
  size_t maxh = 4;
  for (int i = 1; i < op->nl; ++i) if ((size_t)op->dims[i] > maxh) maxh = op->dims[i];
  Carver w(op->ws);
  L.dz = norm ? w.take(M * op->dims[0]) : nullptr;
  L.dt = drop ? w.take(M * op->dims[0]) : nullptr;
  L.dp[0] = w.take(M * maxh);
  L.dp[1] = op->nl > 2 ? w.take(M * maxh) : nullptr;
  L.dp[2] = nullptr;
  L.lnws = norm ? w.take(mmnas_layernorm_bwd_ws_floats(op->M, op->dims[0])) : nullptr;
  L.ws_bwd_bytes = w.off;
  return L;
}

static int mlp_check(const mmnas_mlp_op* op, const char* who) {
  MMNAS_REQUIRE(op, MMNAS_E_ARG, "%s: null descriptor", who);
  MMNAS_REQUIRE(op->nl >= 1 && op->nl <= 3 && op->M > 0, MMNAS_E_SHAPE, "%s: nl=%d M=%d", who, op->nl, op->M);
  MMNAS_REQUIRE(op->dims[0] == op->dims[op->nl], MMNAS_E_SHAPE, "%s: in/out width differ (%d vs %d)", who,
                op->dims[0], op->dims[op->nl]);
  for (int i = 0; i <= op->nl; ++i)
    MMNAS_REQUIRE(op->dims[i] > 0 && op->dims[i] % 4 == 0, MMNAS_E_SHAPE, "%s: dims[%d]=%d", who, i, op->dims[i]);
  return MMNAS_OK;
}

}  // namespace mmnas

extern "C" int mmnas_mlp_op_plan(const mmnas_mlp_op* op, mmnas_plan* plan) {
  int rc = mlp_check(op, "mlp_op_plan");
  if (rc) return rc;
  MMNAS_REQUIRE(plan, MMNAS_E_ARG, "mlp_op_plan: null plan");
  mmnas_mlp_op tmp = *op;
  tmp.save = nullptr; tmp.ws = nullptr;
  MlpLayout L = mlp_layout(&tmp);
  plan->save_bytes = L.save_bytes ? L.save_bytes : 256;
  plan->ws_fwd_bytes = 256;
  plan->ws_bwd_bytes = L.ws_bwd_bytes;
  return MMNAS_OK;
}

namespace mmnas {
static int mlp_fwd_impl(const mmnas_mlp_op* op, void* stream, bool defer_ln, bool* ln_done);
}
extern "C" int mmnas_mlp_op_fwd(const mmnas_mlp_op* op, void* stream) {
  bool done;
  return mmnas::mlp_fwd_impl(op, stream, false, &done);
}
static int mmnas::mlp_fwd_impl(const mmnas_mlp_op* op, void* stream, bool defer_ln, bool* ln_done) {
  *ln_done = true;
  int rc = mlp_check(op, "mlp_op_fwd");
  if (rc) return rc;
  MMNAS_REQUIRE(op->x && op->y && op->save, MMNAS_E_ARG, "mlp_op_fwd: null pointer");
  const int fl = op->flags;
  const bool norm = fl & MMNAS_F_NORM;
  const bool drop = (fl & MMNAS_F_TRAIN) && op->drop_p > 0.f;
  if (norm) MMNAS_REQUIRE(op->ln_a && op->ln_b, MMNAS_E_ARG, "mlp_op_fwd: NORM without ln parameters");
  MlpLayout L = mlp_layout(op);
  const int d = op->dims[0];
  // short row counts (the language stream): both layers, the residual and the LayerNorm as ONE launch (small.hip); a deferred
  // LayerNorm (architecture-step nodes) keeps the general path
  if (!(norm && defer_ln) && ffn_small_applies(op)) return ffn_small_fwd(op, L.h[1], L.z, (hipStream_t)stream);
  const float* in = op->x;
  mmnas_gemm_desc g;
  for (int i = 0; i < op->nl; ++i) {
    MMNAS_REQUIRE(op->W[i], MMNAS_E_ARG, "mlp_op_fwd: W[%d] null", i);
    const bool last = i == op->nl - 1;
    gemm_init(g, MMNAS_GEMM_NT, op->dims[i + 1], op->dims[i], op->dims[i], op->dims[i], op->dims[i + 1]);
    g.g[0].M = op->M; g.g[0].A[0] = in; g.g[0].B[0] = op->W[i]; g.g[0].bias = op->b[i];
    if (drop) { g.drop_p = op->drop_p; g.drop_seed = op->seed; g.drop_site = last ? 1u : (i == 0 ? 0u : 2u); }
    if (last) {
      g.g[0].C = norm ? L.z : op->y;
      if (fl & MMNAS_F_RESIDUAL) { g.g[0].residual = op->x; g.ldres = d; }
    } else {
      g.relu = 1;
      g.g[0].C = L.h[i + 1];
    }
    if (last && norm && !defer_ln && gemm_ln_applies(&g)) return gemm_ln(&g, op->ln_a, op->ln_b, op->y, d, op->eps, (hipStream_t)stream);
    if ((rc = mmnas_gemm(&g, stream))) return rc;
    in = L.h[i + 1];
  }
  if (norm && defer_ln) { *ln_done = false; return MMNAS_OK; }
  if (norm) return mmnas_layernorm_fwd(L.z, op->ln_a, op->ln_b, op->y, op->M, d, op->eps, stream);
  return MMNAS_OK;
}

namespace mmnas {
static int mlp_bwd_impl(const mmnas_mlp_op* op, hipStream_t stream, SideQueue* sq, const AuxReduce* ln_pre = nullptr) {
  const bool side = sq != nullptr && !sq->rel_only;
  int rc = mlp_check(op, "mlp_op_bwd");
  if (rc) return rc;
  MMNAS_REQUIRE(op->x && op->save && op->ws && op->dy && op->dx, MMNAS_E_ARG, "mlp_op_bwd: null pointer");
  const int fl = op->flags;
  const bool norm = fl & MMNAS_F_NORM;
  const bool drop = (fl & MMNAS_F_TRAIN) && op->drop_p > 0.f;
  MlpLayout L = mlp_layout(op);
  const int d = op->dims[0], M = op->M, nl = op->nl;
  for (int i = 0; i < nl; ++i)
    MMNAS_REQUIRE(op->W[i] && op->dW[i], MMNAS_E_ARG, "mlp_op_bwd: W/dW[%d] null", i);

  const float* dz = op->dy;
  const float* dt = op->dy;
  bool last_bias_done = false;
  AuxReduce lnred;   // pending LayerNorm reduction: rides on the first gradient-pair launch
  lnred.part = nullptr;
  if (norm) {
    MMNAS_REQUIRE(op->ln_a && op->dln_a && op->dln_b, MMNAS_E_ARG, "mlp_op_bwd: NORM gradients missing");
    // the column sums of the dropped gradient are the last layer's bias gradient: fused when a
    // separate dt buffer exists
    float* dcol = (drop && op->db[nl - 1]) ? op->db[nl - 1] : nullptr;
    if (ln_pre) lnred = *ln_pre;     // (mixed chains: done inside the node's mix kernel, dcol among its partial rows)
    else if ((rc = layernorm_bwd_deferred(L.z, op->ln_a, op->dy, L.dz, op->dln_a, op->dln_b, drop ? L.dt : nullptr, dcol,
                                          L.lnws, drop ? op->drop_p : 0.f, op->seed, 1, M, d, op->eps, (hipStream_t)stream, &lnred)))
      return rc;
    dz = L.dz; dt = drop ? L.dt : L.dz;
    last_bias_done = dcol != nullptr;
  } else if (drop) {
    if ((rc = mmnas_drop_add(op->dy, nullptr, L.dt, (size_t)M * d, op->drop_p, op->seed, 1, stream))) return rc;
    dt = L.dt;
  }

  const float gate_scale = drop ? 1.0f / (1.0f - op->drop_p) : 1.0f;
  const float* dpre = dt;  // gradient wrt the pre-activation output of layer i
  mmnas_gemm_desc g;
  mmnas_gemm_desc wd[3];   // the weight-gradient products (side-stream mode: issued after the data-gradient chain)
  const bool last_colsum = op->db[nl - 1] && !last_bias_done;
  if (last_colsum && !side)
    if ((rc = mmnas_colsum(dpre, op->db[nl - 1], M, op->dims[nl], op->dims[nl], stream))) return rc;
  for (int i = nl - 1; i >= 0; --i) {
    const float* hin = i == 0 ? op->x : L.h[i];
    const int nout = op->dims[i + 1], nin = op->dims[i];
    // bias gradient: column sums of dpre.  The last layer's ride on the LayerNorm backward when possible (else the
    // colsum launch above / on the side stream); a hidden layer's are accumulated by the epilogue of the data-gradient
    // product that writes its dpre (below)
    // weight gradient dW_i[nout,nin] += dpre^T hin and data gradient, one launch (or: the side stream)
    mmnas_gemm_desc& w = wd[i];
    gemm_init(w, MMNAS_GEMM_TN, nin, M, nout, nin, nin);
    w.g[0].M = nout; w.g[0].A[0] = dpre; w.g[0].B[0] = hin; w.g[0].C = op->dW[i];
    w.accumulate = 1;
    gemm_init(g, MMNAS_GEMM_NN, nin, nout, nout, nin, nin);
    g.g[0].M = M; g.g[0].A[0] = dpre; g.g[0].B[0] = op->W[i];
    if (i == 0) {
      g.g[0].C = op->dx;
      if (fl & MMNAS_F_RESIDUAL) { g.g[0].residual = dz; g.ldres = d; }
    } else {
      float* out = L.dp[(nl - 1 - i) % 3];   // (side-stream mode reads every dpre later: no buffer is reused)
      g.g[0].C = out;
      g.g[0].gate = L.h[i]; g.ldgate = nin; g.gate_scale = gate_scale;  // relu' and dropout replay from h_i
      g.g[0].colsum = op->db[i - 1];   // db_{i-1} += column sums of dpre_{i-1} (may be NULL: layer without bias)
    }
    if (side) {
      if ((rc = mmnas_gemm(&g, stream))) return rc;
    } else if ((rc = gemm_pair_aux(&g, &w, i == nl - 1 ? &lnred : nullptr, (hipStream_t)stream))) return rc;
    if (i > 0) dpre = g.g[0].C;
  }
  if (side) {
    sq->work.push_back([lnred](hipStream_t s) { return launch_aux_reduce(lnred, s); });
    if (last_colsum) {
      float* const db = op->db[nl - 1];
      const int n = op->dims[nl];
      sq->work.push_back([dt, db, M, n](hipStream_t s) { return mmnas_colsum(dt, db, M, n, n, s); });
    }
    for (int i = nl - 1; i >= 0; --i) {
      const mmnas_gemm_desc w = wd[i];
      sq->work.push_back([w](hipStream_t s) { return mmnas_gemm(&w, s); });
    }
  }
  return MMNAS_OK;
}
}  // namespace mmnas

extern "C" int mmnas_mlp_op_bwd(const mmnas_mlp_op* op, void* stream) {
  return mlp_bwd_impl(op, (hipStream_t)stream, nullptr);
}

// ------------------------------------------------------------------------------------------ backbone chain
// One call = every cell operator of a backbone (Backbone_*.forward, hygr_vqa.py:45-52 / full_vqa.py:46-53) in
// evaluation order: the host issues O(1) calls per step instead of one autograd node, six allocations and a descriptor
// per operator and direction.  Intermediate activations, saved blocks and backward scratch live in ONE caller-owned
// arena laid out by mmnas_chain_plan.
#include <map>
#include <mutex>
#include <utility>
namespace mmnas {

struct ChainLayout {
  size_t y[MMNAS_CHAIN_MAX_OPS], save[MMNAS_CHAIN_MAX_OPS], ws[MMNAS_CHAIN_MAX_OPS], dx[MMNAS_CHAIN_MAX_OPS],
      tmp[MMNAS_CHAIN_MAX_OPS];
  size_t dpre, encdy, relws, total;
  int last_x, last_y, first_x, first_y, n_guided;
  // mixed chains: per node (indexed by the node's first operator) the node output and the sampled candidate's output
  // gradient; one scratch block for the gate-gradient partials
  size_t nout[MMNAS_CHAIN_MAX_OPS], ndact[MMNAS_CHAIN_MAX_OPS], mixws;
};

static int chain_check(const mmnas_chain* c, const char* who) {
  MMNAS_REQUIRE(c && c->ops, MMNAS_E_ARG, "%s: null chain", who);
  MMNAS_REQUIRE(c->n_ops >= 1 && c->n_ops <= MMNAS_CHAIN_MAX_OPS, MMNAS_E_SHAPE, "%s: %d operators (1..%d)", who, c->n_ops, MMNAS_CHAIN_MAX_OPS);
  MMNAS_REQUIRE(c->B > 0 && c->Sx > 0 && c->Sy > 0 && c->d > 0, MMNAS_E_SHAPE, "%s: B=%d Sx=%d Sy=%d d=%d", who, c->B, c->Sx, c->Sy, c->d);
  if (c->y_off) MMNAS_REQUIRE(c->Ny > 0 && c->Ny <= c->B * c->Sy && c->Sy <= 128 && !c->use_side_stream, MMNAS_E_ARG,
                              "%s: ragged decoder stream: Ny=%d of B*Sy=%d rows, Sy=%d <= 128, single stream", who, c->Ny, c->B * c->Sy, c->Sy);
  bool seen_y = false;
  for (int i = 0; i < c->n_ops; ++i) {
    const mmnas_chain_op& o = c->ops[i];
    MMNAS_REQUIRE(o.kind == MMNAS_CHAIN_ATT || o.kind == MMNAS_CHAIN_MLP, MMNAS_E_ARG, "%s: operator %d: kind %d", who, i, o.kind);
    if (o.on_y) seen_y = true;
    else MMNAS_REQUIRE(!seen_y, MMNAS_E_ARG, "%s: encoder operators must precede the decoder's (operator %d)", who, i);
    if (o.kind == MMNAS_CHAIN_ATT && !(o.att.flags & MMNAS_F_SELF))
      MMNAS_REQUIRE(o.on_y, MMNAS_E_ARG, "%s: operator %d: guided attention needs the decoder stream", who, i);
  }
  if (c->mixed) {
    MMNAS_REQUIRE(c->gate && c->gate_width >= 1 && c->gate_width <= MMNAS_MIXED_MAX, MMNAS_E_ARG, "%s: mixed chain without a gate block (width %d)", who, c->gate_width);
    MMNAS_REQUIRE(!c->use_side_stream, MMNAS_E_ARG, "%s: mixed chains run on one stream", who);
    MMNAS_REQUIRE(c->d <= 1024, MMNAS_E_SHAPE, "%s: mixed chains need d <= 1024", who);
    int node = -1, live = 0, count = 0;
    for (int i = 0; i < c->n_ops; ++i) {
      const mmnas_chain_op& o = c->ops[i];
      if (o.node != node) {
        MMNAS_REQUIRE(o.node == node + 1, MMNAS_E_ARG, "%s: operator %d: node %d after node %d", who, i, o.node, node);
        MMNAS_REQUIRE(node < 0 || live == 1, MMNAS_E_ARG, "%s: node %d has %d differentiated candidates (exactly one)", who, node, live);
        node = o.node; live = 0; count = 0;
      } else {
        MMNAS_REQUIRE(o.on_y == c->ops[i - 1].on_y, MMNAS_E_ARG, "%s: node %d mixes the two streams", who, node);
      }
      MMNAS_REQUIRE(o.cand >= 0 && o.cand < c->gate_width, MMNAS_E_ARG, "%s: operator %d: candidate %d outside the gate row", who, i, o.cand);
      MMNAS_REQUIRE(++count <= MMNAS_MIXED_MAX, MMNAS_E_SHAPE, "%s: node %d has more than %d candidates", who, node, MMNAS_MIXED_MAX);
      live += o.detached ? 0 : 1;
    }
    MMNAS_REQUIRE(live == 1, MMNAS_E_ARG, "%s: node %d has %d differentiated candidates (exactly one)", who, node, live);
  }
  return MMNAS_OK;
}

// rows of the decoder stream: B * Sy, or the packed row count of a ragged batch
static inline size_t chain_rows_y(const mmnas_chain* c) { return c->y_off ? (size_t)c->Ny : (size_t)c->B * c->Sy; }

// operator i with its stream-dependent fields filled in (shapes, masks, relation tensors); buffers come later
static void chain_op_setup(const mmnas_chain* c, int i, mmnas_att_op& a, mmnas_mlp_op& m) {
  const mmnas_chain_op& o = c->ops[i];
  const int S = o.on_y ? c->Sy : c->Sx;
  if (o.kind == MMNAS_CHAIN_ATT) {
    a = o.att;
    a.B = c->B; a.d = c->d; a.Sq = S;
    const bool self = a.flags & MMNAS_F_SELF;
    a.Sk = self ? S : c->Sx;
    const uint8_t* mask = (self && o.on_y) ? c->y_mask : c->x_mask;
    a.q_off = a.k_off = nullptr; a.Mq = a.Mk = 0; a.rel_tile_off = nullptr; a.rel_ntiles = 0;
    if (o.on_y && c->y_off) {   // ragged decoder stream: packed queries; self-attention over the sample's own packed rows
      a.q_off = c->y_off; a.Mq = c->Ny;
      if (self) { a.k_off = c->y_off; a.Mk = c->Ny; mask = nullptr; a.rel_tile_off = c->y_tile_off; a.rel_ntiles = c->y_ntiles; }
    }
    a.mask = mask;
    if (mask) a.flags |= MMNAS_F_MASK; else a.flags &= ~MMNAS_F_MASK;
    if (a.flags & MMNAS_F_REL) a.rel = o.on_y ? c->y_rel : c->x_rel;
  } else {
    m = o.mlp;
    m.M = (o.on_y && c->y_off) ? c->Ny : c->B * S;
  }
}

static int chain_layout(const mmnas_chain* c, ChainLayout& L) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t r = off; off += al(bytes); return r; };
  L.last_x = L.last_y = L.first_x = L.first_y = -1;
  L.n_guided = 0;
  for (int i = 0; i < c->n_ops; ++i) {
    if (c->ops[i].on_y) { if (L.first_y < 0) L.first_y = i; L.last_y = i; }
    else { if (L.first_x < 0) L.first_x = i; L.last_x = i; }
  }
  const size_t nx = (size_t)c->B * c->Sx * c->d * sizeof(float), ny = chain_rows_y(c) * c->d * sizeof(float);
  for (int i = 0; i < c->n_ops; ++i) {
    const mmnas_chain_op& o = c->ops[i];
    mmnas_att_op a; mmnas_mlp_op m;
    chain_op_setup(c, i, a, m);
    mmnas_plan pl;
    int rc = o.kind == MMNAS_CHAIN_ATT ? mmnas_att_op_plan(&a, &pl) : mmnas_mlp_op_plan(&m, &pl);
    if (rc) return rc;
    const size_t n = o.on_y ? ny : nx;
    L.y[i] = take(n);
    L.save[i] = take(pl.save_bytes);
    L.ws[i] = take(pl.ws_bwd_bytes);
    L.dx[i] = take(n);
    const bool guided = o.kind == MMNAS_CHAIN_ATT && !(a.flags & MMNAS_F_SELF) && !(c->mixed && o.detached);
    L.tmp[i] = guided ? take(nx) : 0;   // (hoisted key / value source gradient of a guided operator: summed once, chain_guided_kv_bwd)
    L.n_guided += guided;
  }
  L.dpre = take(nx);
  L.encdy = take(nx);
  L.relws = take(mmnas_rel_multi_bwd_ws_floats(c->B, c->Sx > c->Sy ? c->Sx : c->Sy) * sizeof(float));
  L.mixws = 0;
  if (c->mixed) {
    for (int i = 0; i < c->n_ops; ++i) {
      const bool first = i == 0 || c->ops[i].node != c->ops[i - 1].node;
      L.nout[i] = first ? take(c->ops[i].on_y ? ny : nx) : 0;
      L.ndact[i] = first ? take(c->ops[i].on_y ? ny : nx) : 0;
    }
    {   // one partial-sum area per node: the gate-gradient reductions of the backward are issued as ONE launch at its end
      int nodes = 0;
      for (int i = 0; i < c->n_ops; ++i) nodes += (i == 0 || c->ops[i].node != c->ops[i - 1].node);
      L.mixws = take((size_t)(nodes > 0 ? nodes : 1) * mmnas_mixed_sum_ws_floats() * sizeof(float));
    }
    // (streams' first / last markers refer to NODES here: the first operator of the first / last node of a stream)
    L.last_x = L.last_y = L.first_x = L.first_y = -1;
    for (int i = 0; i < c->n_ops; ++i) {
      if (!(i == 0 || c->ops[i].node != c->ops[i - 1].node)) continue;
      if (c->ops[i].on_y) { if (L.first_y < 0) L.first_y = i; L.last_y = i; }
      else { if (L.first_x < 0) L.first_x = i; L.last_x = i; }
    }
  }
  L.total = off;
  return MMNAS_OK;
}

// ---- side stream + events, one set per (device, main stream) ----
struct SideCtx { hipStream_t side; hipStream_t enc; hipEvent_t ev[MMNAS_CHAIN_MAX_OPS + 2]; hipEvent_t ovl[4]; bool used; };
static std::mutex g_side_mu;
static std::map<std::pair<int, hipStream_t>, SideCtx> g_side;

static SideCtx* side_ctx(hipStream_t main, bool create) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_side_mu);
  auto key = std::make_pair(dev, main);
  auto it = g_side.find(key);
  if (it == g_side.end()) {
    if (!create) return nullptr;
    SideCtx s;
    memset(&s, 0, sizeof(s));
    // lowest priority: when both streams have workgroups to dispatch, the data-gradient chain goes first
    int least = 0, greatest = 0;
    const bool prio = !(getenv("MMNAS_SIDE_PRIO") && getenv("MMNAS_SIDE_PRIO")[0] == '0') &&
                      hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest;
    if ((prio ? hipStreamCreateWithPriority(&s.side, hipStreamNonBlocking, least)
              : hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking)) != hipSuccess) return nullptr;
    for (auto& e : s.ev)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    // the language stream's own stream (chain overlap): highest priority -- its launches are a handful of workgroups on
    // a latency-bound chain and must not queue behind the image stream's thousand-workgroup launches
    if ((prio ? hipStreamCreateWithPriority(&s.enc, hipStreamNonBlocking, greatest)
              : hipStreamCreateWithFlags(&s.enc, hipStreamNonBlocking)) != hipSuccess) return nullptr;
    for (auto& e : s.ovl)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    it = g_side.emplace(key, s).first;
  }
  return &it->second;
}

// Encoder / decoder overlap.  The language stream's operators run on 896 rows: 56-224 workgroups per launch, ~10 us
// each whatever they compute, on a quarter of the CUs.  The image-stream operators in front of the FIRST guided
// operator do not read the language state, so forward they run beside the encoder, and backward the encoder's
// operators run beside theirs (every guided operator's gradient has reached the language state by then).
// MEASURED: no gain (5.746 vs 5.738 ms per supernet step, 12.15 vs 12.15 ms per training step).  A language-stream launch
// is resident on every CU for its whole latency-bound duration, and its LDS / wave slots cost the image stream's
// GEMM one of its three workgroups per CU meanwhile: in the trace the co-running kernels take 1.3-1.5x their solo time
// (GEMM 26.6 -> 39.6 us, attention 22 -> 32 us), which returns what the overlap saves.  Off by default
// (MMNAS_CHAIN_OVERLAP=1 enables it); kept because the fork / join structure is what a multi-stream caller needs.
static int g_chain_overlap = -1;   // -1: not read yet (MMNAS_CHAIN_OVERLAP, default 0); mmnas_set_chain_overlap() overrides
static bool chain_overlap_on() {
  if (g_chain_overlap < 0) { const char* e = getenv("MMNAS_CHAIN_OVERLAP"); g_chain_overlap = (e && e[0] ? atoi(e) : 0) ? 1 : 0; }
  return g_chain_overlap != 0;
}
static bool head_overlap_on() {
  static const int on = [] { const char* e = getenv("MMNAS_HEAD_OVERLAP"); return e && e[0] ? atoi(e) : 0; }();
  return on != 0;
}
static bool head_glimpse1_on() {   // MMNAS_HEAD_GLIMPSE1=0: the one-glimpse logit layer as GEMM launches (A/B, tests)
  const char* e = getenv("MMNAS_HEAD_GLIMPSE1");
  return !(e && e[0] == '0');
}
static bool head_projT_on() {   // MMNAS_HEAD_PROJT=0: the answer projection's gradients from the untransposed loss gradient
  const char* e = getenv("MMNAS_HEAD_PROJT");
  return !(e && e[0] == '0');
}
static int first_guided(const mmnas_chain* c) {
  for (int i = 0; i < c->n_ops; ++i)
    if (c->ops[i].kind == MMNAS_CHAIN_ATT && !(c->ops[i].att.flags & MMNAS_F_SELF)) return i;
  return c->n_ops;
}
static int ev_fork(hipStream_t from, hipStream_t to, hipEvent_t e) {
  if (hipEventRecord(e, from) != hipSuccess || hipStreamWaitEvent(to, e, 0) != hipSuccess) {
    set_error("chain: event record / wait failed");
    return MMNAS_E_LAUNCH;
  }
  return MMNAS_OK;
}


// ---- relation bias of all lazy-handle relation operators of a stream in one launch per direction (relmulti.hip) ----
static int g_rel_hoist = -1;   // -1: not read yet (MMNAS_REL_HOIST, default 1); mmnas_set_rel_hoist() overrides
static bool rel_hoist_on() {
  if (g_rel_hoist < 0) { const char* e = getenv("MMNAS_REL_HOIST"); g_rel_hoist = (e && e[0] ? atoi(e) : 1) ? 1 : 0; }
  return g_rel_hoist != 0;
}
struct RelGroups {
  std::vector<mmnas_rel_multi> groups[2];      // [0]: the language stream's, [1]: the image stream's
  bool hoisted[MMNAS_CHAIN_MAX_OPS];
  RelGroups() { memset(hoisted, 0, sizeof(hoisted)); }
};
// Which operators take part: attention operators with MMNAS_F_REL | MMNAS_F_RELRAW whose shape relmulti.hip covers; backward
// (bwd = true) only the differentiated ones.  Operators of one stream that share (raw, Wy, by, H, C) form a group.
static void chain_rel_groups(const mmnas_chain* c, const ChainLayout& L, bool bwd, RelGroups& G) {
  if (!rel_hoist_on() || c->use_side_stream) return;
  char* base = (char*)c->arena;
  for (int i = 0; i < c->n_ops; ++i) {
    const mmnas_chain_op& o = c->ops[i];
    if (o.kind != MMNAS_CHAIN_ATT) continue;
    if (bwd && c->mixed && o.detached) continue;
    mmnas_att_op a; mmnas_mlp_op m;
    chain_op_setup(c, i, a, m);
    if (!(a.flags & MMNAS_F_REL) || !(a.flags & MMNAS_F_RELRAW) || !(a.flags & MMNAS_F_SELF)) continue;
    if (!mmnas_rel_multi_supported(a.C, a.R, a.H) || !a.rel || !a.Wy || !a.by) continue;
    if ((long)a.B * a.H * a.Sq * a.Sq >= (1l << 31)) continue;
    a.save = base + L.save[i]; a.ws = base + L.ws[i];
    const AttLayout al_ = att_layout(&a);
    std::vector<mmnas_rel_multi>& gs = G.groups[o.on_y ? 1 : 0];
    mmnas_rel_multi* g = nullptr;
    for (auto& q : gs)
      if (q.raw == a.rel && q.Wy == a.Wy && q.by == a.by && q.H == a.H && q.C == a.C && q.n_ops < MMNAS_REL_MULTI_MAX) { g = &q; break; }
    if (!g) {
      mmnas_rel_multi q;
      memset(&q, 0, sizeof(q));
      q.B = a.B; q.S = a.Sq; q.C = a.C; q.R = a.R; q.H = a.H;
      q.raw = a.rel; q.Wy = a.Wy; q.by = a.by; q.dWy = a.dWy; q.dby = a.dby;
      q.off = a.q_off; q.tile_off = a.rel_tile_off; q.ntiles = a.rel_ntiles;
      q.ws = (float*)(base + L.relws);
      gs.push_back(q);
      g = &gs.back();
    }
    const int j = g->n_ops++;
    g->Wr[j] = a.Wr; g->br[j] = a.br; g->biasT[j] = al_.biasT; g->dbiasT[j] = al_.dbiasT; g->dWr[j] = a.dWr; g->dbr[j] = a.dbr;
    G.hoisted[i] = true;
  }
}
static int chain_rel_fwd(const RelGroups& G, hipStream_t st, int only = -1) {
  for (int s = 0; s < 2; ++s)
    if (only < 0 || only == s)
      for (const auto& g : G.groups[s]) { const int rc = mmnas_rel_multi_fwd(&g, st); if (rc) return rc; }
  return MMNAS_OK;
}
// The image stream's relation launches BESIDE the language stream's operators (round 5, MMNAS_REL_OVERLAP=1; OFF by default:
// MEASURED SLOWER -- supernet step 4.70 -> 4.81-4.88 ms, arch step 8.02 -> 8.21-8.25 ms, three alternations on one box: the
// fourth form of stream overlap tried on this path and the fourth that lost to the cross-stream event waits and to the
// co-running kernels slowing each other; kept because the fork / join structure is tested and documents the negative).  The
// encoder is a dependent chain of ~40 launches of 56-224 workgroups on 896 rows -- 0.25 ms forward, 0.35 ms backward during
// which most of the chip idles -- and the image stream's relation bias depends on nothing it computes: forward it is issued
// on the side stream at chain entry and joined in front of the first relation operator of the decoder; backward it is issued
// on the side stream behind the decoder's first operator (every bias gradient exists) and joined at the end of the call,
// while the encoder's backward runs on the caller's stream.  A bucket mark that covers the relation parameters is recorded
// on the side stream behind the launch (the side stream waited for the caller's stream first: the event covers both).
static int g_rel_overlap = -1;
static bool rel_overlap_on() {
  if (g_rel_overlap < 0) { const char* e = getenv("MMNAS_REL_OVERLAP"); g_rel_overlap = (e && e[0] ? atoi(e) : 0) ? 1 : 0; }
  return g_rel_overlap != 0;
}
static int chain_rel_bwd(const RelGroups& G, int stream_y, hipStream_t st) {
  for (const auto& g : G.groups[stream_y]) {
    MMNAS_REQUIRE(g.dWy && g.dby, MMNAS_E_ARG, "chain_bwd: relation operators without the stem layer's gradient sinks");
    const int rc = mmnas_rel_multi_bwd(&g, st);
    if (rc) return rc;
  }
  return MMNAS_OK;
}


// ---- key / value projections of all guided operators of a chain in grouped launches ----
// Every GuidedAtt of a decoder reads the SAME key / value source -- the final language state (hygr_vqa.py:45-52: pre = x) --
// through its own Wk / Wv: M = B * Sx = 896 rows, 56 tiles per product, a quarter of a round of workgroups.  Forward: right
// behind the encoder, K_n = x Wk_n^T and V_n = x Wv_n^T of up to 4 operators per grouped launch (8 groups); each operator
// then projects only its queries.  Backward: behind the LAST guided operator's backward (the first in chain order), the
// key / value source gradients dxkv_n = dK_n Wk_n + dV_n Wv_n into per-operator buffers and dWk_n / dWv_n, 4 operators per
// gradient-pair launch; one add_many launch sums the buffers (and the head's gradient) into the encoder's output gradient.
static int g_node_lnb = -1;   // MMNAS_NODE_LNB, default 1: the sampled candidate's LayerNorm backward inside the node's mix kernel
static bool node_lnb_on() {
  if (g_node_lnb < 0) { const char* e = getenv("MMNAS_NODE_LNB"); g_node_lnb = (e && e[0] ? atoi(e) : 1) ? 1 : 0; }
  return g_node_lnb != 0;
}
static int g_guided_hoist = -1;   // MMNAS_GUIDED_HOIST, default 1
static bool guided_hoist_on() {
  if (g_guided_hoist < 0) { const char* e = getenv("MMNAS_GUIDED_HOIST"); g_guided_hoist = (e && e[0] ? atoi(e) : 1) ? 1 : 0; }
  return g_guided_hoist != 0;
}
struct GuidedSet {
  int idx[MMNAS_CHAIN_MAX_OPS], n;
  bool hoisted[MMNAS_CHAIN_MAX_OPS];
  GuidedSet() : n(0) { memset(hoisted, 0, sizeof(hoisted)); }
};
constexpr int GUIDED_PER_LAUNCH = 4;   // 2 groups each of <= MMNAS_GEMM_MAX_GROUPS = 9
static void chain_guided_set(const mmnas_chain* c, bool bwd, bool allow, GuidedSet& G) {
  if (!allow || !guided_hoist_on() || c->use_side_stream) return;
  int di = -1;
  for (int i = 0; i < c->n_ops; ++i) {
    const mmnas_chain_op& o = c->ops[i];
    if (o.kind != MMNAS_CHAIN_ATT || (o.att.flags & MMNAS_F_SELF)) continue;
    if (bwd && c->mixed && o.detached) continue;
    if (di < 0) di = o.att.di;
    if (o.att.di != di || G.n >= ADD_MANY_MAX - 2) continue;   // (another head layout: that operator keeps its own launches)
    G.idx[G.n++] = i; G.hoisted[i] = true;
  }
  if (G.n < 2) { for (int j = 0; j < G.n; ++j) G.hoisted[G.idx[j]] = false; G.n = 0; }   // (one operator: nothing to merge)
}
static int chain_guided_kv_fwd(const mmnas_chain* c, const ChainLayout& L, const GuidedSet& G, const float* x_final, hipStream_t st) {
  char* base = (char*)c->arena;
  for (int j0 = 0; j0 < G.n; j0 += GUIDED_PER_LAUNCH) {
    const int n = G.n - j0 < GUIDED_PER_LAUNCH ? G.n - j0 : GUIDED_PER_LAUNCH;
    mmnas_gemm_desc g;
    for (int j = 0; j < n; ++j) {
      const int i = G.idx[j0 + j];
      mmnas_att_op a; mmnas_mlp_op m;
      chain_op_setup(c, i, a, m);
      a.save = base + L.save[i]; a.ws = base + L.ws[i];
      const AttLayout al_ = att_layout(&a);
      if (j == 0) { gemm_init(g, MMNAS_GEMM_NT, a.di, c->d, c->d, c->d, a.di); g.ngroups = 2 * n; }
      mmnas_gemm_group* gg = g.g + 2 * j;
      memset(gg, 0, 2 * sizeof(*gg));
      gg[0].M = (int)al_.Mk; gg[0].A[0] = x_final; gg[0].B[0] = a.Wk; gg[0].C = al_.K;
      gg[1].M = (int)al_.Mk; gg[1].A[0] = x_final; gg[1].B[0] = a.Wv; gg[1].C = al_.V;
    }
    const int rc = mmnas_gemm(&g, st);
    if (rc) return rc;
  }
  return MMNAS_OK;
}
// behind the last guided operator's backward: per-operator key / value source gradients + dWk / dWv, then their sum (+ the
// head's gradient of the language state, when there is one) into `out`
static int chain_guided_kv_bwd(const mmnas_chain* c, const ChainLayout& L, const GuidedSet& G, const float* x_final, const float* extra,
                               const float* extra2, float* out, hipStream_t st) {
  char* base = (char*)c->arena;
  const float* srcs[ADD_MANY_MAX + 2];
  int ns = 0;
  if (extra) srcs[ns++] = extra;
  if (extra2) srcs[ns++] = extra2;
  for (int j0 = 0; j0 < G.n; j0 += GUIDED_PER_LAUNCH) {
    const int n = G.n - j0 < GUIDED_PER_LAUNCH ? G.n - j0 : GUIDED_PER_LAUNCH;
    mmnas_gemm_desc g, w;
    for (int j = 0; j < n; ++j) {
      const int i = G.idx[j0 + j];
      mmnas_att_op a; mmnas_mlp_op m;
      chain_op_setup(c, i, a, m);
      a.save = base + L.save[i]; a.ws = base + L.ws[i];
      const AttLayout al_ = att_layout(&a);
      const int Mk = (int)al_.Mk, di = a.di, d = c->d;
      if (j == 0) {
        gemm_init(g, MMNAS_GEMM_NN, d, di, di, d, d); g.ngroups = n; g.nseg = 2;
        gemm_init(w, MMNAS_GEMM_TN, d, Mk, di, d, d); w.ngroups = 2 * n; w.accumulate = 1;
      }
      float* tmp = (float*)(base + L.tmp[i]);
      memset(&g.g[j], 0, sizeof(g.g[j]));
      g.g[j].M = Mk; g.g[j].C = tmp;
      g.g[j].A[0] = al_.dK; g.g[j].B[0] = a.Wk;
      g.g[j].A[1] = al_.dV; g.g[j].B[1] = a.Wv;
      memset(&w.g[2 * j], 0, 2 * sizeof(w.g[0]));
      w.g[2 * j].M = di; w.g[2 * j].A[0] = al_.dK; w.g[2 * j].B[0] = x_final; w.g[2 * j].C = a.dWk;
      w.g[2 * j + 1].M = di; w.g[2 * j + 1].A[0] = al_.dV; w.g[2 * j + 1].B[0] = x_final; w.g[2 * j + 1].C = a.dWv;
      srcs[ns++] = tmp;
    }
    const int rc = mmnas_gemm_pair(&g, &w, st);
    if (rc) return rc;
  }
  return add_many(srcs, ns, out, (size_t)c->B * c->Sx * c->d, st);
}

}  // namespace mmnas

extern "C" int mmnas_set_rel_overlap(int on) {
  const int prev = mmnas::rel_overlap_on() ? 1 : 0;
  mmnas::g_rel_overlap = on ? 1 : 0;
  return prev;
}

extern "C" int mmnas_set_guided_hoist(int on) {
  const int prev = mmnas::guided_hoist_on() ? 1 : 0;
  mmnas::g_guided_hoist = on ? 1 : 0;
  return prev;
}

extern "C" int mmnas_set_rel_hoist(int on) {
  const int prev = mmnas::rel_hoist_on() ? 1 : 0;
  mmnas::g_rel_hoist = on ? 1 : 0;
  return prev;
}

extern "C" int mmnas_set_chain_overlap(int on) {
  const int prev = mmnas::chain_overlap_on() ? 1 : 0;
  mmnas::g_chain_overlap = on ? 1 : 0;
  return prev;
}

extern "C" int mmnas_chain_plan(const mmnas_chain* c, size_t* arena_bytes) {
  int rc = chain_check(c, "chain_plan");
  if (rc) return rc;
  MMNAS_REQUIRE(arena_bytes, MMNAS_E_ARG, "chain_plan: null output");
  ChainLayout L;
  if ((rc = chain_layout(c, L))) return rc;
  *arena_bytes = L.total;
  return MMNAS_OK;
}

namespace mmnas {
// ---- mixed chains: the architecture step (MixedOp modes 'full' / 'two') ----
// Per node: every evaluated candidate's forward up to its pre-LayerNorm sum, then ONE epilogue launch that normalises
// all of them and forms the gated node output (mmnas_node_mix_fwd).  Backward: one launch turns the node's output
// gradient into the gate gradients of all candidates (their outputs recomputed from the saved sums) and the sampled
// candidate's output gradient; only that candidate's backward runs.
struct NodeView { int first, n, active; const float* z[MMNAS_MIXED_MAX]; const float* a[MMNAS_MIXED_MAX]; const float* b[MMNAS_MIXED_MAX]; int cand[MMNAS_MIXED_MAX]; };

// z / LayerNorm pointers of operator i (buffers from the arena): what the node epilogue reads for this candidate
static void chain_cand_view(const mmnas_chain* c, const ChainLayout& L, int i, bool ln_done, const float** z, const float** la, const float** lb) {
  char* base = (char*)c->arena;
  const mmnas_chain_op& o = c->ops[i];
  mmnas_att_op a; mmnas_mlp_op m;
  chain_op_setup(c, i, a, m);
  const bool norm = (o.kind == MMNAS_CHAIN_ATT ? a.flags : m.flags) & MMNAS_F_NORM;
  if (!norm || ln_done) { *z = (const float*)(base + L.y[i]); *la = nullptr; *lb = nullptr; return; }
  if (o.kind == MMNAS_CHAIN_ATT) {
    a.save = base + L.save[i]; a.ws = nullptr;
    *z = att_layout(&a).z; *la = a.ln_a; *lb = a.ln_b;
  } else {
    m.save = base + L.save[i]; m.ws = nullptr;
    *z = mlp_layout(&m).z; *la = m.ln_a; *lb = m.ln_b;
  }
}

// whether operator i's forward normalises its own output even when asked to defer (the one-launch short-sequence kernel)
static bool chain_ln_done_in_op(const mmnas_chain* c, int i) {
  const mmnas_chain_op& o = c->ops[i];
  if (o.kind != MMNAS_CHAIN_ATT) return false;
  mmnas_att_op a; mmnas_mlp_op m;
  chain_op_setup(c, i, a, m);
  return sa_small_applies(&a);
}

static int chain_fwd_mixed(const mmnas_chain* c, hipStream_t st, const ChainLayout& L) {
  char* base = (char*)c->arena;
  const float* cur_x = c->x_in;
  const float* cur_y = c->y_in;
  const size_t nx = (size_t)c->B * c->Sx * c->d * sizeof(float), ny = chain_rows_y(c) * c->d * sizeof(float);
  int rc;
  RelGroups RG;     // every relation candidate's bias (18 in the VQA search space) before the first node
  chain_rel_groups(c, L, false, RG);
  SideCtx* rsc = (rel_overlap_on() && !RG.groups[1].empty() && L.last_x >= 0 && L.first_y >= 0) ? side_ctx(st, true) : nullptr;
  if (rsc) {   // the image stream's candidates beside the encoder nodes (see rel_overlap_on); joined in front of the first decoder node
    if ((rc = ev_fork(st, rsc->side, rsc->ovl[0]))) return rc;
    if ((rc = chain_rel_fwd(RG, rsc->side, 1))) return rc;
    if ((rc = chain_rel_fwd(RG, st, 0))) return rc;
  } else if ((rc = chain_rel_fwd(RG, st))) return rc;
  for (int i0 = 0; i0 < c->n_ops;) {
    int i1 = i0;
    while (i1 < c->n_ops && c->ops[i1].node == c->ops[i0].node) ++i1;
    const bool oy = c->ops[i0].on_y;
    if (rsc && i0 == L.first_y && (rc = ev_fork(rsc->side, st, rsc->ovl[1]))) return rc;   // every relation bias exists from here on
    const float* cur = oy ? cur_y : cur_x;
    const float* z[MMNAS_MIXED_MAX]; const float* la[MMNAS_MIXED_MAX]; const float* lb[MMNAS_MIXED_MAX];
    float gate_order[MMNAS_MIXED_MAX];
    (void)gate_order;
    float eps = 1e-6f;
    // The node's attention candidates on the general path run stage by stage, each stage ONE launch for all of them: their
    // Q / K / V projections (up to 9 groups: same N = d_inside, K = d), their cores, their merge projections (own dropout
    // seed per group).  3 + 3 launches -> 1 + 1 per decoder node of the VQA search space (self / relation / guided).
    mmnas_att_op ga[MMNAS_MIXED_MAX];
    AttLayout gl[MMNAS_MIXED_MAX];
    int gidx[MMNAS_MIXED_MAX], ng = 0;
    for (int i = i0; i < i1; ++i) {
      const mmnas_chain_op& o = c->ops[i];
      if (o.kind != MMNAS_CHAIN_ATT) continue;
      mmnas_att_op a; mmnas_mlp_op m;
      chain_op_setup(c, i, a, m);
      a.xq = cur;
      a.xkv = (a.flags & MMNAS_F_SELF) ? cur : cur_x;
      a.y = (float*)(base + L.y[i]); a.save = base + L.save[i]; a.ws = base + L.ws[i];
      if (sa_small_applies(&a)) continue;
      if (ng > 0) {
        const mmnas_att_op& f = ga[0];
        const bool d0 = (f.flags & MMNAS_F_TRAIN) && f.drop_p > 0.f, d1 = (a.flags & MMNAS_F_TRAIN) && a.drop_p > 0.f;
        if (a.di != f.di || ((a.flags ^ f.flags) & MMNAS_F_RESIDUAL) || d0 != d1 || (d0 && a.drop_p != f.drop_p)) continue;
      }
      if (3 * (ng + 1) > MMNAS_GEMM_MAX_GROUPS) continue;
      if ((rc = att_fwd_args(&a))) return rc;
      ga[ng] = a; gl[ng] = att_layout(&a); gidx[ng] = i; ++ng;
    }
    if (ng < 2) ng = 0;   // (a single one takes the per-operator route below)
    if (ng) {
      const mmnas_att_op& f = ga[0];
      const bool drop = (f.flags & MMNAS_F_TRAIN) && f.drop_p > 0.f;
      mmnas_gemm_desc g;
      gemm_init(g, MMNAS_GEMM_NT, f.di, c->d, c->d, c->d, f.di);
      g.ngroups = 3 * ng;
      for (int j = 0; j < ng; ++j) att_qkv_groups(&ga[j], gl[j], g.g + 3 * j);
      if ((rc = mmnas_gemm(&g, st))) return rc;
      {   // the cores: two of one geometry (self / relation-self) share a launch, the rest one by one
        mmnas_mha_desc md[MMNAS_MIXED_MAX];
        for (int j = 0; j < ng; ++j)
          if ((rc = att_core_fwd(&ga[j], gl[j], st, &md[j], RG.hoisted[gidx[j]]))) return rc;   // (relation biases first, unless hoisted)
        bool done[MMNAS_MIXED_MAX] = {false};
        for (int j = 0; j < ng; ++j) {
          if (done[j]) continue;
          int mate = -1;
          for (int q = j + 1; q < ng && mate < 0; ++q)
            if (!done[q] && md[q].Sq == md[j].Sq && md[q].Sk == md[j].Sk && md[q].H == md[j].H && md[q].dh == md[j].dh) mate = q;
          if (mate >= 0) { if ((rc = mha_core_fwd_pair(&md[j], &md[mate], st))) return rc; done[mate] = true; }
          else if ((rc = mmnas_mha_core_fwd(&md[j], st))) return rc;
          done[j] = true;
        }
      }
      gemm_init(g, MMNAS_GEMM_NT, c->d, f.di, f.di, f.di, c->d);
      g.ngroups = ng;
      for (int j = 0; j < ng; ++j) att_merge_group(&ga[j], gl[j], &g.g[j]);
      if (f.flags & MMNAS_F_RESIDUAL) g.ldres = c->d;
      if (drop) { g.drop_p = f.drop_p; g.drop_site = 1; g.drop_seed = f.seed; }
      if ((rc = mmnas_gemm(&g, st))) return rc;
    }
    for (int i = i0; i < i1; ++i) {
      const mmnas_chain_op& o = c->ops[i];
      mmnas_att_op a; mmnas_mlp_op m;
      chain_op_setup(c, i, a, m);
      bool ln_done = true;
      bool staged = false;
      for (int j = 0; j < ng; ++j) staged |= gidx[j] == i;
      if (staged) {   // (NORM: the node epilogue normalises; otherwise the merge product wrote the output)
        ln_done = !(a.flags & MMNAS_F_NORM);
        if (a.flags & MMNAS_F_NORM) eps = a.eps;
        chain_cand_view(c, L, i, ln_done, &z[i - i0], &la[i - i0], &lb[i - i0]);
        continue;
      }
      if (o.kind == MMNAS_CHAIN_ATT) {
        a.xq = cur;
        a.xkv = (a.flags & MMNAS_F_SELF) ? cur : cur_x;
        a.y = (float*)(base + L.y[i]); a.save = base + L.save[i]; a.ws = base + L.ws[i];
        if ((rc = att_fwd_impl(&a, st, true, &ln_done, RG.hoisted[i]))) return rc;
        if (a.flags & MMNAS_F_NORM) eps = a.eps;
      } else {
        m.x = cur; m.y = (float*)(base + L.y[i]); m.save = base + L.save[i]; m.ws = base + L.ws[i];
        if ((rc = mlp_fwd_impl(&m, st, true, &ln_done))) return rc;
        if (m.flags & MMNAS_F_NORM) eps = m.eps;
      }
      chain_cand_view(c, L, i, ln_done, &z[i - i0], &la[i - i0], &lb[i - i0]);
    }
    // the gate values of the node's candidates, in operator order: a row of the [n_nodes, width] block picked apart
    // by candidate index -- the kernel reads gate[j] for operator j, so the operators are passed in candidate order
    const float* zs[MMNAS_MIXED_MAX] = {nullptr}; const float* as[MMNAS_MIXED_MAX] = {nullptr}; const float* bs[MMNAS_MIXED_MAX] = {nullptr};
    int width = 0;
    for (int i = i0; i < i1; ++i) {
      const int k = c->ops[i].cand;
      zs[k] = z[i - i0]; as[k] = la[i - i0]; bs[k] = lb[i - i0];
      if (k + 1 > width) width = k + 1;
    }
    // (candidates of the node that were not evaluated -- mode 'two' -- stay NULL: no term in the sum, no gate gradient)
    float* out = i0 == L.last_x ? c->x_out : (i0 == L.last_y ? c->y_out : (float*)(base + L.nout[i0]));
    const int M = oy ? (int)chain_rows_y(c) : c->B * c->Sx;
    if ((rc = mmnas_node_mix_fwd(zs, as, bs, width, c->gate + (size_t)c->ops[i0].node * c->gate_width, out, M, c->d, eps, st))) return rc;
    if (oy) cur_y = out; else cur_x = out;
    i0 = i1;
  }
  if (L.last_x < 0 && hipMemcpyAsync(c->x_out, c->x_in, nx, hipMemcpyDeviceToDevice, st) != hipSuccess) return MMNAS_E_LAUNCH;
  if (L.last_y < 0 && hipMemcpyAsync(c->y_out, c->y_in, ny, hipMemcpyDeviceToDevice, st) != hipSuccess) return MMNAS_E_LAUNCH;
  return MMNAS_OK;
}

static int chain_bwd_mixed(const mmnas_chain* c, hipStream_t st, const ChainLayout& L) {
  MMNAS_REQUIRE(c->dgate, MMNAS_E_ARG, "chain_bwd: mixed chain without a gate-gradient block");
  char* base = (char*)c->arena;
  const size_t ex = (size_t)c->B * c->Sx * c->d, ey = chain_rows_y(c) * c->d;
  float* dpre = (float*)(base + L.dpre);
  int rc;
  GuidedSet GS;     // the sampled guided candidates: dK / dV wait for the grouped launches behind the first of them
  chain_guided_set(c, true, true, GS);
  int Gm = -1;      // chain index of the first differentiated guided candidate (the last in backward order)
  for (int i = 0; i < c->n_ops && Gm < 0; ++i)
    if (GS.hoisted[i]) Gm = i;
  const float* enc_grad = nullptr;
  if (L.n_guided > GS.n && hipMemsetAsync(dpre, 0, ex * sizeof(float), st) != hipSuccess) return MMNAS_E_LAUNCH;
  // node starts in order
  int starts[MMNAS_CHAIN_MAX_OPS + 1], nn = 0;
  for (int i = 0; i < c->n_ops; ++i)
    if (i == 0 || c->ops[i].node != c->ops[i - 1].node) starts[nn++] = i;
  starts[nn] = c->n_ops;
  auto node_out = [&](int i0) -> const float* { return i0 == L.last_x ? c->x_out : (i0 == L.last_y ? c->y_out : (const float*)(base + L.nout[i0])); };
  auto input_of = [&](int k) -> const float* {   // input of node k = output of the previous node on its stream
    const bool oy = c->ops[starts[k]].on_y;
    for (int j = k - 1; j >= 0; --j)
      if ((bool)c->ops[starts[j]].on_y == oy) return node_out(starts[j]);
    return oy ? c->y_in : c->x_in;
  };
  const float* x_final = L.last_x >= 0 ? c->x_out : c->x_in;
  const float* cur_dy = c->dy_out;
  const float* red_part[MMNAS_CHAIN_MAX_OPS]; float* red_out[MMNAS_CHAIN_MAX_OPS]; int red_nwg[MMNAS_CHAIN_MAX_OPS], red_n[MMNAS_CHAIN_MAX_OPS];
  int n_red = 0;
  RelGroups RG;     // the sampled relation candidates: their bias gradients wait for one launch behind the stream's first node
  chain_rel_groups(c, L, true, RG);
  SideCtx* rel_bwd_side = nullptr;
  for (int k = nn - 1; k >= 0; --k) {
    const int i0 = starts[k], i1 = starts[k + 1];
    const bool oy = c->ops[i0].on_y;
    if (!oy && i0 == L.last_x) {   // entering the encoder: its output gradient = the head's + the guided operators'
      MMNAS_REQUIRE(c->dx_out || L.n_guided, MMNAS_E_ARG, "chain_bwd: no gradient reaches the encoder (dx_out NULL, no guided operator)");
      if (enc_grad) cur_dy = enc_grad;
      else if (c->dx_out && L.n_guided) {
        float* g0 = (float*)(base + L.encdy);
        if ((rc = mmnas_drop_add(c->dx_out, dpre, g0, ex, 0.f, 0, 0, st))) return rc;
        cur_dy = g0;
      } else cur_dy = c->dx_out ? c->dx_out : dpre;
    }
    const float* zs[MMNAS_MIXED_MAX] = {nullptr}; const float* as[MMNAS_MIXED_MAX] = {nullptr}; const float* bs[MMNAS_MIXED_MAX] = {nullptr};
    int width = 0, act = -1, act_op = -1;
    float eps = 1e-6f;
    for (int i = i0; i < i1; ++i) {
      const int kc = c->ops[i].cand;
      chain_cand_view(c, L, i, chain_ln_done_in_op(c, i), &zs[kc], &as[kc], &bs[kc]);
      if (kc + 1 > width) width = kc + 1;
      if (!c->ops[i].detached) { act = kc; act_op = i; }
      const mmnas_chain_op& o = c->ops[i];
      if (o.kind == MMNAS_CHAIN_ATT) { if (o.att.flags & MMNAS_F_NORM) eps = o.att.eps; }
      else if (o.mlp.flags & MMNAS_F_NORM) eps = o.mlp.eps;
    }
    const float* nin = input_of(k);
    float* dact = (float*)(base + L.ndact[i0]);
    const int M = oy ? (int)chain_rows_y(c) : c->B * c->Sx;
    const size_t grow = (size_t)c->ops[i0].node * c->gate_width;
    // the sampled candidate: its descriptor first -- its LayerNorm backward rides in the node's mix kernel (round 6) unless the
    // candidate takes the one-launch short-sequence backward, which does its own
    const mmnas_chain_op& o = c->ops[act_op];
    mmnas_att_op a; mmnas_mlp_op m;
    chain_op_setup(c, act_op, a, m);
    float* dx = i0 == L.first_x ? c->dx_in : (i0 == L.first_y ? c->dy_in : (float*)(base + L.dx[act_op]));
    NodeLnBwd lnb;
    lnb.dz = nullptr;
    AuxReduce lnpre;
    lnpre.part = nullptr;
    if (o.kind == MMNAS_CHAIN_ATT) {
      const bool self = a.flags & MMNAS_F_SELF;
      a.xq = nin;
      a.xkv = self ? nin : x_final;
      a.save = base + L.save[act_op]; a.ws = base + L.ws[act_op];
      a.dy = dact; a.dxq = dx;
      a.dxkv = self ? nullptr : dpre;
      a.drel = nullptr;
      if (node_lnb_on() && (a.flags & MMNAS_F_NORM) && c->d <= 256 && !(self && !(a.flags & MMNAS_F_REL) && sa_small_bwd_applies(&a))) {
        const AttLayout AL = att_layout(&a);
        const bool drop = (a.flags & MMNAS_F_TRAIN) && a.drop_p > 0.f;
        lnb.dz = AL.dz; lnb.dt = drop ? AL.dt : nullptr; lnb.part = AL.lnws;
        lnb.drop = make_drop(drop ? a.drop_p : 0.f, a.seed, 1);
        lnpre.part = AL.lnws; lnpre.d = c->d;
        lnpre.out[0] = a.dln_a; lnpre.out[1] = a.dln_b; lnpre.out[2] = nullptr;
      }
    } else {
      m.x = nin; m.save = base + L.save[act_op]; m.ws = base + L.ws[act_op];
      m.dy = dact; m.dx = dx;
      if (node_lnb_on() && (m.flags & MMNAS_F_NORM) && c->d <= 256) {
        const MlpLayout ML = mlp_layout(&m);
        const bool drop = (m.flags & MMNAS_F_TRAIN) && m.drop_p > 0.f;
        lnb.dz = ML.dz; lnb.dt = drop ? ML.dt : nullptr; lnb.part = ML.lnws;
        lnb.drop = make_drop(drop ? m.drop_p : 0.f, m.seed, 1);
        lnpre.part = ML.lnws; lnpre.d = c->d;
        lnpre.out[0] = m.dln_a; lnpre.out[1] = m.dln_b; lnpre.out[2] = (drop && m.db[m.nl - 1]) ? m.db[m.nl - 1] : nullptr;
      }
    }
    {
      float* wsk = (float*)(base + L.mixws) + (size_t)k * mmnas_mixed_sum_ws_floats();
      int nwg = 0;
      if ((rc = node_mix_bwd_impl(zs, as, bs, width, c->gate + grow, cur_dy, dact, act, c->dgate + grow, wsk, M, c->d, eps, st, false, &nwg,
                                  lnb.dz ? &lnb : nullptr)))
        return rc;
      if (nwg > 0) { red_part[n_red] = wsk; red_out[n_red] = c->dgate + grow; red_nwg[n_red] = nwg; red_n[n_red] = width; ++n_red; }
      lnpre.nrows = nwg;
    }
    // the sampled candidate's backward
    if (o.kind == MMNAS_CHAIN_ATT) {
      const bool self = a.flags & MMNAS_F_SELF;
      if ((rc = att_bwd_impl(&a, st, nullptr, !self, RG.hoisted[act_op], GS.hoisted[act_op], lnb.dz ? &lnpre : nullptr))) return rc;
      if (GS.n && act_op == Gm) {
        float* g0 = (float*)(base + L.encdy);
        if ((rc = chain_guided_kv_bwd(c, L, GS, x_final, c->dx_out, GS.n < L.n_guided ? dpre : nullptr, g0, st))) return rc;
        enc_grad = g0;
      }
    } else {
      if ((rc = mlp_bwd_impl(&m, st, nullptr, lnb.dz ? &lnpre : nullptr))) return rc;
    }
    cur_dy = dx;
    hipStream_t mark_stream = st;
    if (i0 == L.first_y && !RG.groups[1].empty()) {
      SideCtx* rsc = (rel_overlap_on() && L.last_x >= 0) ? side_ctx(st, true) : nullptr;
      if (rsc) {   // beside the encoder nodes' backward; joined at the end of the call
        if ((rc = ev_fork(st, rsc->side, rsc->ovl[2]))) return rc;
        if ((rc = chain_rel_bwd(RG, 1, rsc->side))) return rc;
        rel_bwd_side = rsc;
        mark_stream = rsc->side;
      } else if ((rc = chain_rel_bwd(RG, 1, st))) return rc;
    }
    if (i0 == L.first_x && !RG.groups[0].empty()) {
      // (every relation group shares ONE partial-row workspace: the image stream's launches on the side stream are joined
      //  before the language stream's write into it -- ADVICE r5; the join at the end of the call then finds nothing pending)
      if (rel_bwd_side) { if ((rc = ev_fork(rel_bwd_side->side, st, rel_bwd_side->ovl[3]))) return rc; rel_bwd_side = nullptr; }
      if ((rc = chain_rel_bwd(RG, 0, st))) return rc;
    }
    if (c->marks && c->marks[act_op] && hipEventRecord((hipEvent_t)c->marks[act_op], mark_stream) != hipSuccess) {
      set_error("chain_bwd: cannot record the mark event of operator %d", act_op);
      return MMNAS_E_LAUNCH;
    }
  }
  if (rel_bwd_side && (rc = ev_fork(rel_bwd_side->side, st, rel_bwd_side->ovl[3]))) return rc;   // the relation gradients are in
  if (n_red && (rc = mixed_reduce_many(red_part, red_out, red_nwg, red_n, n_red, st))) return rc;   // every node's gate gradients
  if (L.first_y < 0 && hipMemcpyAsync(c->dy_in, c->dy_out, ey * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return MMNAS_E_LAUNCH;
  if (L.first_x < 0) {
    const float* g0 = c->dx_out;
    if (enc_grad) g0 = enc_grad;
    else if (c->dx_out && L.n_guided) {
      if ((rc = mmnas_drop_add(c->dx_out, dpre, c->dx_in, ex, 0.f, 0, 0, st))) return rc;
      g0 = nullptr;
    } else if (!c->dx_out) g0 = dpre;
    if (g0 && hipMemcpyAsync(c->dx_in, g0, ex * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return MMNAS_E_LAUNCH;
  }
  return MMNAS_OK;
}
}  // namespace mmnas

extern "C" int mmnas_chain_fwd(const mmnas_chain* c, void* stream) {
  int rc = chain_check(c, "chain_fwd");
  if (rc) return rc;
  MMNAS_REQUIRE(c->x_in && c->y_in && c->x_out && c->y_out && c->arena, MMNAS_E_ARG, "chain_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  ChainLayout L;
  if ((rc = chain_layout(c, L))) return rc;
  if (c->mixed) return chain_fwd_mixed(c, st, L);
  char* base = (char*)c->arena;
  const float* cur_x = c->x_in;
  const float* cur_y = c->y_in;
  const size_t nx = (size_t)c->B * c->Sx * c->d * sizeof(float), ny = chain_rows_y(c) * c->d * sizeof(float);
  RelGroups RG;     // the relation bias of every relation operator of the chain: one launch per stream, here
  chain_rel_groups(c, L, false, RG);
  const int G = first_guided(c);
  const bool ovl = chain_overlap_on() && !c->use_side_stream && L.last_x >= 0 && L.first_y >= 0 && G > L.first_y;
  // the image stream's bias beside the encoder (see rel_overlap_on): joined in front of the decoder's first relation operator
  int rel_join_at = -1;
  SideCtx* rsc = nullptr;
  if (!ovl && rel_overlap_on() && !RG.groups[1].empty() && L.last_x >= 0 && (rsc = side_ctx(st, true)) != nullptr) {
    for (int i = L.first_y; i < c->n_ops && rel_join_at < 0; ++i)
      if (RG.hoisted[i] && c->ops[i].on_y) rel_join_at = i;
  }
  if (rel_join_at >= 0) {
    if ((rc = ev_fork(st, rsc->side, rsc->ovl[0]))) return rc;
    if ((rc = chain_rel_fwd(RG, rsc->side, 1))) return rc;
    if ((rc = chain_rel_fwd(RG, st, 0))) return rc;
  } else if ((rc = chain_rel_fwd(RG, st))) return rc;
  GuidedSet GS;
  chain_guided_set(c, false, !ovl, GS);
  auto run = [&](int i, hipStream_t s) -> int {
    const mmnas_chain_op& o = c->ops[i];
    mmnas_att_op a; mmnas_mlp_op m;
    chain_op_setup(c, i, a, m);
    const float* cur = o.on_y ? cur_y : cur_x;
    float* out = i == L.last_x ? c->x_out : (i == L.last_y ? c->y_out : (float*)(base + L.y[i]));
    int r;
    if (o.kind == MMNAS_CHAIN_ATT) {
      a.xq = cur;
      a.xkv = (a.flags & MMNAS_F_SELF) ? cur : cur_x;   // guided: keys / values from the FINAL language state
      a.y = out; a.save = base + L.save[i]; a.ws = base + L.ws[i];
      bool done;
      if ((r = att_fwd_impl(&a, s, false, &done, RG.hoisted[i], GS.hoisted[i]))) return r;
    } else {
      m.x = cur; m.y = out; m.save = base + L.save[i]; m.ws = base + L.ws[i];
      if ((r = mmnas_mlp_op_fwd(&m, s))) return r;
    }
    if (o.on_y) cur_y = out; else cur_x = out;
    return MMNAS_OK;
  };
  if (!ovl) {
    for (int i = 0; i < c->n_ops; ++i) {
      // the final language state exists: key / value projections of every guided operator (grouped launches)
      if (GS.n && i == G && (rc = chain_guided_kv_fwd(c, L, GS, cur_x, st))) return rc;
      if (i == rel_join_at && (rc = ev_fork(rsc->side, st, rsc->ovl[1]))) return rc;   // the image stream's bias exists from here on
      if ((rc = run(i, st))) return rc;
    }
  } else {
    SideCtx* sc = side_ctx(st, true);
    MMNAS_REQUIRE(sc, MMNAS_E_LAUNCH, "chain_fwd: cannot create the encoder stream");
    if ((rc = ev_fork(st, sc->enc, sc->ovl[0]))) return rc;
    // encoder operators [first_x, last_x] on their own stream, the decoder's [first_y, G) on the caller's: issued alternately
    int ie = L.first_x, id = L.first_y;
    while (ie <= L.last_x || id < G) {
      if (ie <= L.last_x && (rc = run(ie++, sc->enc))) return rc;
      if (id < G && (rc = run(id++, st))) return rc;
    }
    if ((rc = ev_fork(sc->enc, st, sc->ovl[1]))) return rc;   // the final language state exists from here on
    for (int i = G; i < c->n_ops; ++i)
      if ((rc = run(i, st))) return rc;
  }
  if (L.last_x < 0 && hipMemcpyAsync(c->x_out, c->x_in, nx, hipMemcpyDeviceToDevice, st) != hipSuccess) return MMNAS_E_LAUNCH;
  if (L.last_y < 0 && hipMemcpyAsync(c->y_out, c->y_in, ny, hipMemcpyDeviceToDevice, st) != hipSuccess) return MMNAS_E_LAUNCH;
  return MMNAS_OK;
}

extern "C" int mmnas_chain_bwd(const mmnas_chain* c, void* stream) {
  int rc = chain_check(c, "chain_bwd");
  if (rc) return rc;
  MMNAS_REQUIRE(c->x_in && c->y_in && c->x_out && c->y_out && c->arena && c->dy_out && c->dx_in && c->dy_in, MMNAS_E_ARG,
                "chain_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  ChainLayout L;
  if ((rc = chain_layout(c, L))) return rc;
  if (c->mixed) return chain_bwd_mixed(c, st, L);
  SideCtx* sc = c->use_side_stream ? side_ctx(st, true) : nullptr;
  MMNAS_REQUIRE(!c->use_side_stream || sc, MMNAS_E_LAUNCH, "chain_bwd: cannot create the side stream");
  hipStream_t side = sc ? sc->side : nullptr;
  if (sc) sc->used = true;
  SideQueue q;
  q.rel_only = c->use_side_stream == 2;
  SideQueue* sq = sc ? &q : nullptr;
  // When the queued parameter-gradient work is released: MMNAS_SIDE_FLUSH=op -> behind every operator (it then competes
  // with the data-gradient chain for the CUs); default -> once behind the decoder and once behind the encoder, so it
  // fills the latency-bound tail of the backward pass (encoder on 896 rows, LSTM, stem) instead.
  static const bool per_op = getenv("MMNAS_SIDE_FLUSH") && !strcmp(getenv("MMNAS_SIDE_FLUSH"), "op");
  char* base = (char*)c->arena;
  const size_t ex = (size_t)c->B * c->Sx * c->d, ey = chain_rows_y(c) * c->d;
  float* dpre = (float*)(base + L.dpre);
  // inputs of operator i = outputs of the previous operator on its stream
  auto input_of = [&](int i) -> const float* {
    const bool oy = c->ops[i].on_y;
    for (int j = i - 1; j >= 0; --j)
      if ((bool)c->ops[j].on_y == oy) return j == L.last_x ? c->x_out : (j == L.last_y ? c->y_out : (const float*)(base + L.y[j]));
    return oy ? c->y_in : c->x_in;
  };
  const float* x_final = L.last_x >= 0 ? c->x_out : c->x_in;
  const int G = first_guided(c);
  const bool ovl = chain_overlap_on() && !c->use_side_stream && L.last_x >= 0 && L.first_y >= 0 && G > L.first_y;
  RelGroups RG;     // relation operators whose bias gradient waits for the one launch behind their stream's first operator
  if (!ovl) chain_rel_groups(c, L, true, RG);
  GuidedSet GS;     // guided operators whose dK / dV wait for the grouped launches behind operator G
  chain_guided_set(c, true, !ovl, GS);
  const float* enc_grad = nullptr;   // set by the hoisted launch: the encoder's output gradient, complete
  SideCtx* rel_bwd_side = nullptr;   // set when the image stream's relation backward went to the side stream: joined below
  if (L.n_guided > GS.n && hipMemsetAsync(dpre, 0, ex * sizeof(float), st) != hipSuccess) return MMNAS_E_LAUNCH;
  // one operator's backward on stream s: gradient of its output in, gradient of its input out (returned through *dxo)
  auto run = [&](int i, hipStream_t s, const float* dyi, const float** dxo) -> int {
    const mmnas_chain_op& o = c->ops[i];
    mmnas_att_op a; mmnas_mlp_op m;
    chain_op_setup(c, i, a, m);
    float* dx = i == L.first_x ? c->dx_in : (i == L.first_y ? c->dy_in : (float*)(base + L.dx[i]));
    int r;
    if (o.kind == MMNAS_CHAIN_ATT) {
      const bool self = a.flags & MMNAS_F_SELF;
      a.xq = input_of(i);
      a.xkv = self ? a.xq : x_final;
      a.save = base + L.save[i]; a.ws = base + L.ws[i];
      a.dy = dyi; a.dxq = dx;
      a.dxkv = self ? nullptr : dpre;     // guided: added into the running sum (zeroed above)
      a.drel = nullptr;
      if ((r = att_bwd_impl(&a, s, sq, !self, RG.hoisted[i], GS.hoisted[i]))) return r;
      if (GS.n && i == G) {   // the last guided operator in backward order: every hoisted dK / dV exists
        float* g0 = (float*)(base + L.encdy);
        if ((r = chain_guided_kv_bwd(c, L, GS, x_final, c->dx_out, GS.n < L.n_guided ? dpre : nullptr, g0, s))) return r;
        enc_grad = g0;
      }
    } else {
      m.x = input_of(i); m.save = base + L.save[i]; m.ws = base + L.ws[i];
      m.dy = dyi; m.dx = dx;
      if ((r = mlp_bwd_impl(&m, s, sq))) return r;
    }
    *dxo = dx;
    // (the stream's first operator is its last in backward order: every relation operator's bias gradient exists now --
    //  before this operator's mark, which covers the relation parameters, see nets._chain)
    hipStream_t mark_stream = s;
    if (i == L.first_y && !RG.groups[1].empty()) {
      SideCtx* rsc = (rel_overlap_on() && L.last_x >= 0 && !sq) ? side_ctx(st, true) : nullptr;
      if (rsc) {   // beside the encoder's backward (see rel_overlap_on); joined at the end of the call
        if ((r = ev_fork(s, rsc->side, rsc->ovl[2]))) return r;
        if ((r = chain_rel_bwd(RG, 1, rsc->side))) return r;
        rel_bwd_side = rsc;
        mark_stream = rsc->side;
      } else if ((r = chain_rel_bwd(RG, 1, s))) return r;
    }
    if (i == L.first_x && !RG.groups[0].empty()) {
      // (shared partial-row workspace: join the image stream's relation backward first, see chain_bwd_mixed)
      if (rel_bwd_side) { if ((r = ev_fork(rel_bwd_side->side, s, rel_bwd_side->ovl[3]))) return r; rel_bwd_side = nullptr; }
      if ((r = chain_rel_bwd(RG, 0, s))) return r;
    }
    // (encoder / decoder overlap: an event behind operator i on ONE of the two streams says nothing about the operators
    //  with larger indices still running on the other -- the marks are recorded behind the join below instead)
    if (!ovl && c->marks && c->marks[i] && hipEventRecord((hipEvent_t)c->marks[i], mark_stream) != hipSuccess) {
      set_error("chain_bwd: cannot record the mark event of operator %d", i);
      return MMNAS_E_LAUNCH;
    }
    return MMNAS_OK;
  };
  // entering the encoder: its output gradient = head's + the guided operators'
  auto encoder_dy = [&](hipStream_t s, const float** out) -> int {
    MMNAS_REQUIRE(c->dx_out || L.n_guided, MMNAS_E_ARG, "chain_bwd: no gradient reaches the encoder (dx_out NULL, no guided operator)");
    if (enc_grad) { *out = enc_grad; return MMNAS_OK; }
    if (c->dx_out && L.n_guided) {
      float* g0 = (float*)(base + L.encdy);
      const int r = mmnas_drop_add(c->dx_out, dpre, g0, ex, 0.f, 0, 0, s);
      if (r) return r;
      *out = g0;
    } else *out = c->dx_out ? c->dx_out : dpre;
    return MMNAS_OK;
  };
  const float* cur_dy = c->dy_out;
  if (!ovl) {
    for (int i = c->n_ops - 1; i >= 0; --i) {
      if (sq && i == L.last_x && (rc = q.flush(st, side, sc->ev[0]))) return rc;   // the decoder's parameter-gradient work
      if (!c->ops[i].on_y && i == L.last_x && (rc = encoder_dy(st, &cur_dy))) return rc;
      if ((rc = run(i, st, cur_dy, &cur_dy))) return rc;
      if (sq && per_op && (rc = q.flush(st, side, sc->ev[i + 1]))) return rc;
    }
    if (sq && (rc = q.flush(st, side, sc->ev[1]))) return rc;
  } else {
    SideCtx* oc = side_ctx(st, true);
    MMNAS_REQUIRE(oc, MMNAS_E_LAUNCH, "chain_bwd: cannot create the encoder stream");
    for (int i = c->n_ops - 1; i >= G; --i)        // every guided operator is in here: the language state's gradient completes
      if ((rc = run(i, st, cur_dy, &cur_dy))) return rc;
    if ((rc = ev_fork(st, oc->enc, oc->ovl[2]))) return rc;
    const float* enc_dy = nullptr;
    if ((rc = encoder_dy(oc->enc, &enc_dy))) return rc;
    int ie = L.last_x, id = G - 1;
    while (ie >= L.first_x || id >= L.first_y) {    // encoder on its stream beside the decoder's leading operators
      if (ie >= L.first_x && (rc = run(ie--, oc->enc, enc_dy, &enc_dy))) return rc;
      if (id >= L.first_y && (rc = run(id--, st, cur_dy, &cur_dy))) return rc;
    }
    if ((rc = ev_fork(oc->enc, st, oc->ovl[3]))) return rc;
    if (c->marks)
      for (int i = 0; i < c->n_ops; ++i)
        if (c->marks[i] && hipEventRecord((hipEvent_t)c->marks[i], st) != hipSuccess) {
          set_error("chain_bwd: cannot record the mark event of operator %d", i);
          return MMNAS_E_LAUNCH;
        }
  }
  if (rel_bwd_side && (rc = ev_fork(rel_bwd_side->side, st, rel_bwd_side->ovl[3]))) return rc;   // the relation gradients are in
  if (L.first_y < 0 && hipMemcpyAsync(c->dy_in, c->dy_out, ey * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return MMNAS_E_LAUNCH;
  if (L.first_x < 0) {
    const float* g0 = c->dx_out;
    if (enc_grad) g0 = enc_grad;
    else if (c->dx_out && L.n_guided) {
      if ((rc = mmnas_drop_add(c->dx_out, dpre, c->dx_in, ex, 0.f, 0, 0, stream))) return rc;
      g0 = nullptr;
    } else if (!c->dx_out) g0 = dpre;
    if (g0 && hipMemcpyAsync(c->dx_in, g0, ex * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return MMNAS_E_LAUNCH;
  }
  return MMNAS_OK;
}

extern "C" int mmnas_chain_join(void* main_stream, void* waiting_stream) {
  SideCtx* sc = side_ctx((hipStream_t)main_stream, false);
  if (!sc || !sc->used) return MMNAS_OK;
  hipEvent_t e = sc->ev[MMNAS_CHAIN_MAX_OPS];
  if (hipEventRecord(e, sc->side) != hipSuccess || hipStreamWaitEvent((hipStream_t)waiting_stream, e, 0) != hipSuccess) {
    set_error("chain_join: event record / wait failed");
    return MMNAS_E_LAUNCH;
  }
  return MMNAS_OK;
}

// ------------------------------------------------------------------------------------------ answer head
// AttFlat(x) + AttFlat(y) -> LayerNorm -> answer projection (hygr_vqa.py:113-119 / full_vqa.py:105-114 with
// modules.py:59-85) as one call per direction: ~28 launches issued back to back instead of ~75 behind 40 autograd
// nodes (the head's kernels take 5-15 us each, so the per-node host cost -- not the GPU -- set its duration).
namespace mmnas {

struct HeadSideLayout { float *h, *logit, *probs, *pooled, *dpooled, *dlog, *dh, *dxpool, *g1part; };
static inline int head_rows(const mmnas_head* hd, const mmnas_attflat_side& sd) { return sd.off ? sd.M : hd->B * sd.S; }
struct HeadLayout {
  HeadSideLayout s[2];
  float *xo, *sum, *xy, *dxy, *dsum, *lnws, *g1proj, *dlogT;
  size_t total;
};

static HeadLayout head_layout(const mmnas_head* hd) {
  HeadLayout L;
  Carver c(hd->arena);
  const size_t B = hd->B;
  for (int k = 0; k < 2; ++k) {
    const mmnas_attflat_side& sd = k ? hd->sy : hd->sx;
    const size_t M = sd.off ? (size_t)sd.M : B * sd.S;
    HeadSideLayout& s = L.s[k];
    s.h = c.take(M * hd->MID); s.logit = c.take(M * hd->G); s.probs = c.take(M * hd->G); s.pooled = c.take(B * hd->G * hd->d);
    s.dpooled = c.take(B * hd->G * hd->d); s.dlog = c.take(M * hd->G); s.dh = c.take(M * hd->MID); s.dxpool = c.take(M * hd->d);
    // one glimpse: partial column sums of the glimpse-logit backward (head.hip), reduced by the next pair launch
    s.g1part = (hd->G == 1 && glimpse1_supported(hd->MID)) ? c.take((size_t)glimpse1_bwd_blocks((long)M, hd->MID) * 3 * hd->MID) : nullptr;
  }
  L.xo = c.take(B * hd->OUT); L.sum = c.take(B * hd->OUT); L.xy = c.take(B * hd->OUT);
  L.dxy = c.take(B * hd->OUT); L.dsum = c.take(B * hd->OUT);
  L.lnws = c.take(mmnas_layernorm_bwd_ws_floats(hd->B, hd->OUT));
  // one answer unit (the ITM matching score): the projection through the one-unit kernels of head.hip
  L.g1proj = (hd->ANS == 1 && glimpse1_supported(hd->OUT)) ? c.take((size_t)glimpse1_bwd_blocks((long)B, hd->OUT) * 3 * hd->OUT) : nullptr;
  // an answer layer whose width is no multiple of 4 (3129): its loss gradient [B, ANS] has unaligned rows, which put both
  // gradient products of the projection on the guarded-load path (26.7 + 14.8 us); transposed once to [ANS, B] they are a
  // TN product (dxy = dlogits Wp, reduction over the answers) and an NN product (dWp += dlogits^T xy) with aligned rows
  L.dlogT = (hd->ANS % 4 != 0 && hd->B % 4 == 0 && hd->ANS > 1) ? c.take((size_t)hd->ANS * B) : nullptr;
  L.total = c.off;
  return L;
}

static int head_check(const mmnas_head* hd, const char* who) {
  MMNAS_REQUIRE(hd, MMNAS_E_ARG, "%s: null descriptor", who);
  MMNAS_REQUIRE(hd->B > 0 && hd->d > 0 && hd->MID > 0 && hd->G > 0 && hd->OUT > 0 && hd->ANS > 0 && hd->sx.S > 0 && hd->sy.S > 0,
                MMNAS_E_SHAPE, "%s: B=%d d=%d MID=%d G=%d OUT=%d ANS=%d Sx=%d Sy=%d", who, hd->B, hd->d, hd->MID, hd->G, hd->OUT,
                hd->ANS, hd->sx.S, hd->sy.S);
  MMNAS_REQUIRE(hd->d % 4 == 0 && hd->MID % 4 == 0 && hd->OUT % 4 == 0, MMNAS_E_SHAPE, "%s: d, MID, OUT must be multiples of 4", who);
  for (int k = 0; k < 2; ++k) {
    const mmnas_attflat_side& sd = k ? hd->sy : hd->sx;
    if (sd.off) MMNAS_REQUIRE(sd.M > 0 && sd.M <= hd->B * sd.S, MMNAS_E_ARG, "%s: packed side %d: M=%d of B*S=%d rows", who, k, sd.M, hd->B * sd.S);
  }
  return MMNAS_OK;
}

}  // namespace mmnas

extern "C" int mmnas_head_plan(const mmnas_head* hd, size_t* arena_bytes) {
  int rc = head_check(hd, "head_plan");
  if (rc) return rc;
  MMNAS_REQUIRE(arena_bytes, MMNAS_E_ARG, "head_plan: null output");
  mmnas_head tmp = *hd;
  tmp.arena = nullptr;
  *arena_bytes = head_layout(&tmp).total;
  return MMNAS_OK;
}

extern "C" int mmnas_head_fwd(const mmnas_head* hd, void* stream) {
  int rc = head_check(hd, "head_fwd");
  if (rc) return rc;
  MMNAS_REQUIRE(hd->arena && hd->logits && hd->ln_a && hd->ln_b && hd->Wp, MMNAS_E_ARG, "head_fwd: null pointer");
  HeadLayout L = head_layout(hd);
  const bool drop = (hd->flags & MMNAS_F_TRAIN) && hd->drop_p > 0.f;
  const int B = hd->B, d = hd->d, MID = hd->MID, G = hd->G, OUT = hd->OUT;
  mmnas_gemm_desc g;
  hipStream_t st = (hipStream_t)stream;
  // The two AttFlat branches are independent chains of four small launches each.  MMNAS_HEAD_OVERLAP=1 runs the
  // language side's on the second stream beside the image side's, joined where the image side's merge adds its result.
  // MEASURED: slower (supernet step 5.61 -> 5.70 ms, two alternations): each cross-stream event wait costs more than the
  // ~10 us launch it hides.  Off by default -- the third form of stream overlap tried on this path, none of which paid.
  // Default: one stream, and the two sides' glimpse-logit products share a launch (two groups of one grouped GEMM).
  SideCtx* oc = head_overlap_on() ? side_ctx(st, true) : nullptr;
  if (oc && (rc = ev_fork(st, oc->enc, oc->ovl[0]))) return rc;
  const mmnas_attflat_side* sides[2] = {&hd->sx, &hd->sy};
  for (int k = 0; k < 2; ++k)
    MMNAS_REQUIRE(sides[k]->x && sides[k]->W1 && sides[k]->W2 && sides[k]->Wm, MMNAS_E_ARG, "head_fwd: side %d null pointer", k);
  auto side_stream = [&](int k) -> void* { return (oc && k == 0) ? (void*)oc->enc : stream; };
  // h = drop(relu(x W1^T + b1))                                   (FC, modules.py:13-31; own dropout seed per side)
  if (!oc) {   // both sides as the two groups of one launch (own weights, own dropout stream per group)
    gemm_init(g, MMNAS_GEMM_NT, MID, d, d, d, MID);
    g.ngroups = 2; g.relu = 1;
    if (drop) { g.drop_p = hd->drop_p; g.drop_seed = sides[0]->seed; g.drop_site = 0; }
    for (int k = 0; k < 2; ++k) {
      const mmnas_attflat_side& sd = *sides[k];
      g.g[k].M = head_rows(hd, sd); g.g[k].A[0] = sd.x; g.g[k].B[0] = sd.W1; g.g[k].bias = sd.b1; g.g[k].C = L.s[k].h;
      if (drop) g.g[k].drop_seed = sd.seed;
    }
    if ((rc = mmnas_gemm(&g, stream))) return rc;
  }
  for (int k = 0; k < 2 && oc; ++k) {
    const mmnas_attflat_side& sd = *sides[k];
    gemm_init(g, MMNAS_GEMM_NT, MID, d, d, d, MID);
    g.g[0].M = head_rows(hd, sd); g.g[0].A[0] = sd.x; g.g[0].B[0] = sd.W1; g.g[0].bias = sd.b1; g.g[0].C = L.s[k].h; g.relu = 1;
    if (drop) { g.drop_p = hd->drop_p; g.drop_seed = sd.seed; g.drop_site = 0; }
    if ((rc = mmnas_gemm(&g, side_stream(k)))) return rc;
  }
  // glimpse logits = h W2^T + b2                                   (MLP.linear, modules.py:34-41)
  const bool g1 = G == 1 && !oc && glimpse1_supported(MID) && head_glimpse1_on();
  if (g1) {   // one glimpse: a matrix-vector product per side, both in one launch (head.hip)
    if ((rc = glimpse1_fwd(L.s[0].h, hd->sx.W2, hd->sx.b2, L.s[0].logit, (long)head_rows(hd, hd->sx), L.s[1].h, hd->sy.W2, hd->sy.b2,
                           L.s[1].logit, (long)head_rows(hd, hd->sy), MID, st)))
      return rc;
  }
  gemm_init(g, MMNAS_GEMM_NT, G, MID, MID, MID, G);
  for (int k = 0; k < 2 && !g1; ++k) {
    const mmnas_attflat_side& sd = *sides[k];
    mmnas_gemm_group& gg = g.g[oc ? 0 : k];
    gg.M = head_rows(hd, sd); gg.A[0] = L.s[k].h; gg.B[0] = sd.W2; gg.bias = sd.b2; gg.C = L.s[k].logit;
    if (oc && (rc = mmnas_gemm(&g, side_stream(k)))) return rc;
  }
  if (!oc && !g1) { g.ngroups = 2; if ((rc = mmnas_gemm(&g, stream))) return rc; }
  for (int k = 0; k < 2; ++k) {
    const mmnas_attflat_side& sd = *sides[k];
    const HeadSideLayout& s = L.s[k];
    void* const ks = side_stream(k);
    // masked softmax over the sequence + weighted sum                (modules.py:78-84)
    if ((rc = sd.off ? attflat_pool_fwd_packed(s.logit, sd.x, s.probs, s.pooled, B, sd.S, d, G, sd.off, (hipStream_t)ks)
                     : mmnas_attflat_pool_fwd(s.logit, sd.x, sd.mask, s.probs, s.pooled, B, sd.S, d, G, ks))) return rc;
    // merge; the image side adds the language side's result (x_out + y_out, hygr_vqa.py:116)
    gemm_init(g, MMNAS_GEMM_NT, OUT, G * d, G * d, G * d, OUT);
    g.g[0].M = B; g.g[0].A[0] = s.pooled; g.g[0].B[0] = sd.Wm; g.g[0].bias = sd.bm; g.g[0].C = k ? L.sum : L.xo;
    if (k) { g.g[0].residual = L.xo; g.ldres = OUT; }
    if (k && oc && (rc = ev_fork(oc->enc, st, oc->ovl[1]))) return rc;   // the language side's result exists from here on
    if ((rc = mmnas_gemm(&g, ks))) return rc;
  }
  if ((rc = mmnas_layernorm_fwd(L.sum, hd->ln_a, hd->ln_b, L.xy, B, OUT, hd->eps, stream))) return rc;
  if (L.g1proj && head_glimpse1_on())
    return glimpse1_fwd(L.xy, hd->Wp, hd->bp, hd->logits, B, nullptr, nullptr, nullptr, nullptr, 0, OUT, st);
  gemm_init(g, MMNAS_GEMM_NT, hd->ANS, OUT, OUT, OUT, hd->ANS);
  g.g[0].M = B; g.g[0].A[0] = L.xy; g.g[0].B[0] = hd->Wp; g.g[0].bias = hd->bp; g.g[0].C = hd->logits;
  return mmnas_gemm(&g, stream);
}

extern "C" int mmnas_head_bwd(const mmnas_head* hd, void* stream) {
  int rc = head_check(hd, "head_bwd");
  if (rc) return rc;
  MMNAS_REQUIRE(hd->arena && hd->dlogits && hd->ln_a && hd->dln_a && hd->dln_b && hd->Wp && hd->dWp, MMNAS_E_ARG, "head_bwd: null pointer");
  HeadLayout L = head_layout(hd);
  const bool drop = (hd->flags & MMNAS_F_TRAIN) && hd->drop_p > 0.f;
  const float gate_scale = drop ? 1.0f / (1.0f - hd->drop_p) : 1.0f;
  const int B = hd->B, d = hd->d, MID = hd->MID, G = hd->G, OUT = hd->OUT, ANS = hd->ANS;
  hipStream_t st = (hipStream_t)stream;
  mmnas_gemm_desc g, w;
  // answer projection
  if (L.g1proj && head_glimpse1_on()) {
    AuxReduce pr;
    if ((rc = glimpse1_bwd(hd->dlogits, L.xy, hd->Wp, 1.0f, 0, L.dxy, nullptr, hd->dWp, L.g1proj, B, OUT, st, &pr))) return rc;
    if ((rc = launch_aux_reduce(pr, st))) return rc;
  } else if (L.dlogT && head_projT_on()) {
    if ((rc = transpose2d(hd->dlogits, L.dlogT, B, ANS, st))) return rc;
    gemm_init(g, MMNAS_GEMM_TN, OUT, ANS, B, OUT, OUT);            // dxy [B, OUT] = dlogT^T [B, ANS] Wp [ANS, OUT]
    g.g[0].M = B; g.g[0].A[0] = L.dlogT; g.g[0].B[0] = hd->Wp; g.g[0].C = L.dxy;
    if ((rc = mmnas_gemm(&g, stream))) return rc;
    gemm_init(w, MMNAS_GEMM_NN, OUT, B, B, OUT, OUT);              // dWp [ANS, OUT] += dlogT [ANS, B] xy [B, OUT]
    w.g[0].M = ANS; w.g[0].A[0] = L.dlogT; w.g[0].B[0] = L.xy; w.g[0].C = hd->dWp; w.accumulate = 1;
    if ((rc = mmnas_gemm(&w, stream))) return rc;
  } else {
    gemm_init(w, MMNAS_GEMM_TN, OUT, B, ANS, OUT, OUT);
    w.g[0].M = ANS; w.g[0].A[0] = hd->dlogits; w.g[0].B[0] = L.xy; w.g[0].C = hd->dWp; w.accumulate = 1;
    gemm_init(g, MMNAS_GEMM_NN, OUT, ANS, ANS, OUT, OUT);
    g.g[0].M = B; g.g[0].A[0] = hd->dlogits; g.g[0].B[0] = hd->Wp; g.g[0].C = L.dxy;
    if ((rc = mmnas_gemm_pair(&g, &w, stream))) return rc;
  }
  if (hd->dbp && (rc = mmnas_colsum(hd->dlogits, hd->dbp, B, ANS, ANS, stream))) return rc;
  // proj_norm
  AuxReduce lnred;
  lnred.part = nullptr;
  if ((rc = layernorm_bwd_deferred(L.sum, hd->ln_a, L.dxy, L.dsum, hd->dln_a, hd->dln_b, nullptr, nullptr, L.lnws, 0.f, 0, 0, B, OUT,
                                   hd->eps, st, &lnred)))
    return rc;
  SideCtx* oc = head_overlap_on() ? side_ctx(st, true) : nullptr;
  if (oc && (rc = ev_fork(st, oc->enc, oc->ovl[2]))) return rc;   // the sum's gradient is complete: both sides start
  for (int k = 1; k >= 0; --k) {
    const mmnas_attflat_side& sd = k ? hd->sy : hd->sx;
    const HeadSideLayout& s = L.s[k];
    MMNAS_REQUIRE(sd.x && sd.dx && sd.dW1 && sd.dW2 && sd.dWm, MMNAS_E_ARG, "head_bwd: side %d null pointer", k);
    hipStream_t st = (oc && k == 0) ? oc->enc : (hipStream_t)stream;   // (shadows: the language side on the second stream)
    void* const stream = (void*)st;
    const int M = head_rows(hd, sd);
    // merge: the sum's gradient reaches both sides unchanged
    gemm_init(w, MMNAS_GEMM_TN, G * d, B, OUT, G * d, G * d);
    w.g[0].M = OUT; w.g[0].A[0] = L.dsum; w.g[0].B[0] = s.pooled; w.g[0].C = sd.dWm; w.accumulate = 1;
    gemm_init(g, MMNAS_GEMM_NN, G * d, OUT, OUT, G * d, G * d);
    g.g[0].M = B; g.g[0].A[0] = L.dsum; g.g[0].B[0] = sd.Wm; g.g[0].C = s.dpooled;
    if ((rc = gemm_pair_aux(&g, &w, k ? &lnred : nullptr, st))) return rc;
    if (sd.dbm && (rc = mmnas_colsum(L.dsum, sd.dbm, B, OUT, OUT, stream))) return rc;
    // pooling
    if ((rc = sd.off ? attflat_pool_bwd_packed(s.probs, sd.x, s.dpooled, s.dlog, s.dxpool, B, sd.S, d, G, sd.off, st)
                     : mmnas_attflat_pool_bwd(s.probs, sd.x, sd.mask, s.dpooled, s.dlog, s.dxpool, B, sd.S, d, G, stream))) return rc;
    // glimpse-logit linear: dh = dlog W2 with relu' and the dropout replay from h; db1 rides as column sums of dh
    AuxReduce g1red;
    g1red.part = nullptr;
    if (G == 1 && glimpse1_supported(MID) && head_glimpse1_on()) {   // one glimpse: an outer product + three reductions, one pass over h
      if ((rc = glimpse1_bwd(s.dlog, s.h, sd.W2, gate_scale, 1, s.dh, sd.db1, sd.dW2, s.g1part, M, MID, st, &g1red))) return rc;
      if (sd.db2 && (rc = mmnas_colsum(s.dlog, sd.db2, M, G, G, stream))) return rc;
    } else {
    gemm_init(w, MMNAS_GEMM_TN, MID, M, G, MID, MID);
    w.g[0].M = G; w.g[0].A[0] = s.dlog; w.g[0].B[0] = s.h; w.g[0].C = sd.dW2; w.accumulate = 1;
    gemm_init(g, MMNAS_GEMM_NN, MID, G, G, MID, MID);
    g.g[0].M = M; g.g[0].A[0] = s.dlog; g.g[0].B[0] = sd.W2; g.g[0].C = s.dh;
    g.g[0].gate = s.h; g.ldgate = MID; g.gate_scale = gate_scale;
    g.g[0].colsum = sd.db1;
    if ((rc = mmnas_gemm_pair(&g, &w, stream))) return rc;
    if (sd.db2 && (rc = mmnas_colsum(s.dlog, sd.db2, M, G, G, stream))) return rc;
    }
    // FC: dx = dh W1 + the pooling path's share
    gemm_init(w, MMNAS_GEMM_TN, d, M, MID, d, d);
    w.g[0].M = MID; w.g[0].A[0] = s.dh; w.g[0].B[0] = sd.x; w.g[0].C = sd.dW1; w.accumulate = 1;
    gemm_init(g, MMNAS_GEMM_NN, d, MID, MID, d, d);
    g.g[0].M = M; g.g[0].A[0] = s.dh; g.g[0].B[0] = sd.W1; g.g[0].C = sd.dx;
    g.g[0].residual = s.dxpool; g.ldres = d;
    if ((rc = gemm_pair_aux(&g, &w, g1red.part ? &g1red : nullptr, st))) return rc;
  }
  if (oc && (rc = ev_fork(oc->enc, st, oc->ovl[3]))) return rc;
  return MMNAS_OK;
}

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
  L.Mq = (size_t)op->B * op->Sq; L.Mk = (size_t)op->B * op->Sk;
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

static int att_check(const mmnas_att_op* op, const char* who) {
  MMNAS_REQUIRE(op, MMNAS_E_ARG, "%s: null descriptor", who);
  MMNAS_REQUIRE(op->B > 0 && op->Sq > 0 && op->Sk > 0 && op->d > 0 && op->di > 0, MMNAS_E_SHAPE,
                "%s: B=%d Sq=%d Sk=%d d=%d di=%d", who, op->B, op->Sq, op->Sk, op->d, op->di);
  MMNAS_REQUIRE(op->H * op->dh == op->di, MMNAS_E_SHAPE, "%s: H*dh=%d*%d != di=%d", who, op->H, op->dh, op->di);
  MMNAS_REQUIRE(op->d % 4 == 0 && op->di % 4 == 0, MMNAS_E_SHAPE, "%s: d=%d di=%d must be multiples of 4", who,
                op->d, op->di);
  if (op->flags & MMNAS_F_SELF) MMNAS_REQUIRE(op->Sq == op->Sk, MMNAS_E_SHAPE, "%s: SELF needs Sq == Sk", who);
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

extern "C" int mmnas_att_op_fwd(const mmnas_att_op* op, void* stream) {
  int rc = att_check(op, "att_op_fwd");
  if (rc) return rc;
  MMNAS_REQUIRE(op->xq && op->xkv && op->Wq && op->Wk && op->Wv && op->Wm && op->y && op->save, MMNAS_E_ARG,
                "att_op_fwd: null pointer");
  const int fl = op->flags;
  const bool norm = fl & MMNAS_F_NORM, rel = fl & MMNAS_F_REL;
  const bool drop = (fl & MMNAS_F_TRAIN) && op->drop_p > 0.f;
  if (norm) MMNAS_REQUIRE(op->ln_a && op->ln_b, MMNAS_E_ARG, "att_op_fwd: NORM without ln parameters");
  if (rel) MMNAS_REQUIRE(op->rel && op->Wr && op->br, MMNAS_E_ARG, "att_op_fwd: REL without rel/Wr/br");
  if (fl & MMNAS_F_MASK) MMNAS_REQUIRE(op->mask, MMNAS_E_ARG, "att_op_fwd: MASK without mask");
  AttLayout L = att_layout(op);
  const int d = op->d, di = op->di;

  mmnas_gemm_desc g;
  gemm_init(g, MMNAS_GEMM_NT, di, d, d, d, di);
  g.ngroups = 3;
  g.g[0].M = (int)L.Mq; g.g[0].A[0] = op->xq;  g.g[0].B[0] = op->Wq; g.g[0].C = L.Q;
  g.g[1].M = (int)L.Mk; g.g[1].A[0] = op->xkv; g.g[1].B[0] = op->Wk; g.g[1].C = L.K;
  g.g[2].M = (int)L.Mk; g.g[2].A[0] = op->xkv; g.g[2].B[0] = op->Wv; g.g[2].C = L.V;
  if ((rc = mmnas_gemm(&g, stream))) return rc;

  if (rel) {
    if (fl & MMNAS_F_RELRAW) {  // lazy handle: bias straight from the raw [B,Sq,Sk,C] relations
      MMNAS_REQUIRE(op->Wy && op->by, MMNAS_E_ARG, "att_op_fwd: RELRAW without Wy/by");
      rc = mmnas_rel_fused_fwd(op->rel, op->Wy, op->by, op->Wr, op->br, L.biasT, op->B, op->Sq, op->Sk, op->C, op->R,
                               op->H, stream);
    } else {
      rc = mmnas_rel_bias_fwd(op->rel, op->Wr, op->br, L.biasT, op->B, op->Sq, op->Sk, op->R, op->H, stream);
    }
    if (rc) return rc;
  }

  mmnas_mha_desc m;
  memset(&m, 0, sizeof(m));
  m.B = op->B; m.H = op->H; m.Sq = op->Sq; m.Sk = op->Sk; m.dh = op->dh;
  m.ldq = m.ldk = m.ldv = m.ldo = di;
  m.Q = L.Q; m.K = L.K; m.V = L.V; m.mask = (fl & MMNAS_F_MASK) ? op->mask : nullptr; m.biasT = L.biasT;
  m.O = L.att; m.lse = L.stats;
  m.drop_p = drop ? op->drop_p : 0.f; m.drop_site = 0; m.drop_seed = op->seed;
  if ((rc = mmnas_mha_core_fwd(&m, stream))) return rc;

  gemm_init(g, MMNAS_GEMM_NT, d, di, di, di, d);
  g.g[0].M = (int)L.Mq; g.g[0].A[0] = L.att; g.g[0].B[0] = op->Wm; g.g[0].C = norm ? L.z : op->y;
  if (fl & MMNAS_F_RESIDUAL) { g.g[0].residual = op->xq; g.ldres = d; }
  if (drop) { g.drop_p = op->drop_p; g.drop_site = 1; g.drop_seed = op->seed; }
  if ((rc = mmnas_gemm(&g, stream))) return rc;

  if (norm) return mmnas_layernorm_fwd(L.z, op->ln_a, op->ln_b, op->y, (int)L.Mq, d, op->eps, stream);
  return MMNAS_OK;
}

extern "C" int mmnas_att_op_bwd(const mmnas_att_op* op, void* stream) {
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

  // 1. through LayerNorm and the output dropout
  const float* dz = op->dy;   // gradient wrt z = x + drop(core)
  const float* dt = op->dy;   // gradient wrt core
  AuxReduce lnred;   // the LayerNorm parameter-gradient reduction rides on the first gradient-pair launch below
  lnred.part = nullptr;
  if (norm) {
    if ((rc = layernorm_bwd_deferred(L.z, op->ln_a, op->dy, L.dz, op->dln_a, op->dln_b, drop ? L.dt : nullptr, nullptr,
                                     L.lnws, drop ? op->drop_p : 0.f, op->seed, 1, Mq, d, op->eps, (hipStream_t)stream, &lnred)))
      return rc;
    dz = L.dz; dt = drop ? L.dt : L.dz;
  } else if (drop) {
    if ((rc = mmnas_drop_add(op->dy, nullptr, L.dt, (size_t)Mq * d, op->drop_p, op->seed, 1, stream))) return rc;
    dt = L.dt;
  }

  mmnas_gemm_desc g, w;
  // 2. d(att) = dt Wm            [Mq,d] x [d,di]
  gemm_init(g, MMNAS_GEMM_NN, di, d, d, di, di);
  g.g[0].M = Mq; g.g[0].A[0] = dt; g.g[0].B[0] = op->Wm; g.g[0].C = L.datt;
  // 3. dWm += dt^T att           [d,di], reduction over the Mq rows (same launch: mmnas_gemm_pair)
  gemm_init(w, MMNAS_GEMM_TN, di, Mq, d, di, di);
  w.g[0].M = d; w.g[0].A[0] = dt; w.g[0].B[0] = L.att; w.g[0].C = op->dWm;
  w.accumulate = 1;
  if ((rc = gemm_pair_aux(&g, &w, &lnred, (hipStream_t)stream))) return rc;

  // 4. attention core backward
  mmnas_mha_desc m;
  memset(&m, 0, sizeof(m));
  m.B = op->B; m.H = op->H; m.Sq = op->Sq; m.Sk = op->Sk; m.dh = op->dh;
  m.ldq = m.ldk = m.ldv = m.ldo = di;
  m.Q = L.Q; m.K = L.K; m.V = L.V; m.mask = (fl & MMNAS_F_MASK) ? op->mask : nullptr; m.biasT = L.biasT;
  m.O = L.att; m.lse = L.stats;
  m.drop_p = drop ? op->drop_p : 0.f; m.drop_site = 0; m.drop_seed = op->seed;
  m.dO = L.datt; m.dQ = L.dQ; m.dK = L.dK; m.dV = L.dV; m.dbiasT = L.dbiasT; m.delta = L.delta;
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
    if ((rc = mmnas_gemm_pair(&g, &w, stream))) return rc;
  } else {
    gemm_init(w, MMNAS_GEMM_TN, d, Mq, di, d, d);
    w.g[0].M = di; w.g[0].A[0] = L.dQ; w.g[0].B[0] = op->xq; w.g[0].C = op->dWq;
    w.accumulate = 1;
    gemm_init(g, MMNAS_GEMM_NN, d, di, di, d, d);
    g.g[0].M = Mq; g.g[0].C = op->dxq; g.g[0].A[0] = L.dQ; g.g[0].B[0] = op->Wq;
    if (fl & MMNAS_F_RESIDUAL) { g.g[0].residual = dz; g.ldres = d; }
    if ((rc = mmnas_gemm_pair(&g, &w, stream))) return rc;
    gemm_init(w, MMNAS_GEMM_TN, d, Mk, di, d, d);
    w.ngroups = 2;
    w.g[0].M = di; w.g[0].A[0] = L.dK; w.g[0].B[0] = op->xkv; w.g[0].C = op->dWk;
    w.g[1].M = di; w.g[1].A[0] = L.dV; w.g[1].B[0] = op->xkv; w.g[1].C = op->dWv;
    w.accumulate = 1;
    gemm_init(g, MMNAS_GEMM_NN, d, di, di, d, d);
    g.nseg = 2;
    g.g[0].M = Mk; g.g[0].C = op->dxkv;
    g.g[0].A[0] = L.dK; g.g[0].B[0] = op->Wk;
    g.g[0].A[1] = L.dV; g.g[0].B[1] = op->Wv;
    if ((rc = mmnas_gemm_pair(&g, &w, stream))) return rc;
  }

  // 7. relation bias
  if (rel && (fl & MMNAS_F_RELRAW)) {
    MMNAS_REQUIRE(op->Wy && op->by && op->dWy && op->dby, MMNAS_E_ARG, "att_op_bwd: RELRAW gradients missing");
    return mmnas_rel_fused_bwd(op->rel, op->Wy, op->by, op->Wr, op->br, L.dbiasT, op->dWy, op->dby, op->dWr, op->dbr,
                               L.relws, op->B, op->Sq, op->Sk, op->C, op->R, op->H, stream);
  }
  if (rel)
    return mmnas_rel_bias_bwd(op->rel, op->Wr, op->br, L.dbiasT, op->drel, op->dWr, op->dbr, 0, op->B, op->Sq,
                              op->Sk, op->R, op->H, stream);
  return MMNAS_OK;
}

// ------------------------------------------------------------------------------------------ MLP
namespace mmnas {

struct MlpLayout {
  float* h[3];      // h[i] = input of layer i (h[0] = x, not stored); saved for i >= 1
  float* z;
  float *dz, *dt, *dp[2], *lnws;
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

extern "C" int mmnas_mlp_op_fwd(const mmnas_mlp_op* op, void* stream) {
  int rc = mlp_check(op, "mlp_op_fwd");
  if (rc) return rc;
  MMNAS_REQUIRE(op->x && op->y && op->save, MMNAS_E_ARG, "mlp_op_fwd: null pointer");
  const int fl = op->flags;
  const bool norm = fl & MMNAS_F_NORM;
  const bool drop = (fl & MMNAS_F_TRAIN) && op->drop_p > 0.f;
  if (norm) MMNAS_REQUIRE(op->ln_a && op->ln_b, MMNAS_E_ARG, "mlp_op_fwd: NORM without ln parameters");
  MlpLayout L = mlp_layout(op);
  const int d = op->dims[0];
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
    if ((rc = mmnas_gemm(&g, stream))) return rc;
    in = L.h[i + 1];
  }
  if (norm) return mmnas_layernorm_fwd(L.z, op->ln_a, op->ln_b, op->y, op->M, d, op->eps, stream);
  return MMNAS_OK;
}

extern "C" int mmnas_mlp_op_bwd(const mmnas_mlp_op* op, void* stream) {
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
    if ((rc = layernorm_bwd_deferred(L.z, op->ln_a, op->dy, L.dz, op->dln_a, op->dln_b, drop ? L.dt : nullptr, dcol,
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
  for (int i = nl - 1; i >= 0; --i) {
    const float* hin = i == 0 ? op->x : L.h[i];
    const int nout = op->dims[i + 1], nin = op->dims[i];
    // bias gradient: column sums of dpre.  The last layer's ride on the LayerNorm backward when possible; a hidden
    // layer's are accumulated by the epilogue of the data-gradient product that writes its dpre (below)
    if (op->db[i] && i == nl - 1 && !last_bias_done)
      if ((rc = mmnas_colsum(dpre, op->db[i], M, nout, nout, stream))) return rc;
    // weight gradient dW_i[nout,nin] += dpre^T hin and data gradient, one launch
    mmnas_gemm_desc w;
    gemm_init(w, MMNAS_GEMM_TN, nin, M, nout, nin, nin);
    w.g[0].M = nout; w.g[0].A[0] = dpre; w.g[0].B[0] = hin; w.g[0].C = op->dW[i];
    w.accumulate = 1;
    gemm_init(g, MMNAS_GEMM_NN, nin, nout, nout, nin, nin);
    g.g[0].M = M; g.g[0].A[0] = dpre; g.g[0].B[0] = op->W[i];
    if (i == 0) {
      g.g[0].C = op->dx;
      if (fl & MMNAS_F_RESIDUAL) { g.g[0].residual = dz; g.ldres = d; }
      if ((rc = gemm_pair_aux(&g, &w, i == nl - 1 ? &lnred : nullptr, (hipStream_t)stream))) return rc;
    } else {
      float* out = L.dp[(nl - 1 - i) & 1];
      g.g[0].C = out;
      g.g[0].gate = L.h[i]; g.ldgate = nin; g.gate_scale = gate_scale;  // relu' and dropout replay from h_i
      g.g[0].colsum = op->db[i - 1];   // db_{i-1} += column sums of dpre_{i-1} (may be NULL: layer without bias)
      if ((rc = gemm_pair_aux(&g, &w, i == nl - 1 ? &lnred : nullptr, (hipStream_t)stream))) return rc;
      dpre = out;
    }
  }
  return MMNAS_OK;
}

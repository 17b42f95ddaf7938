"""Autograd bridge between PyTorch tensors (device memory, streams: plumbing) and the C ABI of
libmmnas_hip.so (the product).  One ctypes call per operator forward and one per backward; every
buffer the kernels touch is allocated here through torch's caching allocator and handed over as a
raw device pointer together with the current HIP stream.
"""
import ctypes as C
import os
import threading

import torch

from . import _lib as L

# ------------------------------------------------------------------------------------------
# dropout seeds: every operator call in training mode draws a fresh 64-bit seed; kernels derive
# each keep decision from (seed, site, element index) -- see csrc/rng.h
# ------------------------------------------------------------------------------------------
_seed_lock = threading.Lock()
_seed_state = {'base': None, 'counter': 0}


def manual_seed(seed):
    """Re-seed the dropout stream (also done lazily from torch.initial_seed())."""
    with _seed_lock:
        _seed_state['base'] = int(seed) & 0xFFFFFFFFFFFFFFFF
        _seed_state['counter'] = 0


def next_seed():
    with _seed_lock:
        if _seed_state['base'] is None:
            _seed_state['base'] = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        _seed_state['counter'] += 1
        x = (_seed_state['base'] * 0x9E3779B97F4A7C15 + _seed_state['counter'] * 0xD1B54A32D192ED03)
        return x & 0xFFFFFFFFFFFFFFFF


_plan_cache = {}
_sinks_on = [True]


def runtime_config():
    """What this process runs the library with: where the shared object came from, the launch configuration the library
    established at import (`_lib._launch_configuration`: HIP_FORCE_DEV_KERNARG and who set it) and every MMNAS_* / HIP_* /
    NCCL_* / RCCL_* switch of the environment.  bench.py records it; an integrator prints it once."""
    return {
        'lib_path': L.LIB_PATH,
        'lib_loaded': L._lib is not None,
        'abi_version': L.lib().mmnas_abi_version() if os.path.exists(L.LIB_PATH) else None,
        'hip_force_dev_kernarg': os.environ.get('HIP_FORCE_DEV_KERNARG'),
        'hip_force_dev_kernarg_source': L.KERNARG_SOURCE,
        'hip_initialised': torch.cuda.is_initialized(),
        'torch': torch.__version__, 'hip': getattr(torch.version, 'hip', None),
        'env': {k: v for k, v in sorted(os.environ.items()) if k.startswith(('MMNAS_', 'HIP_', 'NCCL_', 'RCCL_', 'HSA_'))},
    }


class no_sinks:
    """Context manager: inside it the backward kernels hand every parameter gradient to autograd as a tensor instead
    of adding it straight into the flat gradient buffer.  Needed around `torch.autograd.grad(loss, params)`: a sink
    gives autograd None for the parameter, which autograd.grad would report as an unused / zero gradient."""

    def __enter__(self):
        self._old = _sinks_on[0]
        _sinks_on[0] = False

    def __exit__(self, *a):
        _sinks_on[0] = self._old



def _grad_bufs(params, dev):
    """Buffers the backward kernels accumulate parameter gradients into, one per entry of `params`
    (None entries stay None).  A parameter whose `.grad` is a view of a flat gradient buffer
    (dp.FlatGrads.enable_sinks) gets that view -- the kernels then add straight into the buffer that
    is all-reduced / fed to the fused optimizer, and autograd receives None for it (no zero-fill, no
    accumulate kernel).  Everything else comes out of ONE freshly zeroed allocation.
    Returns (bufs, rets, sinks): what to hand to the kernel, what to return to autograd, sinks to notify."""
    bufs, rets, sinks, fresh = [], [], [], []
    for t in params:
        if t is None:
            bufs.append(None); rets.append(None)
            continue
        s = getattr(t, '_mmnas_sink', None)
        if s is not None and t.grad is s.view and _sinks_on[0]:
            bufs.append(s.view); rets.append(None); sinks.append(s)
        else:
            bufs.append(t); rets.append(t); fresh.append(len(bufs) - 1)
    if fresh:
        sizes = [bufs[i].numel() for i in fresh]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        for i, part in zip(fresh, torch.split(flat, sizes)):
            g = part.view(bufs[i].shape)
            bufs[i] = g
            rets[i] = g
    return bufs, rets, sinks


def _bytes(n, dev):
    return torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)


def _f32c(t):
    if t.dtype != torch.float32:
        raise L.MMNasHipError('mmnas_amd operators compute in float32; got %s' % t.dtype)
    return t.contiguous()


def _mask_u8(mask, B, Sk):
    """bool [B,1,1,Sk] (hygr_vqa.py:121-122) -> uint8 [B,Sk] sharing storage."""
    if mask is None:
        return None
    m = mask.reshape(B, Sk)
    if m.dtype == torch.bool:
        m = m.contiguous().view(torch.uint8)
    elif m.dtype != torch.uint8:
        m = (m != 0).to(torch.uint8)
    return m.contiguous()


# ------------------------------------------------------------------------------------------
# attention family
# ------------------------------------------------------------------------------------------
class AttentionOp(torch.autograd.Function):
    """SelfAtt / RelSelfAtt / GuidedAtt / UniimgAtt forward+backward (modules.py:248-325,403-428)."""

    @staticmethod
    def forward(ctx, xq, xkv, mask, rel, Wq, Wk, Wv, Wm, Wr, br, ln_a, ln_b, Wy, by, dh, norm, residual, drop_p,
                training, seed, eps):
        lib = L.lib()
        self_att = xkv is None or xkv is xq
        xq = _f32c(xq)
        xkv = xq if self_att else _f32c(xkv)
        B, Sq, d = xq.shape
        Sk = xkv.shape[1]
        di = Wq.shape[0]
        H = di // dh
        dev = xq.device
        flags = (L.F_NORM if norm else 0) | (L.F_RESIDUAL if residual else 0)
        if self_att:
            flags |= L.F_SELF
        m8 = _mask_u8(mask, B, Sk)
        if m8 is not None:
            flags |= L.F_MASK
        lazy = rel is not None and Wy is not None   # lazy relation handle: rel is the RAW [B,Sq,Sk,C] tensor
        if rel is not None:
            rel = _f32c(rel)
            want = (B, Sq, Sk, Wy.shape[1] if lazy else Wr.shape[1])
            if tuple(rel.shape) != want:
                raise L.MMNasHipError('rel_embed shape %s, expected %s' % (tuple(rel.shape), want))
            flags |= L.F_REL | (L.F_RELRAW if lazy else 0)
        if training and drop_p > 0:
            flags |= L.F_TRAIN
        op = L.AttOp()
        op.B, op.Sq, op.Sk, op.d, op.di, op.H, op.dh = B, Sq, Sk, d, di, H, dh
        op.R = Wr.shape[1] if rel is not None else 0
        op.flags, op.drop_p, op.eps, op.seed = flags, float(drop_p), float(eps), int(seed)
        key = ('att', B, Sq, Sk, d, di, H, op.R, flags)  # flags carry RELRAW, which changes the scratch size
        plan = _plan_cache.get(key)
        if plan is None:
            p = L.Plan()
            L.check(lib.mmnas_att_op_plan(C.byref(op), C.byref(p)))
            plan = (p.save_bytes, p.ws_bwd_bytes)
            _plan_cache[key] = plan
        y = torch.empty_like(xq)
        save = _bytes(plan[0], dev)
        Wq, Wk, Wv, Wm = _f32c(Wq), _f32c(Wk), _f32c(Wv), _f32c(Wm)
        op.xq, op.xkv, op.mask, op.rel = L.fptr(xq), L.fptr(xkv), L.ptr(m8), L.fptr(rel)
        op.Wq, op.Wk, op.Wv, op.Wm = L.fptr(Wq), L.fptr(Wk), L.fptr(Wv), L.fptr(Wm)
        if rel is not None:
            Wr, br = _f32c(Wr), _f32c(br)
            op.Wr, op.br = L.fptr(Wr), L.fptr(br)
        if lazy:
            Wy, by = _f32c(Wy), _f32c(by)
            op.C, op.Wy, op.by = Wy.shape[1], L.fptr(Wy), L.fptr(by)
        if norm:
            ln_a, ln_b = _f32c(ln_a), _f32c(ln_b)
            op.ln_a, op.ln_b = L.fptr(ln_a), L.fptr(ln_b)
        op.y, op.save = L.fptr(y), L.ptr(save)
        L.check(lib.mmnas_att_op_fwd(C.byref(op), L.stream()))
        ctx.op = op
        ctx.plan = plan
        ctx.self_att = self_att
        ctx.has_rel = rel is not None
        ctx.lazy = lazy
        ctx.norm = norm
        ctx.ln_b_param = ln_b if norm else None
        ctx.keep = (xq, xkv, m8, rel, Wq, Wk, Wv, Wm, Wr if rel is not None else None,
                    br if rel is not None else None, ln_a if norm else None, save,
                    Wy if lazy else None, by if lazy else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.keep is None:
            raise RuntimeError('AttentionOp: backward ran a second time -- the saved block is released after the first '
                               'backward (retain_graph / double backward are not supported by the HIP operators)')
        lib = L.lib()
        op = ctx.op
        xq, xkv, m8, rel, Wq, Wk, Wv, Wm, Wr, br, ln_a, save, Wy, by = ctx.keep
        dev = xq.device
        dy = _f32c(dy)
        # Wy/by (linear_y_rel) are shared by every relation operator of a net; the kernels ADD their contribution, so
        # a sink works for them too (the reducer waits for autograd's post-accumulate hook, which fires once after
        # the last use)
        bufs, rets, sinks = _grad_bufs([Wq, Wk, Wv, Wm, Wr, br, ln_a, ctx.ln_b_param] + ([Wy, by] if ctx.lazy else [None, None]), dev)
        dWq, dWk, dWv, dWm, dWr, dbr, dla, dlb, dWy, dby = bufs
        dxq = torch.empty_like(xq)
        dxkv = None if ctx.self_att else torch.empty_like(xkv)
        want_drel = ctx.has_rel and not ctx.lazy and ctx.needs_input_grad[3]
        drel = torch.empty_like(rel) if want_drel else None
        ws = _bytes(ctx.plan[1], dev)
        op.dy, op.dxq, op.dxkv, op.drel = L.fptr(dy), L.fptr(dxq), L.fptr(dxkv), L.fptr(drel)
        op.dWq, op.dWk, op.dWv, op.dWm = L.fptr(dWq), L.fptr(dWk), L.fptr(dWv), L.fptr(dWm)
        op.dWr, op.dbr, op.dln_a, op.dln_b = L.fptr(dWr), L.fptr(dbr), L.fptr(dla), L.fptr(dlb)
        op.dWy, op.dby = L.fptr(dWy), L.fptr(dby)
        op.ws = L.ptr(ws)
        L.check(lib.mmnas_att_op_bwd(C.byref(op), L.stream()))
        ctx.keep = None
        for sk in sinks:
            sk.done()
        return (dxq, dxkv, None, drel) + tuple(rets) + (None, None, None, None, None, None, None)


def attention_op(xq, xkv, mask, rel, Wq, Wk, Wv, Wm, Wr, br, ln_a, ln_b, *, dh, norm, residual, drop_p,
                 training, eps=1e-6, seed=None, rel_Wy=None, rel_by=None):
    """rel is the [B,Sq,Sk,R] relation embedding, or -- with rel_Wy/rel_by (linear_y_rel) given -- the RAW
    [B,Sq,Sk,C] relation tensor of a lazy RelHandle."""
    if seed is None:
        seed = next_seed() if (training and drop_p > 0) else 0
    return AttentionOp.apply(xq, xkv, mask, rel, Wq, Wk, Wv, Wm, Wr, br, ln_a, ln_b, rel_Wy, rel_by, dh, norm,
                             residual, float(drop_p), bool(training), seed, eps)


class MhaCoreFn(torch.autograd.Function):
    """MHAtt.att (modules.py:191-199) on already projected Q [B,Sq,di], K, V [B,Sk,di] (head h = columns
    [h*dh, (h+1)*dh)): mmnas_mha_core_fwd/bwd.  biasT: [B,H,Sk,Sq] additive score bias (RelMHAtt) or None."""

    @staticmethod
    def forward(ctx, Q, K, V, mask, biasT, dh, drop_p, seed):
        lib = L.lib()
        Q, K, V = _f32c(Q), _f32c(K), _f32c(V)
        B, Sq, di = Q.shape
        Sk = K.shape[1]
        H = di // dh
        m8 = _mask_u8(mask, B, Sk)
        biasT = _f32c(biasT) if biasT is not None else None
        O = torch.empty_like(Q)
        lse = torch.empty(B, H, Sq, 2, dtype=torch.float32, device=Q.device)
        d = L.MhaDesc()
        d.B, d.H, d.Sq, d.Sk, d.dh = B, H, Sq, Sk, dh
        d.ldq = d.ldk = d.ldv = d.ldo = di
        d.Q, d.K, d.V, d.mask, d.biasT = L.fptr(Q), L.fptr(K), L.fptr(V), L.ptr(m8), L.fptr(biasT)
        d.O, d.lse = L.fptr(O), L.fptr(lse)
        d.drop_p, d.drop_site, d.drop_seed = float(drop_p), 0, int(seed)
        L.check(lib.mmnas_mha_core_fwd(C.byref(d), L.stream()))
        ctx.desc = d
        ctx.save_for_backward(Q, K, V, O, lse, m8, biasT)
        return O

    @staticmethod
    def backward(ctx, dO):
        lib = L.lib()
        Q, K, V, O, lse, m8, biasT = ctx.saved_tensors
        d = ctx.desc
        dO = _f32c(dO)
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
        dbias = torch.empty_like(biasT) if (biasT is not None and ctx.needs_input_grad[4]) else None
        delta = torch.empty(d.B, d.H, d.Sq, dtype=torch.float32, device=Q.device)
        d.dO, d.dQ, d.dK, d.dV, d.dbiasT, d.delta = L.fptr(dO), L.fptr(dQ), L.fptr(dK), L.fptr(dV), L.fptr(dbias), L.fptr(delta)
        L.check(lib.mmnas_mha_core_bwd(C.byref(d), L.stream()))
        return dQ, dK, dV, None, dbias, None, None, None


def mha_core(Q, K, V, mask, biasT, dh, drop_p=0.0, seed=0):
    return MhaCoreFn.apply(Q, K, V, mask, biasT, int(dh), float(drop_p), int(seed))


class RelBiasFn(torch.autograd.Function):
    """biasT[b,h,k,q] = log(max(relu(rel[b,q,k,:] . Wr[h] + br[h]), 1e-6)) (modules.py:231-235) from a materialised
    relation embedding rel [B,Sq,Sk,R]: mmnas_rel_bias_fwd/bwd."""

    @staticmethod
    def forward(ctx, rel, Wr, br):
        rel, Wr, br = _f32c(rel), _f32c(Wr), _f32c(br)
        B, Sq, Sk, R = rel.shape
        H = Wr.shape[0]
        out = torch.empty(B, H, Sk, Sq, dtype=torch.float32, device=rel.device)
        L.check(L.lib().mmnas_rel_bias_fwd(L.fptr(rel), L.fptr(Wr), L.fptr(br), L.fptr(out), B, Sq, Sk, R, H, L.stream()))
        ctx.save_for_backward(rel, Wr, br)
        return out

    @staticmethod
    def backward(ctx, dbias):
        rel, Wr, br = ctx.saved_tensors
        B, Sq, Sk, R = rel.shape
        H = Wr.shape[0]
        dbias = _f32c(dbias)
        drel = torch.empty_like(rel) if ctx.needs_input_grad[0] else None
        dWr, dbr = torch.zeros_like(Wr), torch.zeros_like(br)
        L.check(L.lib().mmnas_rel_bias_bwd(L.fptr(rel), L.fptr(Wr), L.fptr(br), L.fptr(dbias), L.fptr(drel), L.fptr(dWr),
                                           L.fptr(dbr), 0, B, Sq, Sk, R, H, L.stream()))
        return drel, dWr, dbr


def rel_bias(rel, Wr, br):
    return RelBiasFn.apply(rel, Wr, br)


# ------------------------------------------------------------------------------------------
# MLP family
# ------------------------------------------------------------------------------------------
class MlpOp(torch.autograd.Function):
    """FeedForward / FeedForward_deep forward+backward (modules.py:328-400)."""

    @staticmethod
    def forward(ctx, x, ln_a, ln_b, norm, residual, drop_p, training, seed, eps, *wb):
        lib = L.lib()
        x = _f32c(x)
        nl = len(wb) // 2
        Ws = [_f32c(w) for w in wb[:nl]]
        bs = [(_f32c(b) if b is not None else None) for b in wb[nl:]]
        d = x.shape[-1]
        M = x.numel() // d
        dev = x.device
        flags = (L.F_NORM if norm else 0) | (L.F_RESIDUAL if residual else 0)
        if training and drop_p > 0:
            flags |= L.F_TRAIN
        op = L.MlpOp()
        op.M, op.nl = M, nl
        dims = [d] + [w.shape[0] for w in Ws]
        for i, v in enumerate(dims):
            op.dims[i] = v
        op.flags, op.drop_p, op.eps, op.seed = flags, float(drop_p), float(eps), int(seed)
        key = ('mlp', M, tuple(dims), flags)
        plan = _plan_cache.get(key)
        if plan is None:
            p = L.Plan()
            L.check(lib.mmnas_mlp_op_plan(C.byref(op), C.byref(p)))
            plan = (p.save_bytes, p.ws_bwd_bytes)
            _plan_cache[key] = plan
        y = torch.empty_like(x)
        save = _bytes(plan[0], dev)
        op.x, op.y, op.save = L.fptr(x), L.fptr(y), L.ptr(save)
        for i in range(nl):
            op.W[i] = L.fptr(Ws[i])
            op.b[i] = L.fptr(bs[i])
        if norm:
            ln_a, ln_b = _f32c(ln_a), _f32c(ln_b)
            op.ln_a, op.ln_b = L.fptr(ln_a), L.fptr(ln_b)
        L.check(lib.mmnas_mlp_op_fwd(C.byref(op), L.stream()))
        ctx.op, ctx.plan, ctx.nl, ctx.norm = op, plan, nl, norm
        ctx.ln_b_param = ln_b if norm else None
        ctx.keep = (x, Ws, bs, ln_a if norm else None, save)
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.keep is None:
            raise RuntimeError('MlpOp: backward ran a second time -- the saved block is released after the first '
                               'backward (retain_graph / double backward are not supported by the HIP operators)')
        lib = L.lib()
        op = ctx.op
        x, Ws, bs, ln_a, save = ctx.keep
        dev = x.device
        dy = _f32c(dy)
        nl = ctx.nl
        bufs, rets, sinks = _grad_bufs(list(Ws) + list(bs) + [ln_a, ctx.ln_b_param], dev)
        dWs, dbs = bufs[:nl], bufs[nl:2 * nl]
        dla, dlb = bufs[2 * nl], bufs[2 * nl + 1]
        dx = torch.empty_like(x)
        ws = _bytes(ctx.plan[1], dev)
        op.dy, op.dx, op.ws = L.fptr(dy), L.fptr(dx), L.ptr(ws)
        for i in range(nl):
            op.dW[i] = L.fptr(dWs[i])
            op.db[i] = L.fptr(dbs[i])
        op.dln_a, op.dln_b = L.fptr(dla), L.fptr(dlb)
        L.check(lib.mmnas_mlp_op_bwd(C.byref(op), L.stream()))
        ctx.keep = None
        for sk in sinks:
            sk.done()
        return (dx, rets[2 * nl], rets[2 * nl + 1], None, None, None, None, None, None) + tuple(rets[:2 * nl])


def mlp_op(x, weights, biases, ln_a, ln_b, *, norm, residual, drop_p, training, eps=1e-6, seed=None):
    if seed is None:
        seed = next_seed() if (training and drop_p > 0) else 0
    return MlpOp.apply(x, ln_a, ln_b, norm, residual, float(drop_p), bool(training), seed, eps,
                       *weights, *biases)


# ------------------------------------------------------------------------------------------
# building blocks for the registry-only operators (GLU, convs, activations) and for LayerNorm
# ------------------------------------------------------------------------------------------
def split_planes(w, out=None):
    """mmnas_split_planes: the three bf16 planes [3, *w.shape] of an fp32 weight matrix (w = p0 + p1 + p2 exactly)."""
    w = _f32c(w)
    if out is None:
        out = torch.empty((3,) + tuple(w.shape), dtype=torch.bfloat16, device=w.device)
    L.check(L.lib().mmnas_split_planes(L.fptr(w), L.ptr(out), w.numel(), L.stream()))
    return out


def gemm_desc(layout, groups, N, K, lda, ldb, ldc, nseg=1, relu=False, split_k=1, alpha=1.0, drop=None,
              gate_scale=1.0, ldres=0, ldgate=0, accumulate=False, b_planes=False):
    """mmnas_gemm_desc from Python values.  groups: list of dict(M, A=[..], B=[..], C, bias, residual, gate).
    b_planes: every B is the split_planes() tensor of the weight matrix (layout NT)."""
    g = L.GemmDesc()
    g.layout, g.ngroups, g.nseg, g.N, g.K = layout, len(groups), nseg, N, K
    g.lda, g.ldb, g.ldc, g.ldres, g.ldgate = lda, ldb, ldc, ldres, ldgate
    g.relu, g.split_k, g.alpha, g.gate_scale = int(relu), split_k, alpha, gate_scale
    g.accumulate = int(accumulate)
    g.b_planes = int(b_planes)
    if drop is not None:
        g.drop_p, g.drop_seed, g.drop_site = drop
    for i, grp in enumerate(groups):
        gg = g.g[i]
        gg.M = grp['M']
        for s in range(min(nseg, len(grp['A']))):   # (strided segments: one pointer pair, the descriptor's strides give the rest)
            gg.A[s] = L.fptr(grp['A'][s])
            gg.B[s] = L.ptr(grp['B'][s]) if b_planes else L.fptr(grp['B'][s])
        gg.C = L.fptr(grp['C'])
        gg.bias = L.fptr(grp.get('bias'))
        gg.residual = L.fptr(grp.get('residual'))
        gg.gate = L.fptr(grp.get('gate'))
        gg.colsum = L.fptr(grp.get('colsum'))
    return g


def gemm(*args, **kw):
    """Thin wrapper over mmnas_gemm (arguments of gemm_desc)."""
    g = gemm_desc(*args, **kw)
    L.check(L.lib().mmnas_gemm(C.byref(g), L.stream()))


def gemm_pair(dgrad, wgrad):
    """mmnas_gemm_pair: the data-gradient (NN) and weight-gradient (TN) descriptors of one linear layer, one launch."""
    L.check(L.lib().mmnas_gemm_pair(C.byref(dgrad), C.byref(wgrad), L.stream()))


def _glimpse1_on():
    """MMNAS_HEAD_GLIMPSE1=0: the one-unit linear layers as GEMM launches (A/B, tests) -- read per call, like the native head."""
    return os.environ.get('MMNAS_HEAD_GLIMPSE1', '1') != '0'


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) on the MFMA GEMM (nn.Linear, modules.py:18,115)."""

    @staticmethod
    def forward(ctx, x, W, b, relu):
        ctx.params = (W, b)   # the parameter objects themselves: their gradient sinks are looked up in backward
        x, W = _f32c(x), _f32c(W)
        K = x.shape[-1]
        M = x.numel() // K
        N = W.shape[0]
        y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
        # one output unit (AttFlat's glimpse logits): a matrix-vector product, the kernels the native head uses
        ctx.one = N == 1 and not relu and _glimpse1_on() and bool(L.lib().mmnas_glimpse1_supported(K))
        if ctx.one:
            L.check(L.lib().mmnas_glimpse1_fwd(L.fptr(x), L.fptr(W), L.fptr(_f32c(b) if b is not None else None), L.fptr(y), M, K,
                                               L.stream()))
        else:
            gemm(L.GEMM_NT, [dict(M=M, A=[x], B=[W], C=y, bias=(_f32c(b) if b is not None else None))],
                 N, K, K, K, N, relu=relu)
        ctx.save_for_backward(x, W, y if relu else None)
        ctx.has_bias, ctx.relu = b is not None, relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        dy = _f32c(dy)
        K = x.shape[-1]
        M = x.numel() // K
        N = W.shape[0]
        if ctx.relu:
            dy = torch.ops.aten.threshold_backward(dy, y, 0.0)   # dy * (y > 0), one kernel
        # Both parameter gradients are ADDED into their buffers: the flat gradient buffer's views when the parameter
        # has a sink (no zero-fill, no autograd accumulate kernel), one fresh zeroed allocation otherwise.
        (dW, db), rets, sinks = _grad_bufs(ctx.params, x.device)
        if ctx.one:
            dx = torch.empty_like(x)   # (always formed: the kernel is one pass over x either way)
            ws = torch.empty(L.lib().mmnas_glimpse1_bwd_ws_floats(M, K), dtype=torch.float32, device=x.device)
            L.check(L.lib().mmnas_glimpse1_bwd(L.fptr(dy), L.fptr(x), L.fptr(W), L.fptr(dx), L.fptr(dW),
                                               L.fptr(db if ctx.has_bias else None), L.fptr(ws), M, K, L.stream()))
            for sk in sinks:
                sk.done()
            return (dx if ctx.needs_input_grad[0] else None), rets[0], rets[1], None
        wgrad = gemm_desc(L.GEMM_TN, [dict(M=N, A=[dy], B=[x], C=dW)], K, M, N, K, K, accumulate=True)
        dx = None
        if ctx.needs_input_grad[0]:   # e.g. the relation/region feature inputs of the stem need none
            dx = torch.empty_like(x)
            gemm_pair(gemm_desc(L.GEMM_NN, [dict(M=M, A=[dy], B=[W], C=dx)], K, N, N, K, K), wgrad)
        else:
            L.check(L.lib().mmnas_gemm(C.byref(wgrad), L.stream()))
        if ctx.has_bias:
            L.check(L.lib().mmnas_colsum(L.fptr(dy), L.fptr(db), M, N, N, L.stream()))
        for sk in sinks:
            sk.done()
        return dx, rets[0], rets[1], None


class EmbeddingFn(torch.autograd.Function):
    """nn.Embedding lookup whose backward adds the touched rows straight into the weight's gradient buffer."""

    @staticmethod
    def forward(ctx, idx, weight):
        ctx.weight = weight
        ctx.save_for_backward(idx)
        # data parallel: the ranks' token indices start travelling now (dp.RowExchange), the gradient rows follow in backward
        ctx.rows_key = None
        s = getattr(weight, '_mmnas_sink', None)
        rows = getattr(s.owner, 'row_exchange', None) if s is not None and s.owner is not None else None
        if rows is not None and rows.active and rows.i == s.index and weight.grad is s.view and _sinks_on[0] and ctx.needs_input_grad[1]:
            ctx.rows_key = rows.gather_indices(idx)
        return torch.nn.functional.embedding(idx, weight)

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        w = ctx.weight
        dy = _f32c(dy)
        idx = idx.contiguous()
        (dW,), rets, sinks = _grad_bufs([w], dy.device)
        rows = getattr(sinks[0].owner, 'row_exchange', None) if sinks and sinks[0].owner is not None else None
        if rows is not None and rows.active and rows.i == sinks[0].index:
            # data parallel: the ranks exchange (token index, dy row) pairs -- ~1 MB -- and each adds all of them into its
            # own table gradient, instead of all-reducing the dense 24 MB table (dp.RowExchange)
            rows.exchange(idx, dy, ctx.rows_key)
        else:
            L.check(L.lib().mmnas_embedding_bwd(L.ptr(idx), L.fptr(dy), L.fptr(dW), idx.numel(), w.shape[1], w.shape[0], L.stream()))
        for sk in sinks:
            sk.done()
        return None, rets[0]


def embedding(idx, mod):
    """mod: nn.Embedding without padding_idx / max_norm / sparse gradients (the nets' language stem)."""
    if (idx.is_cuda and idx.dtype == torch.int64 and mod.weight.dtype == torch.float32 and mod.padding_idx is None
            and mod.max_norm is None and not mod.sparse and not mod.scale_grad_by_freq):
        return EmbeddingFn.apply(idx, mod.weight)
    return mod(idx)


class LstmFn(torch.autograd.Function):
    """nn.LSTM(num_layers=1, batch_first=True), zero initial state, full output sequence (hygr_vqa.py:106-107) on
    mmnas_lstm_seq_fwd/bwd: the input projection is one product over all B*T rows, the recurrence ONE persistent
    launch per pass, the parameter / input gradients three products (native gate order: nothing is permuted)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        lib = L.lib()
        ctx.params = (w_ih, w_hh, b_ih, b_hh)   # the parameter objects: their gradient sinks are looked up in backward
        x, Wih, Whh = _f32c(x), _f32c(w_ih), _f32c(w_hh)
        B, T, E = x.shape
        H = Whh.shape[1]
        dev = x.device
        xp = torch.empty(B, T, 4 * H, dtype=torch.float32, device=dev)
        gemm(L.GEMM_NT, [dict(M=B * T, A=[x], B=[Wih], C=xp, bias=_f32c(b_ih))], 4 * H, E, E, E, 4 * H)
        st = torch.empty(3, B, T, H, dtype=torch.float32, device=dev)      # Hprev, Cs, out
        Gall = torch.empty(B, T, 4 * H, dtype=torch.float32, device=dev)
        L.check(lib.mmnas_lstm_seq_fwd(L.fptr(xp), L.fptr(_f32c(b_hh)), L.fptr(Whh), L.fptr(st[0]), L.fptr(st[1]), L.fptr(Gall),
                                       L.fptr(st[2]), T, B, H, L.stream()))
        ctx.save_for_backward(x, Wih, Whh, st, Gall)
        ctx.dims = (B, T, E, H)
        return st[2]

    @staticmethod
    def backward(ctx, dout):
        lib = L.lib()
        x, Wih, Whh, st, Gall = ctx.saved_tensors
        B, T, E, H = ctx.dims
        dev = dout.device
        dout = _f32c(dout)
        DG = torch.empty(B, T, 4 * H, dtype=torch.float32, device=dev)
        L.check(lib.mmnas_lstm_seq_bwd(L.fptr(dout), L.fptr(Whh), L.fptr(st[1]), L.fptr(Gall), L.fptr(DG), T, B, H, L.stream()))
        (dWih, dWhh, dbih, dbhh), rets, sinks = _grad_bufs(ctx.params, dev)
        M = B * T
        wg_ih = gemm_desc(L.GEMM_TN, [dict(M=4 * H, A=[DG], B=[x], C=dWih)], E, M, 4 * H, E, E, accumulate=True)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm_pair(gemm_desc(L.GEMM_NN, [dict(M=M, A=[DG], B=[Wih], C=dx)], E, 4 * H, 4 * H, E, E), wg_ih)
        else:
            L.check(lib.mmnas_gemm(C.byref(wg_ih), L.stream()))
        gemm(L.GEMM_TN, [dict(M=4 * H, A=[DG], B=[st[0]], C=dWhh)], H, M, 4 * H, H, H, accumulate=True)
        L.check(lib.mmnas_colsum(L.fptr(DG), L.fptr(dbih), M, 4 * H, 4 * H, L.stream()))
        L.check(lib.mmnas_colsum(L.fptr(DG), L.fptr(dbhh), M, 4 * H, 4 * H, L.stream()))
        for sk in sinks:
            sk.done()
        return dx, rets[0], rets[1], rets[2], rets[3]


def lstm(x, mod):
    """mod: nn.LSTM(num_layers=1, batch_first=True, unidirectional); returns the output sequence [B, T, H]."""
    return LstmFn.apply(x, mod.weight_ih_l0, mod.weight_hh_l0, mod.bias_ih_l0, mod.bias_hh_l0)


def lstm_enabled():
    """The persistent-kernel LSTM is the default; MMNAS_LSTM=0 falls back to nn.LSTM (MIOpen: ~110 launches per step)."""
    import os
    return os.environ.get('MMNAS_LSTM', '1') != '0'


def lstm_supported(x, mod):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and mod.num_layers == 1 and not mod.bidirectional
            and mod.batch_first and mod.bias and mod.proj_size == 0 and mod.input_size % 4 == 0
            and bool(L.lib().mmnas_lstm_seq_supported(mod.hidden_size, x.shape[0])))


def linear(x, W, b=None, relu=False):
    return LinearFn.apply(x, W, b, relu)


class LayerNormFn(torch.autograd.Function):
    """LayerNorm of modules.py:44-56 (unbiased std, eps on the std)."""

    @staticmethod
    def forward(ctx, x, a, b, eps):
        x, a, b = _f32c(x), _f32c(a), _f32c(b)
        d = x.shape[-1]
        M = x.numel() // d
        y = torch.empty_like(x)
        L.check(L.lib().mmnas_layernorm_fwd(L.fptr(x), L.fptr(a), L.fptr(b), L.fptr(y), M, d, eps, L.stream()))
        ctx.save_for_backward(x, a)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a = ctx.saved_tensors
        dy = _f32c(dy)
        d = x.shape[-1]
        M = x.numel() // d
        dx = torch.empty_like(x)
        dab = torch.zeros(2, d, dtype=torch.float32, device=x.device)
        ws = torch.empty(L.lib().mmnas_layernorm_bwd_ws_floats(M, d), dtype=torch.float32, device=x.device)
        L.check(L.lib().mmnas_layernorm_bwd(L.fptr(x), L.fptr(a), L.fptr(dy), L.fptr(dx), L.fptr(dab[0]),
                                            L.fptr(dab[1]), None, None, L.fptr(ws), 0.0, 0, 0, M, d, ctx.eps,
                                            L.stream()))
        return dx, dab[0], dab[1], None


def layer_norm(x, a, b, eps=1e-6):
    return LayerNormFn.apply(x, a, b, eps)


class EltwiseFn(torch.autograd.Function):
    """kind: 0 zero (modules.py:96-101), 1 relu, 2 leaky-relu, 3 gelu-tanh (modules.py:104-109)."""

    @staticmethod
    def forward(ctx, x, kind):
        x = _f32c(x)
        y = torch.empty_like(x)
        L.check(L.lib().mmnas_eltwise_fwd(kind, L.fptr(x), L.fptr(y), x.numel(), L.stream()))
        ctx.save_for_backward(x)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        L.check(L.lib().mmnas_eltwise_bwd(ctx.kind, L.fptr(x), L.fptr(dy), L.fptr(dx), x.numel(), L.stream()))
        return dx, None


def eltwise(x, kind):
    return EltwiseFn.apply(x, kind)


class GluFn(torch.autograd.Function):
    """nn.GLU (+ optional relu, dropout) of GatedLinear / GLU (modules.py:112-155)."""

    @staticmethod
    def forward(ctx, h, relu, drop_p, seed, site):
        h = _f32c(h)
        C2 = h.shape[-1]
        Cc = C2 // 2
        M = h.numel() // C2
        y = torch.empty(h.shape[:-1] + (Cc,), dtype=torch.float32, device=h.device)
        L.check(L.lib().mmnas_glu_fwd(L.fptr(h), L.fptr(y), M, Cc, int(relu), drop_p, seed, site, L.stream()))
        ctx.save_for_backward(h)
        ctx.args = (M, Cc, int(relu), drop_p, seed, site)
        return y

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        dy = _f32c(dy)
        M, Cc, relu, drop_p, seed, site = ctx.args
        dh = torch.empty_like(h)
        L.check(L.lib().mmnas_glu_bwd(L.fptr(h), L.fptr(dy), L.fptr(dh), M, Cc, relu, drop_p, seed, site, L.stream()))
        return dh, None, None, None, None


def glu(h, relu=False, drop_p=0.0, seed=0, site=0):
    return GluFn.apply(h, relu, float(drop_p), int(seed), int(site))


class DropAddFn(torch.autograd.Function):
    """z = res + dropout(x): the common operator epilogue (modules.py:261-266) for composed operators."""

    @staticmethod
    def forward(ctx, x, res, drop_p, seed, site):
        x = _f32c(x)
        res = _f32c(res) if res is not None else None
        y = torch.empty_like(x)
        L.check(L.lib().mmnas_drop_add(L.fptr(x), L.fptr(res), L.fptr(y), x.numel(), drop_p, seed, site, L.stream()))
        ctx.args = (drop_p, seed, site, res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        drop_p, seed, site, has_res = ctx.args
        dy = _f32c(dy)
        if drop_p > 0:
            dx = torch.empty_like(dy)
            L.check(L.lib().mmnas_drop_add(L.fptr(dy), None, L.fptr(dx), dy.numel(), drop_p, seed, site, L.stream()))
        else:
            dx = dy
        return dx, (dy if has_res else None), None, None, None


def drop_add(x, res, drop_p=0.0, seed=0, site=1):
    return DropAddFn.apply(x, res, float(drop_p), int(seed), int(site))


class Im2ColFn(torch.autograd.Function):
    """x[B,S,d] -> col[B,S,k*d] sliding windows over the sequence (StdConv, modules.py:480-481)."""

    @staticmethod
    def forward(ctx, x, k):
        x = _f32c(x)
        B, S, d = x.shape
        col = torch.empty(B, S, k * d, dtype=torch.float32, device=x.device)
        L.check(L.lib().mmnas_im2col_seq(L.fptr(x), L.fptr(col), B, S, d, k, L.stream()))
        ctx.dims = (B, S, d, k)
        return col

    @staticmethod
    def backward(ctx, dcol):
        B, S, d, k = ctx.dims
        dcol = _f32c(dcol)
        dx = torch.empty(B, S, d, dtype=torch.float32, device=dcol.device)
        L.check(L.lib().mmnas_col2im_seq(L.fptr(dcol), L.fptr(dx), B, S, d, k, L.stream()))
        return dx, None


def _conv_seq_im2col(x, weight, bias):
    """The explicit-window form: col[B,S,k*d] (k x the input's bytes) -> one product.  Kept for shapes outside the
    strided-segment kernels (d % 32 != 0, MMNAS_GEMM_SPLIT=3) and as the A/B baseline of tools/conv_bench.py."""
    co, ci, k = weight.shape
    col = Im2ColFn.apply(x, k)
    wp = weight.permute(0, 2, 1).reshape(co, k * ci)  # [co, t*ci + c] to match the window layout
    return linear(col, wp, bias)


_conv_w_cache = {}
_param_epoch = 0


def note_raw_parameter_write():
    """Called by everything that writes parameters through RAW POINTERS (FlatAdam / ArchAdam: mmnas_adam_step on the flat
    buffer the parameters are views of).  Such writes do not move a tensor's version counter, so every cache of values
    DERIVED from parameters (the re-arranged StdConv weights below) also keys on this epoch.  (without it
    the cached arrangement of a conv weight survived the optimizer step and the operator trained on its initial weights.)"""
    global _param_epoch
    _param_epoch += 1


def _conv_weights(weight):
    """conv.weight [d_out, d_in, k] in the two arrangements the products read, re-made only when the parameter has been
    written: its version counter moves with every in-place torch update, the library's parameter epoch with every
    raw-pointer optimizer step (note_raw_parameter_write); a `.data` re-homing changes the storage pointer:
      fwd [d_out, k * d_in]  column t * d_in + c = W[:, c, t]              (B of the forward NT product)
      rev [k * d_out, d_in]  row j * d_out + o = W[o, :, k - 1 - j]       (B of the data-gradient NN product)"""
    key = id(weight)
    stamp = (weight._version, _param_epoch, weight.data_ptr())
    hit = _conv_w_cache.get(key)
    if hit is not None and hit[0] is weight and hit[1] == stamp and hit[2].device == weight.device:
        return hit[2], hit[3]
    co, ci, k = weight.shape
    w = weight.detach()
    fwd = w.permute(0, 2, 1).reshape(co, k * ci).contiguous()
    rev = w.flip(2).permute(2, 0, 1).reshape(k * co, ci).contiguous()
    if len(_conv_w_cache) > 256:
        _conv_w_cache.clear()
    _conv_w_cache[key] = (weight, stamp, fwd, rev)
    return fwd, rev


def _pad_seq(x, front, Sp, slack):
    """[B, S, d] -> the zero-padded row grid [B * Sp + slack, d] (mmnas_pad_seq), rows rounded up to a multiple of 32."""
    B, S, d = x.shape
    rows = (B * Sp + slack + 31) // 32 * 32
    xp = torch.empty(rows, d, dtype=torch.float32, device=x.device)
    L.check(L.lib().mmnas_pad_seq(L.fptr(x), L.fptr(xp), B, S, d, front, Sp, rows, L.stream()))
    return xp


class ConvSeqFn(torch.autograd.Function):
    """Dense k-tap Conv1d over the sequence axis (StdConv, modules.py:472,480-481) WITHOUT a window buffer.  On the
    zero-padded row grid xp[b * Sp + j] = x[b, j - pad] (Sp = S + 2 pad) the im2col matrix col[m, t d + c] = xp[m + t, c]
    is xp itself read with row stride d: overlapping rows, which mmnas_gemm takes as they are (lda = d, K = k d).
      forward        y''[m]  = col(xp)[m] Wfwd^T            NT, M = B Sp rows; y[b, s] = y''[b Sp + s]
      data gradient  dx''[m] = col(dyp)[m] Wrev             NN on the padded output gradient, taps reversed
      weight grad    dW''    = dye^T col(xp)                ONE TN product; dye = dy with 2 pad zero rows behind every sequence
    Rows of y'' / dx'' in a sequence's padding are never read; a saved tensor is the padded input (1.06x the input at
    S = 100, k = 7) instead of the k-fold window buffer."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _f32c(x)
        B, S, d = x.shape
        co, ci, k = weight.shape
        pad = k // 2
        Sp = S + 2 * pad
        wf, wr = _conv_weights(weight)
        xp = _pad_seq(x, pad, Sp, k)
        M = B * Sp
        yp = torch.empty(M, co, dtype=torch.float32, device=x.device)
        gemm(L.GEMM_NT, [dict(M=M, A=[xp], B=[wf], C=yp, bias=(_f32c(bias) if bias is not None else None))], co, k * ci, ci, k * ci, co)
        ctx.save_for_backward(xp, wr)
        ctx.dims = (B, S, d, co, ci, k, bias is not None)
        return yp.view(B, Sp, co)[:, :S].contiguous()

    @staticmethod
    def backward(ctx, dy):
        xp, wr = ctx.saved_tensors
        B, S, d, co, ci, k, has_bias = ctx.dims
        pad = k // 2
        Sp = S + 2 * pad
        M = B * Sp
        dy = _f32c(dy)
        dev = dy.device
        dx = dw = None
        dgrad = wgrad = None
        if ctx.needs_input_grad[0]:
            # dx[b,s] = sum_t dy[b, s - t + pad] W_t = sum_j dyp[b Sp + s + j] W_{k-1-j}
            dyp = _pad_seq(dy, pad, Sp, k)
            dxp = torch.empty(M, ci, dtype=torch.float32, device=dev)
            dgrad = gemm_desc(L.GEMM_NN, [dict(M=M, A=[dyp], B=[wr], C=dxp)], ci, k * co, co, ci, ci)
        if ctx.needs_input_grad[1]:
            # dW[o, c, t] = sum_{b,s} dy[b,s,o] xp[b Sp + s + t, c]: rows of dy at b Sp + s (padding behind the sequence)
            dye = _pad_seq(dy, 0, Sp, 0)
            Kr = dye.shape[0]                                   # (a multiple of 32; xp has at least as many rows)
            dwf = torch.zeros(co, k * ci, dtype=torch.float32, device=dev)
            wgrad = gemm_desc(L.GEMM_TN, [dict(M=co, A=[dye], B=[xp], C=dwf)], k * ci, Kr, co, ci, k * ci, accumulate=True)
        if dgrad is not None and wgrad is not None:
            gemm_pair(dgrad, wgrad)                             # one launch: the second product starts as the first drains
        elif dgrad is not None or wgrad is not None:
            L.check(L.lib().mmnas_gemm(C.byref(dgrad if dgrad is not None else wgrad), L.stream()))
        if dgrad is not None:
            dx = dxp.view(B, Sp, ci)[:, :S].contiguous()
        if wgrad is not None:
            dw = dwf.view(co, k, ci).permute(0, 2, 1)           # the parameter's own [d_out, d_in, k] layout
        db = None
        if has_bias and ctx.needs_input_grad[2]:
            db = torch.zeros(co, dtype=torch.float32, device=dev)
            L.check(L.lib().mmnas_colsum(L.fptr(dy), L.fptr(db), B * S, co, co, L.stream()))
        return dx, dw, db


def conv_seq(x, weight, bias):
    """Dense Conv1d over the sequence axis: weight [d_out, d_in, k] (conv.weight, modules.py:472)."""
    co, ci, k = weight.shape
    # (overlapping rows read past an operand's M * lda extent: only the buffer-load path answers that with zeros)
    # (and the padded grid computes S + 2 pad rows per sequence: for the 14-token stream with wide kernels the window
    #  buffer is the cheaper form -- profiles/r04_conv_microbench.txt)
    direct = (x.is_cuda and ci % 32 == 0 and co % 32 == 0 and os.environ.get('MMNAS_CONV_IM2COL', '0') != '1'
              and (x.shape[1] + 2 * (k // 2) <= 1.25 * x.shape[1] or os.environ.get('MMNAS_CONV_IM2COL') == '0')
              and os.environ.get('MMNAS_GEMM_GENERIC') is None and 4.0 * x.shape[0] * (x.shape[1] + k) * max(ci, co) * k < 3.9e9)
    if not direct:
        return _conv_seq_im2col(x, weight, bias)
    return ConvSeqFn.apply(x, weight, bias)


class DwConvFn(torch.autograd.Function):
    """Depthwise k-tap stencil over the sequence axis (depthwise_conv, modules.py:438-439)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _f32c(x), _f32c(w)
        B, S, d = x.shape
        k = w.shape[-1]
        y = torch.empty_like(x)
        L.check(L.lib().mmnas_dwconv_seq_fwd(L.fptr(x), L.fptr(w), L.fptr(_f32c(b) if b is not None else None),
                                             L.fptr(y), B, S, d, k, L.stream()))
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _f32c(dy)
        B, S, d = x.shape
        k = w.shape[-1]
        dx = torch.empty_like(x)
        dw = torch.zeros_like(w)
        db = torch.zeros(d, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        L.check(L.lib().mmnas_dwconv_seq_bwd(L.fptr(x), L.fptr(w), L.fptr(dy), L.fptr(dx), L.fptr(dw), L.fptr(db),
                                             B, S, d, k, L.stream()))
        return dx, dw, db


def depthwise_conv_seq(x, weight, bias):
    return DwConvFn.apply(x, weight, bias)


class AttFlatPoolFn(torch.autograd.Function):
    """Pooling stage of AttFlat (modules.py:78-84): masked softmax of the glimpse logits over the sequence and the
    attention-weighted sum of the features."""

    @staticmethod
    def forward(ctx, logits, x, mask):
        logits, x = _f32c(logits), _f32c(x)
        B, S, d = x.shape
        G = logits.shape[-1]
        m8 = _mask_u8(mask, B, S)
        probs = torch.empty(B, S, G, dtype=torch.float32, device=x.device)
        pooled = torch.empty(B, G * d, dtype=torch.float32, device=x.device)
        L.check(L.lib().mmnas_attflat_pool_fwd(L.fptr(logits), L.fptr(x), L.ptr(m8), L.fptr(probs), L.fptr(pooled),
                                               B, S, d, G, L.stream()))
        ctx.save_for_backward(probs, x, m8)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        probs, x, m8 = ctx.saved_tensors
        dpooled = _f32c(dpooled)
        B, S, d = x.shape
        G = probs.shape[-1]
        dlogits = torch.empty_like(probs)
        dx = torch.empty_like(x)
        L.check(L.lib().mmnas_attflat_pool_bwd(L.fptr(probs), L.fptr(x), L.ptr(m8), L.fptr(dpooled), L.fptr(dlogits), L.fptr(dx),
                                               B, S, d, G, L.stream()))
        return dlogits, dx, None


def attflat_pool(logits, x, mask):
    return AttFlatPoolFn.apply(logits, x, mask)


# ------------------------------------------------------------------------------------------
# backbone chain: all cell operators of a backbone in one C call per direction
# ------------------------------------------------------------------------------------------
def chain_enabled():
    import os
    return os.environ.get('MMNAS_CHAIN', '1') != '0'


def mixed_chain_enabled():
    """Architecture step (MixedOp modes 'full' / 'two') through the backbone chain: every evaluated candidate in one native
    call per direction, the candidates' LayerNorms and the gated sum of a node as one kernel.  MMNAS_MIXED_CHAIN=0 keeps
    the per-candidate path (one autograd node per candidate + ops.MixedSumFn)."""
    import os
    return os.environ.get('MMNAS_MIXED_CHAIN', '1') != '0'


def side_stream_enabled():
    """Weight-gradient work of the backbone chain on a second stream (MMNAS_SIDE_STREAM=1).  OFF by default: measured on
    the supernet and training steps it does not pay -- unpaired data- / weight-gradient launches cost more than the
    paired ones save, and workgroups of the deferred products delay the short, latency-bound encoder / LSTM kernels they
    were meant to fill the gaps of (6.99 -> 7.05 ms released per operator, 7.35 ms released per phase)."""
    import os
    v = os.environ.get('MMNAS_SIDE_STREAM', '0')
    return 2 if v == 'rel' else int(v == '1')   # 'rel': only the relation-bias backward (parameter gradients only) moves


_side_pending = []          # arenas / inputs the side stream may still be reading (released by join_side_stream)
_side_join_queued = [False]


def join_side_stream():
    """Make the current stream wait for the parameter-gradient work the last backbone backward put on the library's
    side stream, then release the buffers that work reads.  Runs automatically at the end of every backward pass that
    used the side stream; call it by hand only when reading parameter gradients from inside a backward hook."""
    _side_join_queued[0] = False
    st = L.stream()
    L.check(L.lib().mmnas_chain_join(st, st))
    del _side_pending[:]


def side_stream_barrier(waiting_stream):
    """Make another stream (a communication stream about to all-reduce gradients) wait for the side-stream work issued so
    far for the current stream's backbone backward."""
    L.check(L.lib().mmnas_chain_join(L.stream(), waiting_stream.cuda_stream))


def _sinked(params):
    """True when every parameter's gradient is a view of a flat gradient buffer the kernels may add into."""
    if not _sinks_on[0]:
        return False
    for p in params:
        s = getattr(p, '_mmnas_sink', None)
        if s is None or p.grad is not s.view:
            return False
    return True


def _cached_record(owner, key, first, build):
    """Per-module cache of a prefilled descriptor: rebuilt when the key (stream, mode, relation stem) or the storage of
    the module's first parameter / of its gradient view changes (.to(device), a new flat buffer); the seed is per call."""
    c = owner.__dict__.get('_mmnas_rec')
    g = first.grad
    if c is not None and c[0] == key and c[1] == first.data_ptr() and g is not None and c[2] == g.data_ptr():
        return c[3], c[4]
    rec, params = build()
    if rec is not None:
        owner.__dict__['_mmnas_rec'] = (key, first.data_ptr(), first.grad.data_ptr(), rec, params)
    return rec, params


def chain_att_record_cached(op, on_y, self_att, rel_handle):
    mh = op.mhatt
    key = (on_y, op.training, id(rel_handle.weight) if rel_handle is not None else 0)
    rec, params = _cached_record(op, key, mh.linear_q.weight, lambda: chain_att_record(
        on_y, mh, op.ln if op.norm else None, op.norm, op.residual, self_att, rel_handle, op.training))
    if rec.att.drop_p > 0:
        rec.att.seed = next_seed()
    return rec, params


def chain_mlp_record_cached(op, on_y, weights, biases):
    key = (on_y, op.training, 0)
    rec, params = _cached_record(op, key, weights[0], lambda: chain_mlp_record(
        on_y, weights, biases, op.ln if op.norm else None, op.norm, op.residual, op.drop_p, op.training))
    if rec.mlp.drop_p > 0:
        rec.mlp.seed = next_seed()
    return rec, params


def _grad_ptr(p):
    return p.grad.data_ptr()


# ---- the backbone chain in PLAIN AUTOGRAD use (round 5) ----------------------------------------------------------------
# Without a flat gradient buffer (no reducer / FlatAdam: the unchanged scripts under stock DDP, any other autograd use) the
# chain used to step aside for one autograd node per operator.  Now the parameters of the chain's operators are autograd
# INPUTS of ops.BackboneFn: the kernels accumulate their gradients into the views of one zero-filled buffer made per call,
# and backward returns those views -- autograd, AccumulateGrad hooks and stock DDP get every gradient the ordinary way,
# from ONE node per direction instead of ~60 (and with the chain's grouped launches).  Weight steps / fixed architectures
# only (the arch step's mixed chain needs the gate blocks).
# OFF by default (MMNAS_AUTOGRAD_CHAIN=1 enables it): MEASURED NEUTRAL where it was meant to help -- the unchanged search loop
# is host-bound on what surrounds the operators (bench `search_vqa_dropin`, one box: 14.0 ms per step with the per-operator
# nodes, 14.5 ms with this; cProfile: 190 instead of 480 Function.apply calls, but run_backward stays at 5.9 ms -- 900
# AccumulateGrad nodes, 230 zero + gradient adds -- and stock clip_grad_norm_ + Adam over 900 tensors are 6.3 ms).  Kept,
# tested against the per-operator nodes and the reference, for callers whose loop is not host-bound.
def autograd_chain_enabled():
    return os.environ.get('MMNAS_AUTOGRAD_CHAIN', '0') == '1'


def _null_ptr(_p):
    return None


def _template(owner, key, first, build):
    """Per-module cache of a descriptor WITHOUT gradient pointers (autograd-mode chain): rebuilt when the key or the storage
    of the module's first parameter changes."""
    c = owner.__dict__.get('_mmnas_tmpl')
    if c is not None and c[0] == key and c[1] == first.data_ptr():
        return c[2], c[3]
    rec, params = build()
    owner.__dict__['_mmnas_tmpl'] = (key, first.data_ptr(), rec, params)
    return rec, params


def chain_att_template(op, on_y, self_att, rel_handle):
    mh = op.mhatt
    key = (on_y, op.training, id(rel_handle.weight) if rel_handle is not None else 0)
    return _template(op, key, mh.linear_q.weight, lambda: chain_att_record(
        on_y, mh, op.ln if op.norm else None, op.norm, op.residual, self_att, rel_handle, op.training, gptr=_null_ptr))


def chain_mlp_template(op, on_y, weights, biases):
    return _template(op, (on_y, op.training, 0), weights[0], lambda: chain_mlp_record(
        on_y, weights, biases, op.ln if op.norm else None, op.norm, op.residual, op.drop_p, op.training, gptr=_null_ptr))


def patched_record(tmpl, params, gp):
    """A private copy of a template descriptor with the gradient pointers gp(p) of its parameters (the order the builders
    list them in) and a fresh dropout seed."""
    rec = L.ChainOp.from_buffer_copy(tmpl)
    if rec.kind == 0:
        a = rec.att
        a.dWq, a.dWk, a.dWv, a.dWm = gp(params[0]), gp(params[1]), gp(params[2]), gp(params[3])
        k = 4
        if a.flags & L.F_NORM:
            a.dln_a, a.dln_b = gp(params[k]), gp(params[k + 1])
            k += 2
        if a.flags & L.F_REL:
            a.dWr, a.dbr, a.dWy, a.dby = gp(params[k]), gp(params[k + 1]), gp(params[k + 2]), gp(params[k + 3])
        if a.drop_p > 0:
            a.seed = next_seed()
    else:
        m = rec.mlp
        k = 0
        for i in range(m.nl):
            m.dW[i] = gp(params[k])
            k += 1
            if m.b[i]:
                m.db[i] = gp(params[k])
                k += 1
        if m.flags & L.F_NORM:
            m.dln_a, m.dln_b = gp(params[k]), gp(params[k + 1])
        if m.drop_p > 0:
            m.seed = next_seed()
    return rec


def chain_att_record(on_y, mh, ln, norm, residual, self_att, rel_handle, training, gptr=_grad_ptr):
    """(ChainOp, params) for an attention-family operator: mh = its MHAtt / RelMHAtt, rel_handle = a fusable RelHandle
    or None.  gptr(p): where p's gradient is accumulated (default: its flat-buffer view p.grad; the autograd-mode chain
    builds a template with null pointers and patches a per-call buffer's views in: patch_grad_ptrs)."""
    rec = L.ChainOp()
    rec.kind, rec.on_y = 0, int(on_y)
    a = rec.att
    Wq, Wk, Wv, Wm = mh.linear_q.weight, mh.linear_k.weight, mh.linear_v.weight, mh.linear_merge.weight
    params = [Wq, Wk, Wv, Wm]
    a.di, a.dh = Wq.shape[0], mh.HBASE
    a.H = a.di // a.dh
    drop = mh.drop_p if training else 0.0
    flags = (L.F_NORM if norm else 0) | (L.F_RESIDUAL if residual else 0) | (L.F_SELF if self_att else 0)
    if drop > 0:
        flags |= L.F_TRAIN
    a.drop_p, a.eps = float(drop), float(ln.eps if norm else 1e-6)
    a.seed = 0   # (drawn per call by the cached wrapper)
    a.Wq, a.Wk, a.Wv, a.Wm = Wq.data_ptr(), Wk.data_ptr(), Wv.data_ptr(), Wm.data_ptr()
    a.dWq, a.dWk, a.dWv, a.dWm = gptr(Wq), gptr(Wk), gptr(Wv), gptr(Wm)
    if norm:
        a.ln_a, a.ln_b = ln.a_2.data_ptr(), ln.b_2.data_ptr()
        a.dln_a, a.dln_b = gptr(ln.a_2), gptr(ln.b_2)
        params += [ln.a_2, ln.b_2]
    if rel_handle is not None:
        lr = mh.linear_r
        flags |= L.F_REL | L.F_RELRAW
        a.R, a.C = lr.weight.shape[1], rel_handle.weight.shape[1]
        a.Wr, a.br, a.dWr, a.dbr = lr.weight.data_ptr(), lr.bias.data_ptr(), gptr(lr.weight), gptr(lr.bias)
        a.Wy, a.by = rel_handle.weight.data_ptr(), rel_handle.bias.data_ptr()
        a.dWy, a.dby = gptr(rel_handle.weight), gptr(rel_handle.bias)
        params += [lr.weight, lr.bias, rel_handle.weight, rel_handle.bias]
    a.flags = flags
    return rec, params


def chain_mlp_record(on_y, weights, biases, ln, norm, residual, drop_p, training, gptr=_grad_ptr):
    rec = L.ChainOp()
    rec.kind, rec.on_y = 1, int(on_y)
    m = rec.mlp
    nl = len(weights)
    m.nl = nl
    m.dims[0] = weights[0].shape[1]
    params = []
    for i, (w, b) in enumerate(zip(weights, biases)):
        m.dims[i + 1] = w.shape[0]
        m.W[i], m.dW[i] = w.data_ptr(), gptr(w)
        params.append(w)
        if b is not None:
            m.b[i], m.db[i] = b.data_ptr(), gptr(b)
            params.append(b)
    drop = drop_p if training else 0.0
    flags = (L.F_NORM if norm else 0) | (L.F_RESIDUAL if residual else 0)
    if drop > 0:
        flags |= L.F_TRAIN
    m.flags, m.drop_p, m.eps = flags, float(drop), float(ln.eps if norm else 1e-6)
    m.seed = 0   # (drawn per call by the cached wrapper)
    if norm:
        m.ln_a, m.ln_b, m.dln_a, m.dln_b = ln.a_2.data_ptr(), ln.b_2.data_ptr(), gptr(ln.a_2), gptr(ln.b_2)
        params += [ln.a_2, ln.b_2]
    return rec, params


def _acquire_sinks(ctx, params):
    """Section forward (BackboneFn / HeadFn): when a backward will follow, count this node as a live user of every
    parameter's gradient sink.  Several forwards may precede one backward (train_itm.py:380-391 runs three); a
    parameter's gradient is complete -- `ready()` for a data-parallel reducer -- only when the LAST of them has run
    its backward, not the first."""
    ctx.uniq = list({id(p): p for p in params}.values())   # (the shared relation stem appears once per relation operator)
    if not any(ctx.needs_input_grad[:2]):
        return False
    for p in ctx.uniq:
        p._mmnas_sink.acquire()
    return True


def _last_live(ctx, params):
    """True when this node is the only live user left of every one of its parameters."""
    return not ctx.counted or all(p._mmnas_sink.live() <= 1 for p in ctx.uniq)


def _release_sinks(ctx, params):
    for p in ctx.uniq:
        s = p._mmnas_sink
        if not ctx.counted or s.release():
            s.ready()


# ------------------------------------------------------------------------------------------
# Ragged batches ("unpadding").  The loaders pad every image to 100 region rows behind its detected boxes
# (load_data_vqa.py:221-246); the reference computes on the padding rows and then masks them as keys everywhere and in
# AttFlat (hygr_vqa.py:113-122, modules.py:78-84,195-196), so no logit and no parameter gradient depends on them.  With
# MMNAS_UNPAD=1 (or set_unpad(True)) the backbone chain runs the decoder stream on the valid rows only: packed [sum n_b, d]
# matrices for every projection / FFN / LayerNorm, attention over each sample's own rows, the relation bias over its
# n_b x n_b corner.  Logits and parameter gradients are those of the padded computation (tests/test_chain_gpu.py); the
# decoder OUTPUT rows of the padding are zeros instead of the reference's (unread) values, and dropout draws from other
# element indices.  Off by default.
# ------------------------------------------------------------------------------------------
_unpad = [None]


def unpad_enabled():
    if _unpad[0] is None:
        _unpad[0] = os.environ.get('MMNAS_UNPAD', '0') == '1'
    return _unpad[0]


def set_unpad(on):
    """Switch the ragged decoder stream on / off (returns the previous setting)."""
    prev = unpad_enabled()
    _unpad[0] = bool(on)
    return prev


class Ragged:
    """Device / host description of a ragged batch: off [B+1] int32 prefix sums of the lengths, tile_off [B+1] prefix sums
    of ceil(n_b^2 / 32) (the relation-bias backward's tiles), the totals as host ints."""

    def __init__(self, lengths, device):
        import numpy as np
        n = np.asarray(lengths, dtype=np.int64)
        self.lengths = tuple(int(v) for v in n)
        off = np.concatenate([[0], np.cumsum(n)]).astype(np.int32)
        toff = np.concatenate([[0], np.cumsum((n * n + 31) // 32)]).astype(np.int32)
        both = torch.from_numpy(np.stack([off, toff])).to(device, non_blocking=True)
        self.off, self.tile_off = both[0], both[1]
        self.N, self.ntiles = int(off[-1]), int(toff[-1])


_ragged_cache = {}


def ragged_info_for(feat, mask):
    """Ragged description of the batch whose region features are `feat` [B, S, F] and whose padding mask is `mask` (True =
    padding), or None when the switch is off / the batch has no padding / the valid rows are not a prefix of every sample /
    a sample is empty / S > 128.  The lengths come from `feat._mmnas_lengths` when the data pipeline attached them (no
    synchronisation), else from the mask with ONE device-to-host copy per distinct features tensor (cached on the tensor
    object and its version counter: a resident batch pays it once)."""
    if not unpad_enabled() or not feat.is_cuda or feat.dim() != 3:
        return None
    if side_stream_enabled():      # the chain's side-stream backward runs on padded rows only (mmnas_chain_bwd refuses the
        return None                # combination -- in the middle of autograd): decide here, before anything is packed
    B, S = feat.shape[0], feat.shape[1]
    if S > 128:
        return None
    lens = getattr(feat, '_mmnas_lengths', None)
    if lens is None:
        key = id(feat)
        hit = _ragged_cache.get(key)
        if hit is not None and hit[0]() is feat and hit[1] == feat._version:
            return hit[2]
        m = mask.reshape(B, S)
        n = (~m).sum(1)
        ok = (m == (torch.arange(S, device=m.device)[None, :] >= n[:, None])).all()
        host = torch.cat([n, ok.long().view(1)]).cpu()
        lens = host[:B].tolist() if int(host[B]) else None
        info = _make_ragged(lens, B, S, feat.device)
        import weakref
        if len(_ragged_cache) > 64:
            _ragged_cache.clear()
        _ragged_cache[key] = (weakref.ref(feat), feat._version, info)
        return info
    return _make_ragged([int(v) for v in lens], B, S, feat.device)


def _make_ragged(lens, B, S, device):
    if lens is None or len(lens) != B or min(lens) < 1 or max(lens) > S or sum(lens) >= B * S:
        return None
    return Ragged(lens, device)


def pack_rows(x, rg):
    """[B, S, d] -> packed [rg.N, d]."""
    B, S, d = x.shape
    out = torch.empty(rg.N, d, dtype=torch.float32, device=x.device)
    L.check(L.lib().mmnas_pack_rows(L.fptr(x), L.ptr(rg.off), L.fptr(out), B, S, d, L.stream()))
    return out


def unpack_rows(xp, rg, B, S):
    """packed [rg.N, d] -> [B, S, d], zeros in the padding rows."""
    d = xp.shape[-1]
    out = torch.empty(B, S, d, dtype=torch.float32, device=xp.device)
    L.check(L.lib().mmnas_unpack_rows(L.fptr(xp), L.ptr(rg.off), L.fptr(out), B, S, d, L.stream()))
    return out


class _UnpackRowsFn(torch.autograd.Function):
    """unpack_rows with a gradient (the rare route: packed decoder output, but no native head to take it)."""

    @staticmethod
    def forward(ctx, xp, rg, B, S):
        ctx.rg = rg
        return unpack_rows(_f32c(xp), rg, B, S)

    @staticmethod
    def backward(ctx, g):
        return pack_rows(_f32c(g), ctx.rg), None, None, None


def unpack_rows_fn(xp, rg, B, S):
    return _UnpackRowsFn.apply(xp, rg, B, S)


class _PackRowsFn(torch.autograd.Function):
    """pack_rows with a gradient: the stem packs the raw region features, which carry a graph when the box features ride in
    them (C.BBOX_FEATURE: cat(frcn_feat, bboxfeat_linear(bbox_feat)), full_vqa.py:93-97) -- the raw call would cut it and
    `bboxfeat_linear` would silently never train."""

    @staticmethod
    def forward(ctx, x, rg):
        ctx.rg, ctx.shape = rg, x.shape
        return pack_rows(_f32c(x), rg)

    @staticmethod
    def backward(ctx, g):
        B, S, _ = ctx.shape
        return unpack_rows(_f32c(g), ctx.rg, B, S), None


def pack_rows_fn(x, rg):
    """[B, S, d] -> packed [rg.N, d]; differentiable when x needs a gradient, the raw kernel call otherwise."""
    x = _f32c(x)
    if x.requires_grad and torch.is_grad_enabled():
        return _PackRowsFn.apply(x, rg)
    return pack_rows(x, rg)


class BackboneFn(torch.autograd.Function):
    """Backbone_*.forward (hygr_vqa.py:45-52) through mmnas_chain_fwd/bwd.  Parameter gradients go straight into the
    flat gradient buffer (every parameter of the chain has an attached sink: checked by the caller), so the parameters
    are not autograd inputs of this node."""

    @staticmethod
    def forward(ctx, x, y, x_mask, y_mask, x_rel, y_rel, records, params, op_params=None, mixed=None, ragged=None, packed_io=False,
                gviews=None, *ptensors):
        """gviews + ptensors (plain autograd use, see autograd_chain_enabled): the chain's parameters as autograd inputs and,
        aligned with them, the views of a per-call zero-filled buffer the kernels accumulate their gradients into (the
        descriptors in `records` point there); backward returns the views.
        packed_io (with ragged): `y` arrives PACKED [ragged.N, d] (the stem projected the valid region rows only) and the
        decoder output is returned packed too (the head's AttFlat takes packed rows): no pack / unpack launches at all."""
        lib = L.lib()
        x, y = _f32c(x), _f32c(y)
        B, Sx, d = x.shape
        packed_io = bool(packed_io and ragged is not None)
        Sy = int(y_mask.shape[-1]) if packed_io else y.shape[1]
        if ragged is not None and not packed_io:     # the decoder stream on its valid rows only (see Ragged)
            y = pack_rows(y, ragged)
        n = len(records)
        arr = (L.ChainOp * n)(*records)
        ch = L.Chain()
        ch.n_ops, ch.ops = n, arr
        ch.B, ch.Sx, ch.Sy, ch.d = B, Sx, Sy, d
        xm, ym = _mask_u8(x_mask, B, Sx), _mask_u8(y_mask, B, Sy)
        xr = _f32c(x_rel) if x_rel is not None else None
        yr = _f32c(y_rel) if y_rel is not None else None
        ch.x_in, ch.y_in, ch.x_mask, ch.y_mask = L.fptr(x), L.fptr(y), L.ptr(xm), L.ptr(ym)
        ch.x_rel, ch.y_rel = L.fptr(xr), L.fptr(yr)
        if mixed is not None:      # architecture step: (gate block, gate-gradient block, row width) of the supernet's nodes
            ch.mixed, ch.gate, ch.dgate, ch.gate_width = 1, mixed[0], mixed[1], mixed[2]
        if ragged is not None:
            ch.y_off, ch.y_tile_off, ch.Ny, ch.y_ntiles = L.ptr(ragged.off), L.ptr(ragged.tile_off), ragged.N, ragged.ntiles
        sz = C.c_size_t()
        L.check(lib.mmnas_chain_plan(C.byref(ch), C.byref(sz)))   # (host arithmetic only: a few microseconds)
        arena = _bytes(sz.value, x.device)
        x_out, y_out = torch.empty_like(x), torch.empty_like(y)
        ch.arena, ch.x_out, ch.y_out = L.ptr(arena), L.fptr(x_out), L.fptr(y_out)
        L.check(lib.mmnas_chain_fwd(C.byref(ch), L.stream()))
        ctx.keep = (ch, arr, arena, x, y, xm, ym, xr, yr, x_out, y_out, params)
        ctx.op_params = op_params
        ctx.ragged = (ragged, B, Sy, packed_io)
        ctx.gviews = gviews
        ctx.n_extra = len(ptensors)
        if gviews is not None:
            ctx.counted, ctx.uniq = False, []
        else:
            ctx.counted = _acquire_sinks(ctx, params)
        if ragged is not None and not packed_io:
            return x_out, unpack_rows(y_out, ragged, B, Sy)
        return x_out, y_out

    @staticmethod
    def backward(ctx, dx_out, dy_out):
        if ctx.keep is None:
            raise RuntimeError('BackboneFn: backward ran a second time (its arena is released after the first)')
        lib = L.lib()
        ch, arr, arena, x, y, xm, ym, xr, yr, x_out, y_out, params = ctx.keep
        dx_out = _f32c(dx_out) if dx_out is not None else None
        ragged, B, Sy, packed_io = ctx.ragged
        if packed_io:
            dy_out = _f32c(dy_out) if dy_out is not None else torch.zeros_like(y_out)
        elif ragged is not None:     # (the gradient of the zero-filled padding rows is dropped: nothing was computed there)
            dy_out = pack_rows(_f32c(dy_out), ragged) if dy_out is not None else torch.zeros_like(y_out)
        else:
            dy_out = _f32c(dy_out) if dy_out is not None else torch.zeros_like(y_out)
        dx_in, dy_in = torch.empty_like(x), torch.empty_like(y)
        ch.dx_out, ch.dy_out, ch.dx_in, ch.dy_in = L.fptr(dx_out), L.fptr(dy_out), L.fptr(dx_in), L.fptr(dy_in)
        side = side_stream_enabled() if ctx.gviews is None else 0   # (autograd mode: the returned gradients are read right away)
        ch.use_side_stream = int(side)
        # a data-parallel reducer gets events recorded INSIDE the call, behind the operator that completes each of its
        # buckets, so that the bucket's all-reduce overlaps the backward of the operators issued after it
        marks = marr = None
        owner = getattr(getattr(params[0], '_mmnas_sink', None), 'owner', None) if (params and ctx.gviews is None) else None
        # (marks only from the LAST live node over these parameters: with several forwards before one backward -- the ITM
        #  triplet step -- an earlier node's gradients are a third of the bucket's, not all of it)
        if owner is not None and ctx.op_params is not None and not side and _last_live(ctx, params):
            marks = owner.chain_marks(ctx.op_params)
        if marks is not None:
            marr = (C.c_void_p * len(marks))(*[(ev.cuda_event if ev is not None else None) for ev in marks])
            ch.marks = C.cast(marr, C.c_void_p)
        else:
            ch.marks = None
        L.check(lib.mmnas_chain_bwd(C.byref(ch), L.stream()))
        ch.marks = None
        ctx.keep = None
        if side:
            # the side stream still reads the arena and the saved inputs: keep them until the join at the end of
            # this backward pass (queued once per pass)
            _side_pending.append((arena, x, y, xm, ym, xr, yr, x_out, y_out, dx_out, dy_out, arr))
            if not _side_join_queued[0]:
                _side_join_queued[0] = True
                torch.autograd.Variable._execution_engine.queue_callback(join_side_stream)
        if ctx.gviews is None:
            _release_sinks(ctx, params)   # data-parallel reducers learn which gradients are now completely enqueued
        if ragged is not None and not packed_io:
            dy_in = unpack_rows(dy_in, ragged, B, Sy)
        head = (dx_in, dy_in, None, None, None, None, None, None, None, None, None, None, None)
        if ctx.gviews is not None:
            gv, ctx.gviews = ctx.gviews, None
            return head + tuple(gv)
        return head + (None,) * ctx.n_extra


def backbone_chain(x, y, x_mask, y_mask, x_rel, y_rel, records, params, op_params=None, mixed=None, ragged=None, packed_io=False,
                   gviews=None, ptensors=()):
    return BackboneFn.apply(x, y, x_mask, y_mask, x_rel, y_rel, records, params, op_params, mixed, ragged, packed_io, gviews, *ptensors)


class HeadFn(torch.autograd.Function):
    """AttFlat(x) + AttFlat(y) -> proj_norm -> proj (hygr_vqa.py:113-119) through mmnas_head_fwd/bwd: one native call
    per direction.  As for BackboneFn the parameters' gradients go straight into the flat gradient buffer."""

    @staticmethod
    def forward(ctx, x, y, x_mask, y_mask, hd, params, ragged=None):
        """ragged (ops.Ragged): `y` holds the PACKED image rows [ragged.N, d] of a ragged decoder stream; AttFlat's masked
        softmax runs over each sample's own rows (the mask hides exactly the padding rows: modules.py:78-81)."""
        lib = L.lib()
        x, y = _f32c(x), _f32c(y)
        B, Sx, d = x.shape
        if ragged is not None:
            Sy = int(y_mask.shape[-1])
            hd.sy.off, hd.sy.M = L.ptr(ragged.off), ragged.N
        else:
            Sy = y.shape[1]
            hd.sy.off, hd.sy.M = None, 0
        xm, ym = _mask_u8(x_mask, B, Sx), _mask_u8(y_mask, B, Sy)
        hd.B, hd.d = B, d
        hd.sx.S, hd.sy.S = Sx, Sy
        hd.sx.x, hd.sy.x, hd.sx.mask, hd.sy.mask = L.fptr(x), L.fptr(y), L.ptr(xm), L.ptr(ym)
        key = ('head', B, Sx, Sy, d, hd.MID, hd.G, hd.OUT, hd.ANS)    # (a packed side plans its padded upper bound)
        nbytes = _plan_cache.get(key)
        if nbytes is None:
            sz = C.c_size_t()
            off, M = hd.sy.off, hd.sy.M
            hd.sy.off, hd.sy.M = None, 0
            L.check(lib.mmnas_head_plan(C.byref(hd), C.byref(sz)))
            hd.sy.off, hd.sy.M = off, M
            nbytes = _plan_cache[key] = sz.value
        arena = _bytes(nbytes, x.device)
        logits = torch.empty(B, hd.ANS, dtype=torch.float32, device=x.device)
        hd.arena, hd.logits = L.ptr(arena), L.fptr(logits)
        L.check(lib.mmnas_head_fwd(C.byref(hd), L.stream()))
        ctx.keep = (hd, arena, x, y, xm, ym, params, ragged)
        ctx.counted = _acquire_sinks(ctx, params)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.keep is None:
            raise RuntimeError('HeadFn: backward ran a second time (its arena is released after the first)')
        hd, arena, x, y, xm, ym, params, ragged = ctx.keep
        dlogits = _f32c(dlogits)
        dx, dy = torch.empty_like(x), torch.empty_like(y)
        hd.dlogits, hd.sx.dx, hd.sy.dx = L.fptr(dlogits), L.fptr(dx), L.fptr(dy)
        L.check(L.lib().mmnas_head_bwd(C.byref(hd), L.stream()))
        ctx.keep = None
        _release_sinks(ctx, params)
        return dx, dy, None, None, None, None, None


def head_record(att_x, att_y, ln, proj, training):
    """(Head descriptor, params) for AttFlat modules att_x / att_y, the proj_norm LayerNorm and the proj Linear; None when
    a parameter's gradient does not live in a flat buffer.  Cached on the projection module (see _cached_record)."""
    first = att_x.mlp.fc.linear.weight
    if not _sinked((first,)):
        return None, None
    hd, params = _cached_record(proj, (training,), first, lambda: _head_record(att_x, att_y, ln, proj, training))
    if hd is None or not _sinked(params):   # EVERY parameter's gradient must (still) be its flat-buffer view: the cached
        return None, None                   # descriptor holds their raw pointers

    hd = L.Head.from_buffer_copy(hd)   # a private copy per call: several forwards may be alive before one backward (ITM triplets)
    if hd.drop_p > 0:
        hd.sx.seed = next_seed()
        hd.sy.seed = next_seed()
    return hd, params


def _head_record(att_x, att_y, ln, proj, training):
    params = []
    hd = L.Head()
    for side, af in ((hd.sx, att_x), (hd.sy, att_y)):
        fc, lin, mg = af.mlp.fc.linear, af.mlp.linear, af.linear_merge
        ps = [fc.weight, fc.bias, lin.weight, lin.bias, mg.weight, mg.bias]
        if not _sinked(ps):
            return None, None
        side.W1, side.b1, side.W2, side.b2, side.Wm, side.bm = [p.data_ptr() for p in ps]
        side.dW1, side.db1, side.dW2, side.db2, side.dWm, side.dbm = [p.grad.data_ptr() for p in ps]
        side.seed = 0   # (drawn per call by head_record)
        params += ps
    ps = [ln.a_2, ln.b_2, proj.weight, proj.bias]
    if not _sinked(ps):
        return None, None
    params += ps
    fc = att_x.mlp.fc
    drop = fc.dropout_r if training else 0.0
    hd.MID, hd.G, hd.OUT, hd.ANS = fc.linear.weight.shape[0], att_x.mlp.linear.weight.shape[0], proj.weight.shape[1], proj.weight.shape[0]
    hd.flags, hd.drop_p, hd.eps = (L.F_TRAIN if drop > 0 else 0), float(drop), float(ln.eps)
    hd.ln_a, hd.ln_b, hd.dln_a, hd.dln_b = ln.a_2.data_ptr(), ln.b_2.data_ptr(), ln.a_2.grad.data_ptr(), ln.b_2.grad.data_ptr()
    hd.Wp, hd.bp, hd.dWp, hd.dbp = proj.weight.data_ptr(), proj.bias.data_ptr(), proj.weight.grad.data_ptr(), proj.bias.grad.data_ptr()
    return hd, params


class BceLogitsSumFn(torch.autograd.Function):
    """BCEWithLogitsLoss(reduction='sum') (search_vqa.py:211, train_vqa.py:237): one kernel forward, one backward."""

    @staticmethod
    def forward(ctx, logits, target):
        logits, target = _f32c(logits), _f32c(target)
        loss = torch.zeros((), dtype=torch.float32, device=logits.device)
        L.check(L.lib().mmnas_bce_logits_sum_fwd(L.fptr(logits), L.fptr(target), L.fptr(loss), logits.numel(), L.stream()))
        ctx.save_for_backward(logits, target)
        return loss

    @staticmethod
    def backward(ctx, go):
        logits, target = ctx.saved_tensors
        go = _f32c(go)
        dl = torch.empty_like(logits)
        L.check(L.lib().mmnas_bce_logits_bwd(L.fptr(logits), L.fptr(target), L.fptr(go), L.fptr(dl), logits.numel(), L.stream()))
        return dl, None


def bce_with_logits_sum(logits, target):
    return BceLogitsSumFn.apply(logits, target)


class MixedSumFn(torch.autograd.Function):
    """out = sum_j gate[j] * o_j over the evaluated candidates of one supernet node (MixedOp.forward in modes 'full' /
    'two', mixed.py:59-68).  Only the active candidate's output is differentiated; every gate gets its gradient
    <dout, o_j>.  When the gate parameter's .grad is a row of the net's flat gate-gradient block (begin_arch_step) the
    kernel adds straight into it and autograd receives None."""

    @staticmethod
    def forward(ctx, gate, o_active, active, n, idx, *detached):
        lib = L.lib()
        o_active = _f32c(o_active)
        outs = [None] * n
        outs[active] = o_active
        for i, t in zip(idx, detached):
            outs[i] = _f32c(t)
        arr = (C.c_void_p * n)(*[L.fptr(t) for t in outs])
        g = _f32c(gate.detach())
        out = torch.empty_like(o_active)
        L.check(lib.mmnas_mixed_sum_fwd(arr, n, L.fptr(g), L.fptr(out), out.numel(), L.stream()))
        ctx.keep = (outs, g, arr)
        ctx.gate_param, ctx.active, ctx.n, ctx.n_detached = gate, active, n, len(detached)
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.keep is None:
            raise RuntimeError('MixedSumFn: backward ran a second time (its buffers are released after the first)')
        lib = L.lib()
        outs, g, arr = ctx.keep
        dout = _f32c(dout)
        gate = ctx.gate_param
        sink = getattr(gate, '_mmnas_gate_grad', None)
        if sink is not None and gate.grad is sink:
            dgate, ret = sink, None
        else:
            dgate = torch.zeros(ctx.n, dtype=torch.float32, device=dout.device)
            ret = dgate
        d_active = torch.empty_like(dout)
        ws = torch.empty(lib.mmnas_mixed_sum_ws_floats(), dtype=torch.float32, device=dout.device)
        L.check(lib.mmnas_mixed_sum_bwd(arr, ctx.n, L.fptr(g), L.fptr(dout), L.fptr(d_active), ctx.active, L.fptr(dgate),
                                        L.fptr(ws), dout.numel(), L.stream()))
        ctx.keep = None
        return (ret, d_active, None, None, None) + (None,) * ctx.n_detached


def mixed_sum(gate, outs, active):
    """outs: list over the node's candidates (None = not evaluated); outs[active] carries the autograd graph."""
    idx = tuple(i for i, t in enumerate(outs) if t is not None and i != active)
    return MixedSumFn.apply(gate, outs[active], active, len(outs), idx, *[outs[i] for i in idx])


def alpha_full_step(prob, gate_grad, m, v, prob_grad, lr, betas, eps, step):
    """All nodes' architecture update in one launch (mmnas_alpha_full_step): prob/gate_grad/m/v [nodes, width]."""
    L.check(L.lib().mmnas_alpha_full_step(L.fptr(prob), L.fptr(gate_grad), L.fptr(m), L.fptr(v), L.fptr(prob_grad),
                                          prob.shape[0], prob.shape[1], float(lr), float(betas[0]), float(betas[1]),
                                          float(eps), int(step), L.stream()))


def row_is_zero(feature):
    """make_mask (hygr_vqa.py:121-122) for a float feature tensor [..., d]: True where the whole row is zero."""
    f = _f32c(feature)
    d = f.shape[-1]
    rows = f.numel() // d
    out = torch.empty(f.shape[:-1], dtype=torch.uint8, device=f.device)
    L.check(L.lib().mmnas_row_is_zero(L.fptr(f), L.ptr(out), rows, d, L.stream()))
    return out.view(torch.bool)


def relation_embedding(bbox, nobj=None):
    """Box-geometry relation features of the reference's loaders (relation_embedding, load_data_vqa.py:224-239),
    batched on the GPU: bbox [B,S,4] float32 (x1,y1,x2,y2), nobj [B] int32 valid boxes per sample (None: all S)
    -> [B,S,S,4], zero outside the first nobj[b] rows/columns (the loaders' zero padding)."""
    bbox = _f32c(bbox)
    B, S, _ = bbox.shape
    out = torch.empty(B, S, S, 4, dtype=torch.float32, device=bbox.device)
    n = None if nobj is None else nobj.to(device=bbox.device, dtype=torch.int32).contiguous()
    L.check(L.lib().mmnas_relation_embedding(L.fptr(bbox), L.ptr(n), L.fptr(out), B, S, L.stream()))
    return out


def semantic_embedding(ques_ix, nwords, emb):
    """Token-relation features of the loaders (semantic_embedding, load_data_vqa.py:36-58), batched on the GPU:
    ques_ix [B,S] int64, nwords [B] (min(#words, S) per question), emb [V,E] float32 -> [B,S,S,3]."""
    ques_ix = ques_ix.contiguous()
    emb = _f32c(emb)
    B, S = ques_ix.shape
    n = nwords.to(device=ques_ix.device, dtype=torch.int32).contiguous()
    out = torch.empty(B, S, S, 3, dtype=torch.float32, device=ques_ix.device)
    L.check(L.lib().mmnas_semantic_embedding(L.ptr(ques_ix), L.ptr(n), L.fptr(emb), L.fptr(out), B, S, emb.shape[1], emb.shape[0],
                                             L.stream()))
    return out


def dropout_mask(n, p, seed, site, device):
    """Materialise the multiplier stream of one dropout site (tests / mask replay)."""
    out = torch.empty(n, dtype=torch.float32, device=device)
    L.check(L.lib().mmnas_dropout_mask(L.fptr(out), n, float(p), int(seed), int(site), L.stream()))
    return out

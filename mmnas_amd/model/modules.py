"""Candidate operators with the reference's class names, constructor signatures, call signature
``op(x, y, x_mask, y_mask, rel_embed)`` and state_dict keys (mmnas/model/modules.py), executing on
the MI355X through libmmnas_hip.so.  Parameters live in ordinary nn.Linear / nn.Conv1d
containers so checkpoints of the reference load unchanged; the containers are never
called -- their tensors are handed to the HIP operators in mmnas_amd.ops.

No CPU path exists here: a CPU tensor raises MMNasHipError at the HIP boundary.
"""
import torch
import torch.nn as nn

from .. import ops

__all__ = ['FC', 'MLP', 'LayerNorm', 'AttFlat', 'Identity', 'Zero', 'GELU', 'ReLU', 'LeakyReLU',
           'GatedLinear', 'GLU', 'MHAtt', 'RelMHAtt', 'SelfAtt', 'RelSelfAtt', 'GuidedAtt', 'FeedForward',
           'FeedForward_deep', 'UniimgAtt', 'SepConv', 'StdConv', 'RelHandle']


def _seed(mod, p):
    return ops.next_seed() if (mod.training and p > 0) else 0


class RelHandle:
    """Lazy relation embedding.  The reference materialises
    ``rel = relu(linear_y_rel(raw))`` as a [B,S,S,64] tensor in the stem (hygr_vqa.py:111,
    full_vqa.py:103) -- 164 MB per batch, read by every RelSelfAtt and written again as a gradient.
    The nets pass this handle through the cells instead (``rel_embed`` argument, consumed only by
    RelSelfAtt): the relation-bias kernel recomputes the embedding in registers from the 4-channel raw
    tensor and reduces linear_y_rel's gradient in-kernel.  ``materialize()`` gives the plain tensor for
    any consumer that wants one; a plain tensor is still accepted everywhere a handle is."""

    def __init__(self, raw, weight, bias):
        self.raw, self.weight, self.bias = raw, weight, bias
        self._dense = None

    def fusable(self, heads):
        from .. import _lib as L
        return bool(L.lib().mmnas_rel_fused_supported(self.weight.shape[1], self.weight.shape[0], heads))

    def materialize(self):
        if self._dense is None:
            self._dense = ops.linear(self.raw, self.weight, self.bias, relu=True)
        return self._dense


class FC(nn.Module):
    """Linear -> ReLU -> dropout (modules.py:13-31)."""

    def __init__(self, in_size, out_size, dropout_r=0., use_relu=True):
        super().__init__()
        self.dropout_r = dropout_r
        self.use_relu = use_relu
        self.linear = nn.Linear(in_size, out_size)

    def forward(self, x):
        h = ops.linear(x, self.linear.weight, self.linear.bias, relu=self.use_relu)
        if self.dropout_r > 0 and self.training:
            h = ops.drop_add(h, None, self.dropout_r, ops.next_seed(), 0)
        return h


class MLP(nn.Module):
    """FC followed by a Linear (modules.py:34-41)."""

    def __init__(self, in_size, mid_size, out_size, dropout_r=0., use_relu=True):
        super().__init__()
        self.fc = FC(in_size, mid_size, dropout_r=dropout_r, use_relu=use_relu)
        self.linear = nn.Linear(mid_size, out_size)

    def forward(self, x):
        return ops.linear(self.fc(x), self.linear.weight, self.linear.bias)


class LayerNorm(nn.Module):
    """a_2 * (x - mean) / (std_unbiased + eps) + b_2 (modules.py:44-56)."""

    def __init__(self, size, eps=1e-6, dim=-1):
        super().__init__()
        self.eps = eps
        self.dim = dim
        self.a_2 = nn.Parameter(torch.ones(size))
        self.b_2 = nn.Parameter(torch.zeros(size))

    def forward(self, x):
        if self.dim in (-1, x.dim() - 1):
            return ops.layer_norm(x, self.a_2, self.b_2, self.eps)
        # statistics over another axis (no operator does this; kept because the class accepts `dim`): normalise with
        # that axis moved last (unit scale, zero shift), then the reference's broadcast of a_2 / b_2 over the LAST axis
        xt = x.transpose(self.dim, -1).contiguous()
        one = torch.ones(xt.shape[-1], dtype=x.dtype, device=x.device)
        n = ops.layer_norm(xt, one, torch.zeros_like(one), self.eps).transpose(self.dim, -1)
        return self.a_2 * n + self.b_2


class AttFlat(nn.Module):
    """Attentional pooling head (modules.py:59-85): the two GEMM stages run on the HIP GEMM, the masked softmax
    over the sequence and the attention-weighted sum in one pooling kernel (mmnas_attflat_pool_*)."""

    def __init__(self, __C):
        super().__init__()
        self.glimpses = __C.ATTFLAT_GLIMPSES
        self.mlp = MLP(in_size=__C.HSIZE, mid_size=__C.ATTFLAT_MLP_SIZE, out_size=__C.ATTFLAT_GLIMPSES,
                       dropout_r=__C.DROPOUT_R, use_relu=True)
        self.linear_merge = nn.Linear(__C.HSIZE * __C.ATTFLAT_GLIMPSES, __C.ATTFLAT_OUT_SIZE)

    def forward(self, x, x_mask=None):
        pooled = ops.attflat_pool(self.mlp(x), x, x_mask)
        return ops.linear(pooled, self.linear_merge.weight, self.linear_merge.bias)


class Identity(nn.Module):
    """'skip_connect' (modules.py:88-93)."""

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        return x


class Zero(nn.Module):
    """'none' (modules.py:96-101): x * 0."""

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        return ops.eltwise(x, 0)


class ReLU(nn.Module):
    """'relu' registry entry (ops_adapter.py:27).  The reference registers a bare nn.ReLU whose
    forward takes one argument; this accepts the 5-argument cell signature as well."""

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        return ops.eltwise(x, 1)


class LeakyReLU(nn.Module):
    """'leakyrelu' registry entry (ops_adapter.py:29), slope 0.01."""

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        return ops.eltwise(x, 2)


class GELU(nn.Module):
    """tanh-form GELU (modules.py:104-109)."""

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        return ops.eltwise(x, 3)


class GatedLinear(nn.Module):
    """Linear to 2*out then nn.GLU (modules.py:112-119)."""

    def __init__(self, input_size, output_size):
        super().__init__()
        self.linear = nn.Linear(input_size, output_size * 2)

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None, relu=False, drop_p=0.0, seed=0):
        return ops.glu(ops.linear(x, self.linear.weight, self.linear.bias), relu=relu, drop_p=drop_p, seed=seed, site=0)


class _Wrapped(nn.Module):
    """dropout -> residual -> LayerNorm epilogue shared by the composed operators (modules.py:261-271)."""

    def _finish(self, x, core):
        p = self.drop_p if self.training else 0.0
        if self.residual or p > 0:
            z = ops.drop_add(core, x if self.residual else None, p, _seed(self, p), 1)
        else:
            z = core
        return self.ln(z) if self.norm else z


class GLU(_Wrapped):
    """gated_linear_{1,2} (modules.py:122-155)."""

    def __init__(self, __C, norm=False, residual=False, layers=1):
        super().__init__()
        assert layers in [1, 2]
        self.layers, self.norm, self.residual = layers, norm, residual
        self.drop_p = __C.DROPOUT_R
        if layers == 1:
            self.unit = GatedLinear(__C.HSIZE, __C.HSIZE)
        else:
            self.unit_0 = GatedLinear(__C.HSIZE, __C.HSIZE * 2)
            self.unit_1 = GatedLinear(__C.HSIZE * 2, __C.HSIZE)
        if norm:
            self.ln = LayerNorm(__C.HSIZE)

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        if self.layers == 1:
            core = self.unit(x)
        else:
            p = self.drop_p if self.training else 0.0
            core = self.unit_1(self.unit_0(x, relu=True, drop_p=p, seed=_seed(self, p)))
        return self._finish(x, core)


class MHAtt(nn.Module):
    """Parameter container + stand-alone multi-head attention (modules.py:158-199)."""

    def __init__(self, __C, base=64, hsize_k=None, bias=False):
        super().__init__()
        self.has_bias = bool(bias)   # (no reference call site enables it, modules.py:159; served by the composed path)
        self.HBASE = base
        self.HSIZE_INSIDE = int(__C.HSIZE * hsize_k) if hsize_k else __C.HSIZE
        assert self.HSIZE_INSIDE % self.HBASE == 0
        self.HHEAD = int(self.HSIZE_INSIDE / self.HBASE)
        self.drop_p = __C.DROPOUT_R
        self.linear_v = nn.Linear(__C.HSIZE, self.HSIZE_INSIDE, bias=bias)
        self.linear_k = nn.Linear(__C.HSIZE, self.HSIZE_INSIDE, bias=bias)
        self.linear_q = nn.Linear(__C.HSIZE, self.HSIZE_INSIDE, bias=bias)
        self.linear_merge = nn.Linear(self.HSIZE_INSIDE, __C.HSIZE, bias=bias)

    def run(self, xq, xkv, mask, rel, ln, norm, residual, training):
        if self.has_bias:
            raise ValueError('the fused attention operators have bias-free projections, as every registry entry does')
        lr = getattr(self, 'linear_r', None)
        wy = by = None
        if lr is not None and isinstance(rel, RelHandle):
            if rel.fusable(lr.weight.shape[0]):
                rel, wy, by = rel.raw, rel.weight, rel.bias      # bias computed in-kernel from the raw tensor
            else:
                rel = rel.materialize()
        return ops.attention_op(
            xq, xkv, mask, rel if lr is not None else None,
            self.linear_q.weight, self.linear_k.weight, self.linear_v.weight, self.linear_merge.weight,
            lr.weight if lr is not None else None, lr.bias if lr is not None else None,
            ln.a_2 if norm else None, ln.b_2 if norm else None,
            dh=self.HBASE, norm=norm, residual=residual, drop_p=self.drop_p, training=training,
            eps=ln.eps if norm else 1e-6, rel_Wy=wy, rel_by=by)

    def forward(self, v, k, q, mask=None):
        # stand-alone call: only the attention-map dropout applies (modules.py:197); the operator
        # wrappers below add the output dropout / residual / LayerNorm
        return _bare_attention(self, v, k, q, mask, None)


class RelMHAtt(MHAtt):
    """MHAtt plus the relation bias log(clamp(relu(linear_r(rel)), 1e-6)) (modules.py:202-245)."""

    def __init__(self, __C, base=64, hsize_k=None, bias=False):
        super().__init__(__C, base=base, hsize_k=hsize_k, bias=bias)
        self.linear_r = nn.Linear(__C.REL_SIZE, self.HHEAD, bias=True)

    def forward(self, v, k, q, mask=None, rel_embed=None):
        assert rel_embed is not None
        return _bare_attention(self, v, k, q, mask, rel_embed)


def _bare_attention(m, v, k, q, mask, rel):
    """MHAtt / RelMHAtt called directly (modules.py:178-199, 224-245): no output dropout / residual / LayerNorm.
    With key and value from the same tensor and no active dropout this is the fused operator; the general form --
    distinct key / value sources, or training mode, where ONLY the attention map is dropped (modules.py:197) -- is
    composed from the same kernels: three projections, (relation bias,) the attention core, the merge projection."""
    train_drop = m.training and m.drop_p > 0
    if v is k and not train_drop and not m.has_bias:
        return m.run(q, None if q is k else k, mask, rel, None, False, False, False)
    Q = ops.linear(q, m.linear_q.weight, m.linear_q.bias)
    K = ops.linear(k, m.linear_k.weight, m.linear_k.bias)
    V = ops.linear(v, m.linear_v.weight, m.linear_v.bias)
    biasT = None
    if rel is not None:
        if isinstance(rel, RelHandle):
            rel = rel.materialize()
        biasT = ops.rel_bias(rel, m.linear_r.weight, m.linear_r.bias)
    p = m.drop_p if train_drop else 0.0
    att = ops.mha_core(Q, K, V, mask, biasT, m.HBASE, p, ops.next_seed() if p > 0 else 0)
    return ops.linear(att, m.linear_merge.weight, m.linear_merge.bias)


class _AttWrapper(nn.Module):
    def __init__(self, __C, norm, residual):
        super().__init__()
        self.norm, self.residual = norm, residual
        if norm:
            self.ln = LayerNorm(__C.HSIZE)

    def _run(self, xq, xkv, mask, rel):
        return self.mhatt.run(xq, xkv, mask, rel, self.ln if self.norm else None, self.norm, self.residual,
                              self.training)


class SelfAtt(_AttWrapper):
    """LN(x + drop(MHAtt(x, x, x, x_mask))) (modules.py:248-271)."""

    def __init__(self, __C, norm=False, residual=False, base=64, hsize_k=None):
        super().__init__(__C, norm, residual)
        self.mhatt = MHAtt(__C, base=base, hsize_k=hsize_k)

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        return self._run(x, None, x_mask, None)


class RelSelfAtt(_AttWrapper):
    """Self-attention with relation bias (modules.py:274-298)."""

    def __init__(self, __C, norm=False, residual=False, base=64, hsize_k=None):
        super().__init__(__C, norm, residual)
        self.mhatt = RelMHAtt(__C, base=base, hsize_k=hsize_k)

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        assert rel_embed is not None
        return self._run(x, None, x_mask, rel_embed)


class GuidedAtt(_AttWrapper):
    """q = x, k = v = y, mask = y_mask (modules.py:301-325)."""

    def __init__(self, __C, norm=False, residual=False, base=64, hsize_k=None):
        super().__init__(__C, norm, residual)
        self.mhatt = MHAtt(__C, base=base, hsize_k=hsize_k)

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        assert y is not None
        return self._run(x, y, y_mask, None)


class UniimgAtt(_AttWrapper):
    """k = v = cat(x, y), no mask (modules.py:403-428)."""

    def __init__(self, __C, norm=False, residual=False, base=64, hsize_k=None):
        super().__init__(__C, norm, residual)
        self.mhatt = MHAtt(__C, base=base, hsize_k=hsize_k)

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        assert y is not None
        return self._run(x, torch.cat((x, y), dim=1), None, None)


class FeedForward(nn.Module):
    """LN(x + drop(W2 drop(relu(W1 x + b1)) + b2)) (modules.py:328-362)."""

    def __init__(self, __C, norm=False, residual=False, mid_k=None):
        super().__init__()
        self.norm, self.residual = norm, residual
        self.drop_p = __C.DROPOUT_R
        self.MID_SIZE = __C.HSIZE * mid_k if mid_k else __C.HSIZE * 4
        self.mlp = MLP(in_size=__C.HSIZE, mid_size=self.MID_SIZE, out_size=__C.HSIZE, dropout_r=__C.DROPOUT_R,
                       use_relu=True)
        if norm:
            self.ln = LayerNorm(__C.HSIZE)

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        m = self.mlp
        return ops.mlp_op(x, [m.fc.linear.weight, m.linear.weight], [m.fc.linear.bias, m.linear.bias],
                          self.ln.a_2 if self.norm else None, self.ln.b_2 if self.norm else None,
                          norm=self.norm, residual=self.residual, drop_p=self.drop_p, training=self.training,
                          eps=self.ln.eps if self.norm else 1e-6)


class FeedForward_deep(nn.Module):
    """Three-layer variant (modules.py:365-400)."""

    def __init__(self, __C, norm=False, residual=False, mid_k=None):
        super().__init__()
        self.norm, self.residual = norm, residual
        self.drop_p = __C.DROPOUT_R
        self.MID_SIZE = __C.HSIZE * mid_k if mid_k else __C.HSIZE * 2
        self.fc = FC(__C.HSIZE, self.MID_SIZE, dropout_r=__C.DROPOUT_R, use_relu=True)
        self.mlp = MLP(in_size=self.MID_SIZE, mid_size=self.MID_SIZE, out_size=__C.HSIZE, dropout_r=__C.DROPOUT_R,
                       use_relu=True)
        if norm:
            self.ln = LayerNorm(__C.HSIZE)

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        m = self.mlp
        return ops.mlp_op(x, [self.fc.linear.weight, m.fc.linear.weight, m.linear.weight],
                          [self.fc.linear.bias, m.fc.linear.bias, m.linear.bias],
                          self.ln.a_2 if self.norm else None, self.ln.b_2 if self.norm else None,
                          norm=self.norm, residual=self.residual, drop_p=self.drop_p, training=self.training,
                          eps=self.ln.eps if self.norm else 1e-6)


class SepConv(_Wrapped):
    """Depthwise k-tap conv over the sequence + pointwise conv (modules.py:431-462)."""

    def __init__(self, __C, norm=False, residual=False, k=3):
        super().__init__()
        self.norm, self.residual, self.k = norm, residual, k
        self.drop_p = __C.DROPOUT_R
        d = __C.HSIZE
        self.depthwise_conv = nn.Conv1d(d, d, kernel_size=k, groups=d, padding=k // 2, bias=True)
        self.pointwise_conv = nn.Conv1d(d, d, kernel_size=1, padding=0, bias=True)
        nn.init.kaiming_normal_(self.depthwise_conv.weight)
        nn.init.constant_(self.depthwise_conv.bias, 0.0)
        nn.init.kaiming_normal_(self.pointwise_conv.weight)
        nn.init.constant_(self.pointwise_conv.bias, 0.0)
        if norm:
            self.ln = LayerNorm(d)

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        t = ops.depthwise_conv_seq(x, self.depthwise_conv.weight, self.depthwise_conv.bias)
        d = x.shape[-1]
        core = ops.linear(t, self.pointwise_conv.weight.view(d, d), self.pointwise_conv.bias)
        return self._finish(x, core)


class StdConv(_Wrapped):
    """Dense k-tap Conv1d over the sequence (modules.py:465-491)."""

    def __init__(self, __C, norm=False, residual=False, k=3):
        super().__init__()
        self.norm, self.residual, self.k = norm, residual, k
        self.drop_p = __C.DROPOUT_R
        d = __C.HSIZE
        self.conv = nn.Conv1d(d, d, kernel_size=k, padding=k // 2, bias=True)
        nn.init.kaiming_normal_(self.conv.weight)
        nn.init.constant_(self.conv.bias, 0.0)
        if norm:
            self.ln = LayerNorm(d)

    def forward(self, x, y=None, x_mask=None, y_mask=None, rel_embed=None):
        core = ops.conv_seq(x, self.conv.weight, self.conv.bias)
        return self._finish(x, core)

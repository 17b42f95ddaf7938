"""TEST INFRASTRUCTURE ONLY.  CPU restatement (torch, fp32 or fp64) of the MMNas
candidate-operator hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file; the product package mmnas_amd/ never does.

Parity status: PINNED.  tests/golden/make_golden.py imports the reference
(/root/reference, this container only) and stores its outputs/gradients; the
not-gpu test-suite checks every function below against those vectors
(tests/test_oracle_golden.py).

Everything is functional: an operator is `op_forward(name, P, cfg, x, y, x_mask,
y_mask, rel, drops)` where `P` maps the reference's state_dict key (relative to
the operator module, e.g. "mhatt.linear_q.weight") to a tensor.  Gradients come
from torch autograd over these functions.  `drops` optionally maps a dropout
site name to an explicit multiplier tensor (mask replay, see dropout_rng.py);
without it dropout is the identity (eval mode / DROPOUT_R = 0).

Reference lines cited as modules.py:N are /root/reference/mmnas/model/modules.py.
"""
import math
import re
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# registry (ops_adapter.py:24-74) -- name -> (kind, hyper-parameters)
# ----------------------------------------------------------------------------

def parse_op_name(name):
    """Return (kind, kwargs) for a registry name of ops_adapter.py:24-74."""
    fixed = {
        'none': ('zero', {}), 'skip_connect': ('identity', {}), 'relu': ('relu', {}),
        'gelu': ('gelu', {}), 'leakyrelu': ('leakyrelu', {}),
        'feed_forward': ('ffn', {'mid_k': 4}), 'feed_forward_deep': ('ffn_deep', {'mid_k': 2}),
        'gated_linear_1': ('glu', {'layers': 1}), 'gated_linear_2': ('glu', {'layers': 2}),
    }
    if name in fixed:
        return fixed[name]
    m = re.fullmatch(r'(self_att|rel_self_att|guided_att|uniimg_att)_(\d+)(_2)?', name)
    if m:
        return m.group(1), {'base': int(m.group(2)), 'hsize_k': 2 if m.group(3) else 1}
    m = re.fullmatch(r'(sep_conv|std_conv)_(\d+)', name)
    if m:
        return m.group(1), {'k': int(m.group(2))}
    m = re.fullmatch(r'feed_forward_(\d+)', name)
    if m:
        return 'ffn', {'mid_k': int(m.group(1))}
    raise KeyError(name)


ALL_OP_NAMES = (
    ['none', 'skip_connect', 'relu', 'gelu', 'leakyrelu']
    + ['self_att_%s' % s for s in ('256', '128', '64', '32', '16', '64_2')]
    + ['rel_self_att_%d' % s for s in (256, 128, 64, 32, 16)]
    + ['guided_att_%s' % s for s in ('256', '128', '64', '32', '16', '64_2')]
    + ['uniimg_att_%d' % s for s in (128, 64, 32)]
    + ['sep_conv_%d' % k for k in (3, 5, 7, 11)]
    + ['std_conv_%d' % k for k in (3, 5, 7, 11)]
    + ['feed_forward_2', 'feed_forward', 'feed_forward_8', 'feed_forward_16', 'feed_forward_32',
       'gated_linear_1', 'gated_linear_2', 'feed_forward_deep']
)

USED_OPS = {  # ops_adapter.py:7-22
    'enc_safe': ['self_att_64', 'feed_forward'],
    'dec_safe': ['self_att_64', 'rel_self_att_64', 'guided_att_64', 'feed_forward'],
}
USED_OPS['enc'] = USED_OPS['enc_safe'] + ['none']
USED_OPS['dec'] = USED_OPS['dec_safe'] + ['none']


def op_param_shapes(name, cfg, norm=True):
    """state_dict key -> shape for one operator (verified against the reference's state_dict())."""
    kind, kw = parse_op_name(name)
    d = cfg.HSIZE
    sh = {}
    if kind in ('self_att', 'rel_self_att', 'guided_att', 'uniimg_att'):
        di = d * kw['hsize_k']
        for n in ('v', 'k', 'q'):
            sh['mhatt.linear_%s.weight' % n] = (di, d)
        if kind == 'rel_self_att':
            sh['mhatt.linear_r.weight'] = (di // kw['base'], cfg.REL_SIZE)
            sh['mhatt.linear_r.bias'] = (di // kw['base'],)
        sh['mhatt.linear_merge.weight'] = (d, di)
    elif kind == 'ffn':
        mid = d * kw['mid_k']
        sh['mlp.fc.linear.weight'] = (mid, d); sh['mlp.fc.linear.bias'] = (mid,)
        sh['mlp.linear.weight'] = (d, mid); sh['mlp.linear.bias'] = (d,)
    elif kind == 'ffn_deep':
        mid = d * kw['mid_k']
        sh['fc.linear.weight'] = (mid, d); sh['fc.linear.bias'] = (mid,)
        sh['mlp.fc.linear.weight'] = (mid, mid); sh['mlp.fc.linear.bias'] = (mid,)
        sh['mlp.linear.weight'] = (d, mid); sh['mlp.linear.bias'] = (d,)
    elif kind == 'glu':
        if kw['layers'] == 1:
            sh['unit.linear.weight'] = (2 * d, d); sh['unit.linear.bias'] = (2 * d,)
        else:
            sh['unit_0.linear.weight'] = (4 * d, d); sh['unit_0.linear.bias'] = (4 * d,)
            sh['unit_1.linear.weight'] = (2 * d, 2 * d); sh['unit_1.linear.bias'] = (2 * d,)
    elif kind == 'sep_conv':
        k = kw['k']
        sh['depthwise_conv.weight'] = (d, 1, k); sh['depthwise_conv.bias'] = (d,)
        sh['pointwise_conv.weight'] = (d, d, 1); sh['pointwise_conv.bias'] = (d,)
    elif kind == 'std_conv':
        sh['conv.weight'] = (d, d, kw['k']); sh['conv.bias'] = (d,)
    else:
        return {}
    if norm:
        sh['ln.a_2'] = (d,); sh['ln.b_2'] = (d,)
    return sh


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------

def layer_norm(x, a_2, b_2, eps=1e-6):
    """modules.py:52-56 -- Bessel-corrected std, eps added to the std (not the variance)."""
    mu = x.mean(-1, keepdim=True)
    c = x - mu
    sd = torch.sqrt((c * c).sum(-1, keepdim=True) / (x.shape[-1] - 1))
    return a_2 * c / (sd + eps) + b_2


def layer_norm_backward(x, a_2, dy, eps=1e-6):
    """Closed-form backward of layer_norm; used to pin the HIP kernel's formula."""
    n = x.shape[-1]
    mu = x.mean(-1, keepdim=True)
    c = x - mu
    sd = torch.sqrt((c * c).sum(-1, keepdim=True) / (n - 1))
    s = sd + eps
    g = dy * a_2
    dx = (g - g.mean(-1, keepdim=True)) / s - c * (g * c).sum(-1, keepdim=True) / ((n - 1) * sd * s * s)
    red = tuple(range(x.dim() - 1))
    da = (dy * c / s).sum(red)
    db = dy.sum(red)
    return dx, da, db


def _drop(t, drops, site):
    """drops: None (identity) | dict site -> multiplier tensor (mask replay) | float p (nn.Dropout with
    torch's own RNG, as the reference's modules do -- used for the timed CPU baseline)."""
    if drops is None:
        return t
    if isinstance(drops, float):
        return F.dropout(t, drops, training=True) if drops > 0 else t
    if site not in drops or drops[site] is None:
        return t
    return t * drops[site].reshape(t.shape)


def _linear(x, w, b=None):
    y = x.matmul(w.t())
    return y if b is None else y + b


def _heads(t, H, dh):
    B, S, _ = t.shape
    return t.reshape(B, S, H, dh).permute(0, 2, 1, 3)  # [B,H,S,dh]


def mh_att(P, pre, v_in, k_in, q_in, mask, base, rel=None, drops=None):
    """MHAtt.forward/att (modules.py:178-199) and RelMHAtt.forward (modules.py:224-245).

    `mask` is bool [B,1,1,S_k] (True = padded key) or None.  With `rel` given
    ([B,S_q,S_k,R]) the relation bias log(clamp(relu(linear_r(rel)),1e-6)) is added to
    the scaled scores BEFORE masking (modules.py:231-237).
    """
    Wv, Wk, Wq = P[pre + 'linear_v.weight'], P[pre + 'linear_k.weight'], P[pre + 'linear_q.weight']
    Wm = P[pre + 'linear_merge.weight']
    di = Wq.shape[0]
    H = di // base
    v = _heads(_linear(v_in, Wv), H, base)
    k = _heads(_linear(k_in, Wk), H, base)
    q = _heads(_linear(q_in, Wq), H, base)
    z = q.matmul(k.transpose(-1, -2)) / math.sqrt(base)
    if rel is not None:
        r = torch.relu(_linear(rel, P[pre + 'linear_r.weight'], P[pre + 'linear_r.bias']))  # [B,Sq,Sk,H]
        z = torch.log(torch.clamp(r.permute(0, 3, 1, 2), min=1e-6)) + z
    if mask is not None:
        z = z.masked_fill(mask, -1e9)
    a = _drop(torch.softmax(z, dim=-1), drops, 'att_map')
    o = a.matmul(v)  # [B,H,Sq,dh]
    B, _, Sq, _ = o.shape
    o = o.permute(0, 2, 1, 3).reshape(B, Sq, di)
    return _linear(o, Wm)


def _wrap(core, x, P, norm, residual, drops):
    """Common epilogue of every wrapped operator (modules.py:261-271 pattern)."""
    t = _drop(core, drops, 'out')
    z = x + t if residual else t
    return layer_norm(z, P['ln.a_2'], P['ln.b_2']) if norm else z


def gelu_tanh(x):
    """modules.py:109."""
    return 0.5 * x * (1 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))


def _glu_half(t):
    a, b = t.chunk(2, dim=-1)
    return a * torch.sigmoid(b)


def _conv1d_seq(x, w, b, groups):
    """Conv1d over the sequence axis of x[B,S,C] with 'same' zero padding (modules.py:438-452,472-481)."""
    k = w.shape[-1]
    return F.conv1d(x.transpose(1, 2), w, b, padding=k // 2, groups=groups).transpose(1, 2)


def op_forward(name, P, cfg, x, y=None, x_mask=None, y_mask=None, rel=None,
               norm=True, residual=True, drops=None):
    """Forward of registry operator `name` with the 5-argument cell signature (mixed.py:63,104)."""
    kind, kw = parse_op_name(name)
    if kind == 'zero':          # modules.py:96-101
        return x * 0.
    if kind == 'identity':      # modules.py:88-93
        return x
    if kind == 'relu':
        return torch.relu(x)
    if kind == 'leakyrelu':
        return F.leaky_relu(x, 0.01)
    if kind == 'gelu':
        return gelu_tanh(x)
    if kind == 'self_att':      # modules.py:260-271
        core = mh_att(P, 'mhatt.', x, x, x, x_mask, kw['base'], drops=drops)
    elif kind == 'rel_self_att':  # modules.py:286-298
        assert rel is not None
        core = mh_att(P, 'mhatt.', x, x, x, x_mask, kw['base'], rel=rel, drops=drops)
    elif kind == 'guided_att':  # modules.py:313-325  (q = x, k = v = y, mask = y_mask)
        assert y is not None
        core = mh_att(P, 'mhatt.', y, y, x, y_mask, kw['base'], drops=drops)
    elif kind == 'uniimg_att':  # modules.py:415-428  (k = v = cat(x, y), no mask)
        assert y is not None
        xy = torch.cat((x, y), dim=1)
        core = mh_att(P, 'mhatt.', xy, xy, x, None, kw['base'], drops=drops)
    elif kind == 'ffn':         # modules.py:351-362 with FC/MLP modules.py:13-41
        h = _drop(torch.relu(_linear(x, P['mlp.fc.linear.weight'], P['mlp.fc.linear.bias'])), drops, 'hid0')
        core = _linear(h, P['mlp.linear.weight'], P['mlp.linear.bias'])
    elif kind == 'ffn_deep':    # modules.py:389-400
        h0 = _drop(torch.relu(_linear(x, P['fc.linear.weight'], P['fc.linear.bias'])), drops, 'hid0')
        h1 = _drop(torch.relu(_linear(h0, P['mlp.fc.linear.weight'], P['mlp.fc.linear.bias'])), drops, 'hid1')
        core = _linear(h1, P['mlp.linear.weight'], P['mlp.linear.bias'])
    elif kind == 'glu':         # modules.py:141-155
        if kw['layers'] == 1:
            core = _glu_half(_linear(x, P['unit.linear.weight'], P['unit.linear.bias']))
        else:
            u = torch.relu(_glu_half(_linear(x, P['unit_0.linear.weight'], P['unit_0.linear.bias'])))
            u = _drop(u, drops, 'hid0')
            core = _glu_half(_linear(u, P['unit_1.linear.weight'], P['unit_1.linear.bias']))
    elif kind == 'sep_conv':    # modules.py:451-462
        t = _conv1d_seq(x, P['depthwise_conv.weight'], P['depthwise_conv.bias'], groups=x.shape[-1])
        core = _conv1d_seq(t, P['pointwise_conv.weight'], P['pointwise_conv.bias'], groups=1)
    elif kind == 'std_conv':    # modules.py:480-491
        core = _conv1d_seq(x, P['conv.weight'], P['conv.bias'], groups=1)
    else:
        raise KeyError(kind)
    return _wrap(core, x, P, norm, residual, drops)


def att_flat(P, x, x_mask, glimpses, drops=None):
    """AttFlat.forward (modules.py:73-85)."""
    h = _drop(torch.relu(_linear(x, P['mlp.fc.linear.weight'], P['mlp.fc.linear.bias'])), drops, 'hid0')
    att = _linear(h, P['mlp.linear.weight'], P['mlp.linear.bias'])  # [B,S,G]
    if x_mask is not None:
        att = att.masked_fill(x_mask.squeeze(1).squeeze(1).unsqueeze(2), -1e9)
    att = torch.softmax(att, dim=1)
    pooled = torch.cat([(att[:, :, g:g + 1] * x).sum(1) for g in range(glimpses)], dim=1)
    return _linear(pooled, P['linear_merge.weight'], P['linear_merge.bias'])


def relation_embedding(bbox):
    """Loader-side relation features (load_data_vqa.py:7-33 = load_data_vgd.py:7-33 = load_data_itm.py:5-31) for one
    sample: bbox [n,4] -> [n,n,4].  Pinned by tests/golden/loader.npz (the reference function, compiled from the
    loader's source by `ast` in the build container: the module itself imports en_vectors_web_lg and cannot be
    imported)."""
    x_min, y_min, x_max, y_max = torch.chunk(bbox, 4, dim=1)
    cx, cy = (x_min + x_max) * 0.5, (y_min + y_max) * 0.5
    w, h = (x_max - x_min) + 1., (y_max - y_min) + 1.
    dx = torch.log(torch.clamp(torch.abs((cx - cx.view(1, -1)) / w), min=1e-3))
    dy = torch.log(torch.clamp(torch.abs((cy - cy.view(1, -1)) / h), min=1e-3))
    dw = torch.log(w / w.view(1, -1))
    dh = torch.log(h / h.view(1, -1))
    return torch.stack((dx, dy, dw, dh), -1)


def semantic_embedding(ques_ix, pretrained_emb, size):
    """Token-relation features of one question (load_data_vqa.py:36-58): `size` = min(#words, max_token) leading
    tokens of ques_ix -> [size,size,3] = (L2 distance of the GloVe rows, their dot product over
    sqrt(|a|)*sqrt(|b|) + 1e-6 -- the reference takes the square root of the norms --, |i-j|/size)."""
    g = torch.as_tensor(pretrained_emb)[torch.as_tensor(ques_ix[:size], dtype=torch.long)].float()
    l2 = torch.norm(g.view(size, 1, -1) - g.view(1, size, -1), dim=-1)
    mod = torch.sqrt(torch.norm(g, dim=-1))
    cos = (g.view(size, 1, -1) * g.view(1, size, -1)).sum(-1) / (mod.view(size, 1) * mod.view(1, size) + 1e-6)
    pos = torch.arange(size).float()
    sub = torch.abs(pos.view(-1, 1) - pos.view(1, -1)) / size
    return torch.stack((l2, cos, sub), -1)


def tokenize(question, token_to_ix, max_token):
    """proc_ques (load_data_vqa.py:278-296): lower-case, strip punctuation, '-' and '/' to spaces, UNK for unknown
    words, zero padding.  Returns (ques_ix int64 [max_token], number of words)."""
    words = re.sub(r"([.,'!?\"()*#:;])", '', question.lower()).replace('-', ' ').replace('/', ' ').split()
    ix = np.zeros(max_token, np.int64)
    for i, w in enumerate(words[:max_token]):
        ix[i] = token_to_ix.get(w, token_to_ix['UNK'])
    return ix, len(words)


def pad_rows(feat, pad_size):
    """proc_img_feat (load_data_vqa.py:252-263)."""
    feat = np.asarray(feat)[:pad_size]
    return np.pad(feat, ((0, pad_size - feat.shape[0]), (0, 0)), mode='constant', constant_values=0)


def bbox_features(bbox, img_shape):
    """proc_bbox_feat (load_data_vqa.py:266-275); img_shape = (h, w)."""
    out = np.zeros((bbox.shape[0], 5), dtype=np.float32)
    out[:, 0] = bbox[:, 0] / float(img_shape[1])
    out[:, 1] = bbox[:, 1] / float(img_shape[0])
    out[:, 2] = bbox[:, 2] / float(img_shape[1])
    out[:, 3] = bbox[:, 3] / float(img_shape[0])
    out[:, 4] = (bbox[:, 2] - bbox[:, 0]) * (bbox[:, 3] - bbox[:, 1]) / float(img_shape[0] * img_shape[1])
    return out


def make_mask(feature):
    """hygr_vqa.py:121-122 -- True where the whole feature row is zero (padding)."""
    return (feature.abs().sum(-1) == 0).unsqueeze(1).unsqueeze(2)


def lstm_forward(P, pre, x):
    """Single-layer batch_first nn.LSTM (hygr_vqa.py:64-69) restated; gate order i,f,g,o (ATen)."""
    w_ih, w_hh = P[pre + 'weight_ih_l0'], P[pre + 'weight_hh_l0']
    b = P[pre + 'bias_ih_l0'] + P[pre + 'bias_hh_l0']
    B, S, _ = x.shape
    Hs = w_hh.shape[1]
    h = x.new_zeros(B, Hs)
    c = x.new_zeros(B, Hs)
    outs = []
    for t in range(S):
        g = x[:, t].matmul(w_ih.t()) + h.matmul(w_hh.t()) + b
        i, f, gg, o = g.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, dim=1)


# ----------------------------------------------------------------------------
# MixedOp algebra (mixed.py:59-208)
# ----------------------------------------------------------------------------

def mixed_forward(mode, outputs, alpha_gate, active_index, inactive_index):
    """mixed.py:59-68,103-104.  `outputs[i]` is candidate i's output (or None if not evaluated)."""
    if mode in ('full', 'two'):
        out = 0
        for i in active_index:
            out = out + alpha_gate[i] * outputs[i]
        for i in inactive_index:
            out = out + alpha_gate[i] * outputs[i].detach()
        return out
    return outputs[active_index[0]]


def alpha_prob_grad_full(alpha_prob, gate_grad):
    """mixed.py:194-198: dL/dalpha_i = sum_j g_j p_j (delta_ij - p_i)."""
    p = torch.softmax(alpha_prob, dim=0)
    gp = gate_grad * p
    return gp - p * gp.sum()


def alpha_prob_grad_two(alpha_prob, gate_grad, involved):
    """mixed.py:179-186: the same over the two sampled indices with their pairwise softmax."""
    idx = torch.as_tensor(involved, dtype=torch.long)
    p = torch.softmax(alpha_prob[idx], dim=0)
    gp = gate_grad[idx] * p
    out = torch.zeros_like(alpha_prob)
    out[idx] = gp - p * gp.sum()
    return out


def rescale_two(new_alpha, old_alpha_pair, involved):
    """mixed.py:200-208: keep the pair's total probability mass unchanged after the optimizer step."""
    idx = list(involved)
    offset = math.log(sum(math.exp(float(new_alpha[i])) for i in idx) / sum(math.exp(float(a)) for a in old_alpha_pair))
    out = new_alpha.clone()
    for i in idx:
        out[i] -= offset
    return out


# ----------------------------------------------------------------------------
# networks (hygr_*.py / full_*.py)
# ----------------------------------------------------------------------------

def _sub(P, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in P.items() if k.startswith(prefix)}


def net_forward(task, P, cfg, inputs, genotype=None, search=None, drops_for=None):
    """Net_Full.forward (full_vqa.py:85-114, full_vgd.py, full_itm.py) when `genotype` is given,
    Net_Search.forward (hygr_vqa.py:92-119, hygr_vgd.py, hygr_itm.py) when `search` is given.

    search = {'mode': None|'full'|'two', 'enc': [(active, inactive), ...], 'dec': [...]} with
    one entry per node; candidate lists are USED_OPS['enc_safe'/'dec_safe'] (hygr_vqa.py:20).
    P holds the full state_dict (keys as in the reference, including the 'backnone' spelling).
    drops_for(op_key) -> drops dict for that operator instance, or None.
    """
    frcn, bbox, y_rel, ques_ix, x_rel = inputs
    x_mask = make_mask(ques_ix.unsqueeze(2))
    y_mask = make_mask(frcn)
    lang = P['embedding.weight'][ques_ix]
    x = lstm_forward(P, 'lstm.', lang)
    if getattr(cfg, 'BBOX_FEATURE', False):
        frcn = torch.cat((frcn, _linear(bbox, P['bboxfeat_linear.weight'], P['bboxfeat_linear.bias'])), -1)
    y = _linear(frcn, P['imgfeat_linear.weight'], P['imgfeat_linear.bias'])
    if 'linear_x_rel.weight' in P:
        x_rel = torch.relu(_linear(x_rel, P['linear_x_rel.weight'], P['linear_x_rel.bias']))
    y_rel = torch.relu(_linear(y_rel, P['linear_y_rel.weight'], P['linear_y_rel.bias']))

    norm, residual = cfg.OPS_NORM, cfg.OPS_RESIDUAL

    def run_cell(kind, layer, s, pre, s_mask, pre_mask, rel):
        base = 'backnone.cells_%s.%d.dag.' % (kind, layer)
        if genotype is not None:
            for ni, node in enumerate(genotype[kind]):
                acc = 0
                for j, opname in enumerate(node):
                    key = '%s%d.%d.' % (base, ni, j)
                    dr = drops_for(key) if drops_for else None
                    acc = acc + op_forward(opname, _sub(P, key), cfg, s, pre, s_mask, pre_mask, rel,
                                           norm, residual, dr)
                s = acc
            return s
        names = USED_OPS[kind + '_safe']
        for ni, (active, inactive) in enumerate(search[kind]):
            key = '%s%d.0.' % (base, ni)
            gate = P[key + 'alpha_gate']
            outs = [None] * len(names)
            involved = list(active) + (list(inactive) if search['mode'] in ('full', 'two') else [])
            for i in involved:
                ck = key + 'candidate_ops.%d.' % i
                dr = drops_for(ck) if drops_for else None
                outs[i] = op_forward(names[i], _sub(P, ck), cfg, s, pre, s_mask, pre_mask, rel,
                                     norm, residual, dr)
            s = 0 + mixed_forward(search['mode'], outs, gate, list(active), list(inactive))
        return s

    for l in range(cfg.LAYERS):
        x = run_cell('enc', l, x, None, x_mask, None, x_rel)
    for l in range(cfg.LAYERS):
        y = run_cell('dec', l, y, x, y_mask, x_mask, y_rel)

    G = cfg.ATTFLAT_GLIMPSES
    xo = att_flat(_sub(P, 'attflat_x.'), x, x_mask, G, drops_for('attflat_x.') if drops_for else None)
    if task == 'vgd':  # full_vgd.py:105-114
        xo = xo.unsqueeze(1)
        yo = _linear(y, P['attfc_y.weight'], P['attfc_y.bias'])
        xy = layer_norm(xo + yo, P['proj_norm.a_2'], P['proj_norm.b_2'])
        scores = _linear(xy, P['proj_scores.weight'], P['proj_scores.bias']).squeeze(-1)
        if cfg.SCORES_LOSS == 'kld':
            scores = torch.log_softmax(scores, dim=-1)
        return scores, _linear(xy, P['proj_reg.weight'], P['proj_reg.bias'])
    yo = att_flat(_sub(P, 'attflat_y.'), y, y_mask, G, drops_for('attflat_y.') if drops_for else None)
    xy = layer_norm(xo + yo, P['proj_norm.a_2'], P['proj_norm.b_2'])
    out = _linear(xy, P['proj.weight'], P['proj.bias'])
    if task == 'itm':  # full_itm.py:105-112
        return torch.sigmoid(out.squeeze(-1))
    return out


def genotype_from_alphas(alphas_enc, alphas_dec):
    """Net_Search.genotype / parse (hygr_vqa.py:242-273): top-1 of alpha_prob per node."""
    def one(alphas, kind):
        return [[USED_OPS[kind][int(torch.argmax(a))]] for a in alphas]
    return {'enc': one(alphas_enc, 'enc'), 'dec': one(alphas_dec, 'dec')}


def bce_with_logits_sum(pred, target):
    """search_vqa.py:211 with REDUCTION='sum'."""
    return F.binary_cross_entropy_with_logits(pred, target, reduction='sum')


def default_cfg(**over):
    """A config namespace carrying every field the model code reads."""
    c = dict(HSIZE=512, DROPOUT_R=0.1, REL_SIZE=64, OPS_NORM=True, OPS_RESIDUAL=True, LAYERS=1,
             NODES={'enc': 12, 'dec': 18}, ATTFLAT_GLIMPSES=1, ATTFLAT_OUT_SIZE=1024,
             ATTFLAT_MLP_SIZE=512, FRCNFEAT_SIZE=2048, BBOX_FEATURE=False, BBOXFEAT_EMB_SIZE=1024,
             WORD_EMBED_SIZE=300, ALPHA_INIT_TYPE='normal', SCORES_LOSS='kld', GENOTYPE=None)
    c.update(over)
    return SimpleNamespace(**c)


# ----------------------------------------------------------------------------
# step harness: losses and the optimizer of the bilevel loop
# ----------------------------------------------------------------------------

def itm_bce_loss(scores_pos, scores_negc, scores_negi):
    """BCE_Loss (mmnas/utils/itm_loss.py:12-24, REDUCTION='sum'): the positive term is counted twice."""
    lp = F.binary_cross_entropy(scores_pos, torch.ones_like(scores_pos), reduction='sum')
    lc = F.binary_cross_entropy(scores_negc, torch.zeros_like(scores_negc), reduction='sum')
    li = F.binary_cross_entropy(scores_negi, torch.zeros_like(scores_negi), reduction='sum')
    return lp + lc + lp + li


def vgd_loss(pred_scores, pred_reg, scores, scores_mask, bbox, bbox_mask, lam=0.5):
    """train_vgd.py:316-333 with SCORES_LOSS='kld', REDUCTION='sum', LOSS_AVG=True, LOSS_LAMBDA=0.5: KLDiv of the masked
    log-scores against the masked soft labels / sum(mask) + lam * SmoothL1 of the masked boxes / sum(mask)."""
    ls = F.kl_div(pred_scores * scores_mask, scores * scores_mask, reduction='sum') / scores_mask.sum()
    lr = F.smooth_l1_loss(pred_reg * bbox_mask, bbox * bbox_mask, reduction='sum') / bbox_mask.sum()
    return ls + lam * lr, ls, lr


def answer_target(answers, ans_to_ix):
    """DataSet.proc_ans with get_score (load_data_vqa.py:299-333): soft target over the answer vocabulary from the
    annotators' (already normalised) answers -- 0 / .3 / .6 / .9 / 1 for 0 / 1 / 2 / 3 / >= 4 occurrences."""
    out = np.zeros(len(ans_to_ix), np.float32)
    counts = {}
    for a in answers:
        counts[a] = counts.get(a, 0) + 1
    for a, n in counts.items():
        if a in ans_to_ix:
            out[ans_to_ix[a]] = (0.0, 0.3, 0.6, 0.9)[n] if n < 4 else 1.0
    return out


def warmup_rate(lr_base, step, epoch_steps, warmup=True):
    """WarmupOptimizer.rate (mmnas/utils/optimizer.py:24-42)."""
    if warmup:
        for k in (1, 2, 3):
            if step <= int(epoch_steps * k):
                return lr_base * k / 4.
    return lr_base


def clip_grad_norm(grads, max_norm):
    """nn.utils.clip_grad_norm_ (search_vqa.py:298): scales the list in place, returns the total norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return float(total)


class Adam:
    """torch.optim.Adam's arithmetic (no amsgrad, no weight decay) over a dict of tensors with ONE global step count:
    what the reference loop amounts to, since its `0 * sum(p.sum())` terms (search_vqa.py:285-288) give every parameter
    -- also those of candidates that were not sampled -- a (zero) gradient at every step."""

    def __init__(self, params, betas, eps):
        self.params, self.betas, self.eps = params, betas, eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    @torch.no_grad()
    def step(self, grads, lr):
        self.t += 1
        b1, b2 = self.betas
        for k, p in self.params.items():
            g = grads.get(k)
            if g is None:
                g = torch.zeros_like(p)
            self.m[k].mul_(b1).add_(g, alpha=1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (self.v[k].sqrt() / math.sqrt(1 - b2 ** self.t)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-lr / (1 - b1 ** self.t))

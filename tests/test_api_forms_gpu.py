"""Call forms of the reference classes that no shipped operator uses but the classes accept (modules.py:44-56,
158-245): stand-alone MHAtt / RelMHAtt with distinct key and value sources, in training mode (only the attention map
is dropped, modules.py:197), with projection biases, and LayerNorm over an axis other than the last."""
import numpy as np
import pytest
import torch

from oracle import dropout_rng
from oracle import mmnas_oracle as O
from tests.golden import cases
from tests.util import TOL, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _torch_mhatt(P, v, k, q, mask, dh, rel=None, drop=None, bias=False):
    """modules.py:178-199 / 224-245 in float64."""
    B, Sq, _ = q.shape
    lin = lambda x, n: x @ P[n + '.weight'].T + (P[n + '.bias'] if bias else 0)
    heads = lambda t: t.view(B, -1, t.shape[-1] // dh, dh).transpose(1, 2)
    V, K, Q = heads(lin(v, 'linear_v')), heads(lin(k, 'linear_k')), heads(lin(q, 'linear_q'))
    Z = Q @ K.transpose(-2, -1) / np.sqrt(dh)
    if rel is not None:
        r = torch.relu(rel @ P['linear_r.weight'].T + P['linear_r.bias']).permute(0, 3, 1, 2)
        Z = torch.log(torch.clamp(r, min=1e-6)) + Z
    if mask is not None:
        Z = Z.masked_fill(mask, -1e9)
    A = torch.softmax(Z, -1)
    if drop is not None:
        A = A * drop
    out = (A @ V).transpose(1, 2).reshape(B, Sq, -1)
    return lin(out, 'linear_merge')


@pytest.mark.parametrize('variant', ['v_is_not_k', 'train_dropout', 'bias', 'rel_train'])
def test_standalone_mhatt_forms(variant):
    from mmnas.model.modules import MHAtt, RelMHAtt
    from mmnas_amd import ops
    rs = np.random.RandomState(11)
    cfg = cases.small_cfg(HSIZE=128, DROPOUT_R=0.25 if 'train' in variant else 0.0)
    rel_on = variant == 'rel_train'
    m = (RelMHAtt if rel_on else MHAtt)(cfg, base=64, bias=(variant == 'bias')).to(DEV)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.from_numpy((rs.standard_normal(tuple(p.shape)) / np.sqrt(p.shape[-1])).astype(np.float32)))
    B, Sq, Sk, d = 3, 7, 5, 128
    q = torch.from_numpy(rs.standard_normal((B, Sq, d)).astype(np.float32)).to(DEV).requires_grad_(True)
    if rel_on:
        Sk = Sq
        k = q
        v = q
    else:
        k = torch.from_numpy(rs.standard_normal((B, Sk, d)).astype(np.float32)).to(DEV).requires_grad_(True)
        v = torch.from_numpy(rs.standard_normal((B, Sk, d)).astype(np.float32)).to(DEV).requires_grad_(True) if variant == 'v_is_not_k' else k
    mask = torch.from_numpy(cases.masks(rs, B, Sk, full_pad_last=False)).to(DEV)
    rel = torch.from_numpy(np.maximum(rs.standard_normal((B, Sq, Sk, 64)), 0).astype(np.float32)).to(DEV) if rel_on else None
    gout = torch.from_numpy(rs.standard_normal((B, Sq, d)).astype(np.float32)).to(DEV)
    train = 'train' in variant
    m.train(train)
    drop = None
    if train:
        ops.manual_seed(123)
        seed = ops.next_seed()
        ops.manual_seed(123)                       # the module draws the same seed again
        keep = dropout_rng.scaled_mask(seed, 0, (B, 2, Sq, Sk), 0.25)
        drop = torch.from_numpy(keep).double().to(DEV)
    y = m(v, k, q, mask, rel) if rel_on else m(v, k, q, mask)
    (y * gout).sum().backward()
    P = {n: p.detach().double().requires_grad_(True) for n, p in m.named_parameters()}
    q64 = q.detach().double().requires_grad_(True)
    k64 = q64 if k is q else k.detach().double().requires_grad_(True)
    v64 = k64 if v is k else v.detach().double().requires_grad_(True)
    want = _torch_mhatt(P, v64, k64, q64, mask, 64, rel.double() if rel_on else None, drop, bias=(variant == 'bias'))
    (want * gout.double()).sum().backward()
    assert rel_err(y.detach().cpu().numpy(), want.detach().cpu().numpy()) < TOL
    assert rel_err(q.grad.cpu().numpy(), q64.grad.cpu().numpy()) < TOL
    if k is not q:
        assert rel_err(k.grad.cpu().numpy(), k64.grad.cpu().numpy()) < TOL
    if v is not k:
        assert rel_err(v.grad.cpu().numpy(), v64.grad.cpu().numpy()) < TOL
    # (the key bias shifts every score of a row by the same amount: its gradient is mathematically zero, round-off on
    #  both sides -- the denominator is floored at 1e-4 of the largest parameter gradient)
    gscale = max(float(P[n].grad.abs().max()) for n in P)
    for n, p in m.named_parameters():
        diff = float((p.grad.double() - P[n].grad).abs().max())
        assert diff <= TOL * max(float(P[n].grad.abs().max()), 1e-4 * gscale), n


def test_layernorm_over_another_axis():
    from mmnas.model.modules import LayerNorm
    rs = np.random.RandomState(2)
    x = torch.from_numpy(rs.standard_normal((3, 8, 16)).astype(np.float32)).to(DEV).requires_grad_(True)
    ln = LayerNorm(16, dim=1).to(DEV)
    with torch.no_grad():
        ln.a_2.copy_(torch.from_numpy((1 + 0.2 * rs.standard_normal(16)).astype(np.float32)))
        ln.b_2.copy_(torch.from_numpy((0.1 * rs.standard_normal(16)).astype(np.float32)))
    g = torch.from_numpy(rs.standard_normal((3, 8, 16)).astype(np.float32)).to(DEV)
    y = ln(x)
    (y * g).sum().backward()
    x64 = x.detach().double().requires_grad_(True)
    a, b = ln.a_2.detach().double().requires_grad_(True), ln.b_2.detach().double().requires_grad_(True)
    want = a * (x64 - x64.mean(1, keepdim=True)) / (x64.std(1, keepdim=True) + 1e-6) + b     # modules.py:52-56
    (want * g.double()).sum().backward()
    assert rel_err(y.detach().cpu().numpy(), want.detach().cpu().numpy()) < 1e-5
    assert rel_err(x.grad.cpu().numpy(), x64.grad.cpu().numpy()) < 1e-4
    assert rel_err(ln.a_2.grad.cpu().numpy(), a.grad.cpu().numpy()) < 1e-4
    assert rel_err(ln.b_2.grad.cpu().numpy(), b.grad.cpu().numpy()) < 1e-4

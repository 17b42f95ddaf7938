"""The reference loops' `0 * sum(p.sum() for p in ...)` lines as ONE autograd node (mmnas_amd/zeroterm.py): CPU tests of
the Parameter subclass and the lazy expression -- same loss, same gradients (a gradient for EVERY parameter, exact zeros
for the ones the forward did not use), everything else a parameter's `.sum()` can be asked still answers with a tensor;
stock DistributedDataParallel (gloo, world 2, find_unused_parameters=False -- the reference's setting) accepts it."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from mmnas_amd import zeroterm as Z


class _Part(nn.Sequential):
    """layers 3 and 4 take no part in the forward: the unsampled candidates of a supernet step"""

    def forward(self, x):
        return self[2](self[1](self[0](x)))


def _nets():
    torch.manual_seed(3)
    mk = lambda: _Part(nn.Linear(5, 4), nn.ReLU(), nn.Linear(4, 3), nn.Linear(3, 3), nn.LayerNorm(3))
    ref = mk()
    net = mk()
    net.load_state_dict(ref.state_dict())
    return Z.adopt(net), ref


def _step(n, x):
    """the scripts' statements (train_vqa.py:295-301) on a net whose last two layers take no part in the forward"""
    loss = n(x).pow(2).sum()
    loss += 0 * sum(p.sum() for p in n.parameters())
    n.zero_grad()
    loss.backward()
    return loss


def test_every_parameter_gets_the_literal_lines_gradient():
    net, ref = _nets()
    x = torch.randn(7, 5)
    for _ in range(2):
        a, b = _step(net, x), _step(ref, x)
        assert torch.equal(a.detach(), b.detach())
        for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            assert type(p) is Z.SumParameter and isinstance(p, nn.Parameter)
            assert p.grad is not None and torch.equal(p.grad, q.grad), k
    unused = [p for k, p in net.named_parameters() if k[0] in '34']
    assert unused and all(not p.grad.any() for p in unused)
    # the unused parameters' zeros are views of ONE zero-filled buffer (no clone per parameter)
    assert len({p.grad.untyped_storage().data_ptr() for p in unused}) == 1


def test_one_autograd_node_instead_of_thousands():
    net, ref = _nets()
    x = torch.randn(2, 5)

    def nodes(loss):
        seen, stack = set(), [loss.grad_fn]
        while stack:
            f = stack.pop()
            if f is None or f in seen:
                continue
            seen.add(f)
            stack += [g for g, _ in f.next_functions]
        return len(seen)
    la = net[0](x).sum()
    la = la + 0 * sum(p.sum() for p in net.parameters())
    lb = ref[0](x).sum()
    lb = lb + 0 * sum(p.sum() for p in ref.parameters())
    n_p = len(list(net.parameters()))
    assert nodes(lb) - nodes(la) >= 2 * n_p - 1        # a SumBackward + an AddBackward per parameter are gone


def test_torch_adam_and_clip_step_every_parameter():
    net, ref = _nets()
    x = torch.randn(7, 5)
    oa = torch.optim.Adam(net.parameters(), lr=0.01, betas=(0.9, 0.98), eps=1e-9)
    ob = torch.optim.Adam(ref.parameters(), lr=0.01, betas=(0.9, 0.98), eps=1e-9)
    for _ in range(3):
        _step(net, x), _step(ref, x)
        na = nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        nb = nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        assert torch.allclose(na, nb)
        oa.step(), ob.step()
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert torch.equal(p.data, q.data), k
    assert all(len(oa.state[p]) for p in net.parameters())      # Adam holds state for the unused parameters too


def test_lazy_sum_is_a_tensor_whenever_asked():
    net, ref = _nets()
    w, wr = net[0].weight, ref[0].weight
    t = w.sum()
    assert isinstance(t, Z.LazySum)
    assert float(t) == float(wr.sum()) and t.item() == wr.sum().item()
    assert torch.equal(w.sum(0), wr.sum(0)) and torch.equal(w.sum(dim=1, keepdim=True), wr.sum(dim=1, keepdim=True))
    assert torch.equal(torch.sum(w), wr.sum())                         # the function form is untouched
    assert torch.allclose(t * 2 + 1.0, wr.sum() * 2 + 1.0)
    assert torch.allclose(torch.ones(()) + t, 1 + wr.sum()) and torch.allclose(t + torch.ones(()), 1 + wr.sum())
    assert torch.allclose(t - 1, wr.sum() - 1) and torch.allclose(1 - t, 1 - wr.sum()) and torch.allclose(-t + 0.5, 0.5 - wr.sum())
    assert torch.allclose(torch.stack([t.detach(), torch.zeros(())]), torch.stack([wr.sum().detach(), torch.zeros(())]))
    # a real regulariser (non-zero scale) differentiates as usual
    loss = 0.5 * sum(p.sum() for p in net.parameters()) + torch.zeros(())
    loss.backward()
    assert all(torch.equal(p.grad, torch.full_like(p, 0.5)) for p in net.parameters())
    # no_grad / parameters that need no gradient: the value passes through
    with torch.no_grad():
        one = torch.ones(())
        one += 0 * sum(p.sum() for p in net.parameters())
        assert float(one) == 1.0
    # mixed scales fall back to tensors
    assert torch.allclose(net[0].bias.sum() * 2 + net[2].bias.sum(), ref[0].bias.sum() * 2 + ref[2].bias.sum())


def test_deepcopy_state_dict_and_switch(monkeypatch):
    net, ref = _nets()
    cp = copy.deepcopy(net)
    assert all(type(p) is Z.SumParameter for p in cp.parameters())
    assert all(type(v) is torch.Tensor for v in net.state_dict().values())
    ref.load_state_dict(net.state_dict())
    monkeypatch.setenv('MMNAS_ZERO_TERMS', '0')
    plain = Z.adopt(nn.Linear(2, 2))
    assert all(type(p) is nn.Parameter for p in plain.parameters())


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        net, ref = _nets()
        ddp = nn.parallel.DistributedDataParallel(net)         # find_unused_parameters=False, as search_vqa.py:210
        dref = nn.parallel.DistributedDataParallel(ref)
        x = torch.randn(7, 5, generator=torch.Generator().manual_seed(10 + rank))
        for _ in range(3):                                      # (an unused parameter without a gradient fails the SECOND forward)
            for d in (ddp, dref):
                loss = d(x).pow(2).sum()
                loss += 0 * sum(p.sum() for p in d.module.parameters())
                d.zero_grad()
                loss.backward()
        for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            assert p.grad is not None and torch.allclose(p.grad, q.grad, atol=1e-7), k
    finally:
        dist.destroy_process_group()


def test_stock_ddp_sees_every_parameter_ready():
    mp.spawn(_ddp_worker, args=(2, _free_port()), nprocs=2, join=True)


def test_torch_optimizers_keep_their_multi_tensor_path():
    """torch.optim chooses `foreach` only for exact parameter types it knows: SumParameter must be one of them (without it
    Adam over the supernet's ~900 parameters took 24 ms of host time per step instead of ~2)."""
    from torch.optim.optimizer import _default_to_fused_or_foreach
    net, _ = _nets()
    if torch.cuda.is_available():
        net = net.cuda()
        _, foreach = _default_to_fused_or_foreach(list(net.parameters()), differentiable=False, use_fused=False)
        assert foreach is True
    from torch.optim import optimizer as opt
    assert Z.SumParameter in opt._foreach_supported_types


def test_lazy_sum_inside_containers_and_under_python_operators():
    """torch.stack([p.sum() ...]) recursed forever (only top-level arguments were evaluated), and the operators
    Python resolves on the type (comparisons, **, abs, bool) raised TypeError.  Anything but the `0 * sum(...)` idiom is the
    ordinary tensor."""
    import torch
    from mmnas_amd import zeroterm
    lin = zeroterm.adopt(torch.nn.Linear(3, 2))
    ps = list(lin.parameters())
    want = torch.stack([torch.Tensor.sum(p) for p in ps])
    assert torch.equal(torch.stack([p.sum() for p in ps]), want)
    assert torch.equal(torch.cat([p.sum().reshape(1) for p in ps]), want)
    s0 = ps[0].sum()
    t0 = torch.Tensor.sum(ps[0])
    assert bool(s0 > -1e9) and bool(s0 >= t0) and bool(s0 <= t0) and not bool(s0 < t0) and bool(s0 == t0) and not bool(s0 != t0)
    assert torch.equal(s0 ** 2, t0 ** 2) and torch.equal(abs(s0), abs(t0)) and torch.equal(2.0 ** s0, 2.0 ** t0)
    assert torch.equal(1.0 / s0, 1.0 / t0) and bool(s0) == bool(t0) and int(s0) == int(t0)
    assert torch.equal(torch.maximum(s0, ps[1].sum()), torch.maximum(t0, torch.Tensor.sum(ps[1])))

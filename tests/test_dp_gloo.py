"""Data-parallel gradient exchange (mmnas_amd/dp.py) on 2 and 4 CPU processes over gloo: the same code
path that runs over RCCL on the GPUs, minus the HIP pack kernel (CPU tensors use the host copy).
The workers read the world size from MMNAS_GLOO_WORLD (the spawned interpreters import this module anew)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden import cases

WORLD = int(os.environ.get('MMNAS_GLOO_WORLD', '2'))
WORLDS = [2, 4]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(fn, port, world=2):
    old = os.environ.get('MMNAS_GLOO_WORLD')
    os.environ['MMNAS_GLOO_WORLD'] = str(world)
    try:
        mp.spawn(_entry, args=(fn, port), nprocs=world, join=True)
    finally:
        if old is None:
            os.environ.pop('MMNAS_GLOO_WORLD', None)
        else:
            os.environ['MMNAS_GLOO_WORLD'] = old


def _entry(rank, fn, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        globals()[fn](rank)
    finally:
        dist.destroy_process_group()


def _w_grad_reducer(rank):
    from mmnas_amd import dp
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 300), torch.nn.ReLU(), torch.nn.Linear(300, 40),
                              torch.nn.Linear(40, 7))
    dp.broadcast_parameters(net)
    red = dp.GradReducer(list(net.parameters()), bucket_mb=0.02)   # several buckets
    assert len(red.buckets) >= 2
    x = torch.randn(5, 16, generator=torch.Generator().manual_seed(10 + rank))
    ref = torch.nn.Sequential(torch.nn.Linear(16, 300), torch.nn.ReLU(), torch.nn.Linear(300, 40), torch.nn.Linear(40, 7))
    ref.load_state_dict(net.state_dict())
    for _ in range(2):   # two steps: buffers are re-armed
        red.begin_step()
        net(x).pow(2).sum().backward()
        red.finish()
    # expected: mean over ranks of the per-rank gradients
    grads = []
    for r in range(WORLD):
        xr = torch.randn(5, 16, generator=torch.Generator().manual_seed(10 + r))
        ref.zero_grad()
        ref(xr).pow(2).sum().backward()
        grads.append([p.grad.clone() for p in ref.parameters()])
    for i, p in enumerate(net.parameters()):
        want = sum(g[i] for g in grads) / WORLD
        assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), i
        assert p.grad.data_ptr() == red.fg.views[i].data_ptr()   # gradients live in the flat buffer


def _w_supernet_reducer(rank):
    from mmnas_amd import dp
    from mmnas.model import mixed
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas.model.mixed import MixedOp
    c = cases.net_case('vqa', None, 3, search=True, HSIZE=64)
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    torch.manual_seed(100 + rank)            # different initial weights per rank ...
    net = Net_Search(c['cfg'], init)
    dp.broadcast_parameters(net)             # ... made identical, as DDP's constructor does
    w0 = net.imgfeat_linear.weight.detach().clone()
    t = w0.clone()
    dist.broadcast(t, 0)
    assert torch.equal(t, w0)
    red = dp.SupernetReducer(net)
    MixedOp.MODE = None
    mixed.seed_arch_sampler(888)             # same seed on every rank -> same sampled architecture
    net.reset_binary_gates()
    assert dp.check_same_architecture(net)
    red.begin_weight_step()
    sampled, unsampled = [], []
    for m in net.redundant_modules:
        for i, op in enumerate(m.candidate_ops):
            (sampled if i in m.active_index else unsampled).extend(op.parameters())
    assert all(p.grad is None for p in unsampled)          # "avoid over-regularization", mixed.py:160-163
    live = list(red.shared) + sampled
    assert all(p.grad is not None for p in live)
    for k, p in enumerate(live):                            # stand-in for backward: rank-dependent grads
        p.grad.fill_(float(rank + 1) * (1 + k % 5))
    red.finish_weight_step()
    for k, p in enumerate(live):
        assert torch.allclose(p.grad, torch.full_like(p.grad, 0.5 * (WORLD + 1) * (1 + k % 5))), k
    assert all(p.grad is None for p in unsampled)
    # arch step: only the alpha_gate gradients travel
    for m in net.redundant_modules:
        m.alpha_gate.grad = torch.full((m.n_choices,), float(rank))
    red.reduce_alpha_gate_grads()
    for m in net.redundant_modules:
        assert torch.allclose(m.alpha_gate.grad, torch.full((m.n_choices,), 0.5 * (WORLD - 1)))
    # a rank that sampled differently is detected
    if rank == 1:
        m = net.redundant_modules[0]
        m.set_active([1 - m.active_index[0]], m.active_index)
    assert not dp.check_same_architecture(net)


def _w_zero_grad_between_begin_and_backward(rank):
    """The reference's order (search_vqa.py:290: net.zero_grad() right before loss.backward()): the views attached by
    begin_step() are dropped, autograd allocates fresh gradient tensors -- they must be adopted into the flat buffer
    before the all-reduce, not silently left out of it."""
    from mmnas_amd import dp
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 300), torch.nn.ReLU(), torch.nn.Linear(300, 40), torch.nn.Linear(40, 7))
    dp.broadcast_parameters(net)
    red = dp.GradReducer(list(net.parameters()), bucket_mb=0.02)
    x = torch.randn(5, 16, generator=torch.Generator().manual_seed(10 + rank))
    red.begin_step()
    net.zero_grad()                                     # set_to_none=True: every p.grad is None now
    net(x).pow(2).sum().backward()
    assert any(red._launched)                           # buckets still go out from the backward thread
    red.finish()
    ref = torch.nn.Sequential(torch.nn.Linear(16, 300), torch.nn.ReLU(), torch.nn.Linear(300, 40), torch.nn.Linear(40, 7))
    ref.load_state_dict(net.state_dict())
    grads = []
    for r in range(WORLD):
        xr = torch.randn(5, 16, generator=torch.Generator().manual_seed(10 + r))
        ref.zero_grad()
        ref(xr).pow(2).sum().backward()
        grads.append([p.grad.clone() for p in ref.parameters()])
    for i, p in enumerate(net.parameters()):
        want = sum(g[i] for g in grads) / WORLD
        assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), i
        assert p.grad.data_ptr() == red.fg.views[i].data_ptr()


class _FakeNode(torch.nn.Module):
    def __init__(self, d, n):
        super().__init__()
        self.candidate_ops = torch.nn.ModuleList([torch.nn.Linear(d, d) for _ in range(n)])
        self.alpha_prob = torch.nn.Parameter(torch.zeros(n))
        self.alpha_gate = torch.nn.Parameter(torch.zeros(n))
        self.n_choices = n
        self.active_index, self.inactive_index = [0], list(range(1, n))

    def forward(self, x):
        return x + torch.tanh(self.candidate_ops[self.active_index[0]](x))


class _FakeSupernet(torch.nn.Module):
    """A torch-only stand-in with the surface SupernetReducer reads (redundant_modules, net_parameters, the head's
    attribute names): the real Net_Search only runs on the GPU."""

    def __init__(self, d=24, nodes=6):
        super().__init__()
        self.imgfeat_linear = torch.nn.Linear(10, d)                  # stem
        self.backnone = torch.nn.ModuleList([_FakeNode(d, 2 + (i % 3)) for i in range(nodes)])
        self.proj_norm = torch.nn.LayerNorm(d)                        # head
        self.proj = torch.nn.Linear(d, 5)
        self.linear_y_rel = torch.nn.Linear(4, 8)                     # stem parameter that gets no gradient here

    @property
    def redundant_modules(self):
        return list(self.backnone)

    def net_parameters(self):
        return [p for n, p in self.named_parameters() if 'alpha' not in n]

    def forward(self, x):
        h = self.imgfeat_linear(x)
        for node in self.backnone:
            h = node(h)
        return self.proj(self.proj_norm(h))


def _w_supernet_reducer_overlapped_buckets(rank):
    from mmnas_amd import dp
    torch.manual_seed(5)
    net = _FakeSupernet()
    dp.broadcast_parameters(net)
    red = dp.SupernetReducer(net, n_buckets=3)
    assert red.comm and red.n_buckets == 3
    assert sorted(k for b in red.bucket_nodes for k in b) == list(range(6))
    assert 5 in red.bucket_nodes[0] and 0 in red.bucket_nodes[2]      # backward order: last node first
    ref = _FakeSupernet()
    ref.load_state_dict(net.state_dict())
    for step in range(3):
        for k, (m, mr) in enumerate(zip(net.redundant_modules, ref.redundant_modules)):
            a = (step + k) % m.n_choices                              # same sample on both ranks
            m.active_index = mr.active_index = [a]
            m.inactive_index = mr.inactive_index = [i for i in range(m.n_choices) if i != a]
        red.begin_weight_step()
        if step == 2:
            net.zero_grad()                                           # the reference's order: adopt the strays
        x = torch.randn(7, 10, generator=torch.Generator().manual_seed(100 * step + rank))
        net(x).pow(2).sum().backward()
        launched_early = list(red._launched)
        red.finish_weight_step()
        assert launched_early[0] and not launched_early[2]            # head bucket left during backward, the stem's at the end
        grads = []
        for r in range(WORLD):
            xr = torch.randn(7, 10, generator=torch.Generator().manual_seed(100 * step + r))
            ref.zero_grad()
            ref(xr).pow(2).sum().backward()
            grads.append({n: (p.grad.clone() if p.grad is not None else None) for n, p in ref.named_parameters()})
        for n, p in net.named_parameters():
            if 'alpha' in n:
                continue
            if grads[0][n] is None:
                assert p.grad is None or not torch.any(p.grad != 0), n
                continue
            want = sum(g[n] for g in grads) / WORLD
            assert p.grad is not None and torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), (step, n)
    # arch step: gate gradients in the flat block travel in place
    block = torch.zeros(6, 4)
    net._flat_grads = (block, torch.zeros(6, 4))
    for i, m in enumerate(net.redundant_modules):
        row = block[i, :m.n_choices]
        row.fill_(float(rank + i))
        m.alpha_gate.grad = row
        m.alpha_gate._mmnas_gate_grad = row
    red.reduce_alpha_gate_grads()
    for i, m in enumerate(net.redundant_modules):
        assert torch.allclose(m.alpha_gate.grad, torch.full((m.n_choices,), 0.5 * (WORLD - 1) + i))
        assert m.alpha_gate.grad.data_ptr() == block[i].data_ptr()


@pytest.mark.parametrize('world', WORLDS)
def test_grad_reducer_buckets_overlap_and_average(world):
    _run('_w_grad_reducer', _free_port(), world)


@pytest.mark.parametrize('world', WORLDS)
def test_supernet_reducer_sampled_segments_and_alpha(world):
    _run('_w_supernet_reducer', _free_port(), world)


@pytest.mark.parametrize('world', WORLDS)
def test_zero_grad_between_begin_step_and_backward(world):
    _run('_w_zero_grad_between_begin_and_backward', _free_port(), world)


@pytest.mark.parametrize('world', WORLDS)
def test_supernet_reducer_overlapped_buckets_real_backward(world):
    _run('_w_supernet_reducer_overlapped_buckets', _free_port(), world)


def test_single_process_is_a_noop():
    from mmnas_amd import dp
    net = torch.nn.Linear(4, 3)
    red = dp.GradReducer(list(net.parameters()))
    red.begin_step()
    net(torch.ones(2, 4)).sum().backward()
    red.finish()
    assert net.weight.grad is not None and torch.allclose(net.weight.grad, torch.full((3, 4), 2.0))


def test_sink_live_count_several_forwards_one_backward():
    """train_itm.py:380-391 runs three forwards before the one backward, so three section nodes (ops.BackboneFn / HeadFn)
    hold every parameter.  A sink reports the parameter's gradient as complete only when the LAST of them has released it
    (the first to finish holds a third of it), and a new step (FlatGrads.zero) forgets nodes that never ran a backward."""
    from mmnas_amd import dp
    ps = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(3, 2))]
    fg = dp.FlatGrads(ps)
    arrived = []
    fg.enable_sinks(None, arrived.append, None)
    s0, s1 = ps[0]._mmnas_sink, ps[1]._mmnas_sink
    for _ in range(3):
        s0.acquire()
    s1.acquire()
    assert s0.live() == 3 and s1.live() == 1
    assert not s0.release() and not s0.release()
    assert s1.release()
    assert s0.live() == 1
    assert s0.release() and s0.live() == 0
    s0.acquire()                 # a forward whose backward never runs (an evaluation pass under grad mode) ...
    fg.zero()                    # ... is forgotten when the next step begins
    assert s0.live() == 0
    s0.acquire()
    assert s0.release()
    assert s0.release()          # (an unmatched release -- a node that did not count itself -- stays at zero)

"""(1) The reference's OWN statement sequence (search_vqa.py:279-337) around this repo's Net_Search: stock
torch DistributedDataParallel, the `0 * sum(p.sum())` terms, net.zero_grad(), clip_grad_norm_, torch.optim.Adam behind
WarmupOptimizer, unused_modules_off/back -- two ranks on the box's one GPU over gloo (RCCL refuses two ranks on one
device).  What is checked is what DDP promises: after backward every rank holds the average of the per-rank
gradients, and both ranks take the same optimizer step.
(2) The RCCL path itself in a one-rank group: init_process_group('nccl', device_id=...), GradReducer and
SupernetReducer with forced collectives (ReduceOp.AVG, comm stream, bucket launches from the backward thread, the
alpha-gate block)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden import cases

pytestmark = pytest.mark.gpu
WORLD = 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = torch.from_numpy


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _entry(rank, fn, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        globals()[fn](rank)
    finally:
        dist.destroy_process_group()


def _build(c):
    from mmnas.model.hygr_vqa import Net_Search
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = Net_Search(c['cfg'], init)
    net.load_state_dict({k: T(v) for k, v in c['P'].items()})
    return net.cuda().train()


def _w_reference_loop(rank):
    from torch.nn.parallel import DistributedDataParallel as DDP
    from mmnas.model.mixed import MixedOp
    from mmnas.utils.optimizer import WarmupOptimizer
    cs = [cases.net_case('vqa', None, 600 + r, search=True, HSIZE=64) for r in range(WORLD)]
    for c in cs:
        c['P'] = cs[0]['P']                               # same weights, per-rank batches
    plan_w = cases.search_plan(np.random.RandomState(1), None)
    plan_a = cases.search_plan(np.random.RandomState(2), 'full')
    flat_w, flat_a = plan_w['enc'] + plan_w['dec'], plan_a['enc'] + plan_a['dec']
    loss_fn = torch.nn.BCEWithLogitsLoss(reduction='sum')
    batch = lambda r: (tuple(T(a).cuda() for a in cs[r]['inputs']), T(cs[r]['target']).cuda())

    # expected: per-rank gradients from a plain (non-DDP) copy, averaged
    def plain_grads(mode, flat):
        out = []
        for r in range(WORLD):
            ref = _build(cs[0])
            MixedOp.MODE = mode
            ref.set_sampled(flat)
            inp, tgt = batch(r)
            loss_fn(ref(inp), tgt).backward()
            out.append({k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in ref.named_parameters()})
            MixedOp.MODE = None
        return {k: (sum(g[k] for g in out) / WORLD if out[0][k] is not None else None) for k in out[0]}

    want_w = plain_grads(None, flat_w)
    want_a = plain_grads('full', flat_a)

    net = _build(cs[0])
    ddp = DDP(net, device_ids=[0])                        # search_vqa.py:210
    net_optim = WarmupOptimizer(4e-4, torch.optim.Adam(ddp.module.net_parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9),
                                epoch_steps=10, warmup=True)
    alpha_optim = torch.optim.Adam(ddp.module.alpha_prob_parameters(), 0.1, betas=(0.0, 0.999))
    _, tgt = batch(rank)
    inp = tuple(T(a) for a in cs[rank]['inputs'])       # CPU tensors, as the reference's loader yields them: DDP moves them
    named = dict(ddp.module.named_parameters())          # (before unused_modules_off() hides the idle candidates)

    # ---- network step, search_vqa.py:279-300 (sampling replaced by an injected sample, as in the goldens) ----
    MixedOp.MODE = None
    ddp.module.set_sampled(flat_w)
    ddp.module.unused_modules_off()
    pred = ddp(inp)
    loss = loss_fn(pred, tgt)
    loss += 0 * sum(p.sum() for p in ddp.module.alpha_prob_parameters())
    loss += 0 * sum(p.sum() for p in ddp.module.alpha_gate_parameters())
    loss += 0 * sum(p.sum() for p in ddp.module.net_parameters())
    ddp.zero_grad()
    loss.backward()
    top = max(float(g.abs().max()) for g in want_w.values() if g is not None)
    for k, g in want_w.items():
        if 'alpha' in k:
            continue
        have = named[k].grad
        assert have is not None, k                        # the 0 * sum terms give every parameter a gradient
        ref = g if g is not None else torch.zeros_like(have)
        assert float((have - ref).abs().max()) <= 1e-4 * top, (k, float((have - ref).abs().max()))
    torch.nn.utils.clip_grad_norm_(ddp.module.net_parameters(), 1.0)
    before = named['proj.weight'].detach().clone()
    net_optim.step()
    ddp.module.unused_modules_back()
    assert float((named['proj.weight'] - before).abs().max()) > 0
    chk = named['proj.weight'].detach().clone()
    dist.broadcast(chk, 0)
    assert torch.equal(chk, named['proj.weight'].detach())      # both ranks took the same step

    # ---- arch step, search_vqa.py:317-336 ----
    MixedOp.MODE = 'full'
    try:
        ddp.module.set_sampled(flat_a)
        ddp.module.unused_modules_off()
        pred = ddp(inp)
        loss = loss_fn(pred, tgt)
        loss += 0 * sum(p.sum() for p in ddp.module.alpha_prob_parameters())
        loss += 0 * sum(p.sum() for p in ddp.module.net_parameters())
        ddp.zero_grad()
        loss.backward()
    finally:
        pass
    # the weights moved by one Adam step since want_a was computed: compare the gate gradients loosely, exactly across ranks
    gates = torch.stack([torch.nn.functional.pad(m.alpha_gate.grad, (0, 4 - m.n_choices)) for m in ddp.module.redundant_modules])
    want = torch.stack([torch.nn.functional.pad(want_a[k], (0, 4 - want_a[k].numel())) for k in want_a if k.endswith('alpha_gate')])
    assert float((gates - want).abs().max()) <= 5e-2 * float(want.abs().max())
    g0 = gates.clone()
    dist.broadcast(g0, 0)
    assert torch.equal(g0, gates)
    ddp.module.set_arch_param_grad()
    a0 = torch.stack([torch.nn.functional.pad(p.detach(), (0, 4 - p.numel())) for p in ddp.module.alpha_prob_parameters()]).clone()
    alpha_optim.step()
    ddp.module.unused_modules_back()
    MixedOp.MODE = None
    a1 = torch.stack([torch.nn.functional.pad(p.detach(), (0, 4 - p.numel())) for p in ddp.module.alpha_prob_parameters()])
    assert float((a1 - a0).abs().max()) > 0


def test_reference_statement_sequence_under_stock_ddp():
    mp.spawn(_entry, args=('_w_reference_loop', _free_port()), nprocs=WORLD, join=True)


NCCL_CHILD = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from tests.golden import cases
from tests.util import REL_PATH_SELF_TOL, is_rel_path
from mmnas_amd import dp
from mmnas_amd.harness import SearchLoop
from mmnas.model.hygr_vqa import Net_Search
from mmnas.model.full_vqa import Net_Full
T = torch.from_numpy
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == 'nccl'

def build(cls, c):
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = cls(c['cfg'], init)
    net.load_state_dict({k: T(v) for k, v in c['P'].items()})
    return net.to(dev).train()

loss_fn = torch.nn.BCEWithLogitsLoss(reduction='sum')
# ---- GradReducer: bucketed, overlapped, ReduceOp.AVG over RCCL ----
c = cases.net_case('vqa', 'mmnas_vqa', 4242, HSIZE=64, B=3, Sx=6, Sy=9)
net = build(Net_Full, c)
inp = tuple(T(a).to(dev) for a in c['inputs']); tgt = T(c['target']).to(dev)
loss_fn(net(inp), tgt).backward()
plain = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
for p in net.parameters():
    p.grad = None
red = dp.GradReducer(list(net.parameters()), bucket_mb=0.05, force_collectives=True)
assert red.comm and len(red.buckets) >= 3 and red.comm_stream is not None
for it in range(2):
    red.begin_step()
    loss_fn(net(inp), tgt).backward()
    assert any(red._launched), 'no bucket was launched from the backward thread'
    red.finish()
torch.cuda.synchronize()
# (the backbone is ONE native call: buckets completed inside it wait on events the chain records behind their last
#  operator, not on the end of the call)
assert getattr(red, 'marks_made', 0) >= 2, getattr(red, 'marks_made', 0)
for k, p in net.named_parameters():     # (relation path: the chain's one launch for all relation operators vs one per operator)
    if k in plain:
        assert float((p.grad - plain[k]).abs().max()) <= (REL_PATH_SELF_TOL if is_rel_path(k) else 1e-5) * float(plain[k].abs().max() + 1e-12), k
# begin_step -> net.zero_grad() -> backward (the reference's order, search_vqa.py:290): gradients land outside the flat
# buffer and must still be reduced
red.begin_step()
net.zero_grad()
loss_fn(net(inp), tgt).backward()
red.finish()
torch.cuda.synchronize()
for k, p in net.named_parameters():
    if k in plain:
        assert p.grad.data_ptr() == red.fg.views[red.fg.index[id(p)]].data_ptr(), k
        assert float((p.grad - plain[k]).abs().max()) <= (REL_PATH_SELF_TOL if is_rel_path(k) else 1e-5) * float(plain[k].abs().max() + 1e-12), k
red.fg.disable_sinks()
# ---- SupernetReducer through SearchLoop: weight step (3 overlapped buckets) + arch step (gate block) ----
c = cases.net_case('vqa', None, 77, search=True, HSIZE=64)
plan = cases.search_plan(np.random.RandomState(3), None)
flat = plan['enc'] + plan['dec']
ref = build(Net_Search, c)
ref.set_sampled(flat)
inp = tuple(T(a).to(dev) for a in c['inputs']); tgt = T(c['target']).to(dev)
loss_fn(ref(inp), tgt).backward()
plain = {k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.grad is not None and 'alpha' not in k}
net = build(Net_Search, c)
loop = SearchLoop(net, force_collectives=True)
r = loop.reducer
assert r.comm and r.n_buckets == 3
loss = loop.weight_step(inp, tgt, optimize=False, plan=flat)
torch.cuda.synchronize()
assert all(r._launched) and getattr(r, 'marks_made', 0) >= 1, getattr(r, 'marks_made', 0)
named = dict(net.named_parameters())
top = max(float(g.abs().max()) for g in plain.values())
for k, g in plain.items():      # (the loop's net takes the backbone / head chains, `ref` the per-operator path: same
    assert named[k].grad is not None, k     # kernels, other summation orders in the bias / weight gradients)
    assert float((named[k].grad - g).abs().max()) <= (REL_PATH_SELF_TOL if is_rel_path(k) else 1e-4) * max(float(g.abs().max()), 1e-3 * top), k
pa = cases.search_plan(np.random.RandomState(4), 'full')
loss = loop.arch_step(inp, tgt, plan=pa['enc'] + pa['dec'])
torch.cuda.synchronize()
gg, pg = net._flat_grads
assert float(gg.abs().max()) > 0 and bool(torch.isfinite(gg).all()) and float(pg.abs().max()) > 0
r.fg.disable_sinks()
dist.destroy_process_group()
print('NCCL_OK')
'''


def test_rccl_backend_one_rank_full_steps():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-c', NCCL_CHILD % {'root': ROOT}], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0 and 'NCCL_OK' in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])

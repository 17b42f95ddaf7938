"""The data-parallel reducers on DEVICE tensors with two ranks: both processes share the one GPU of the test box and
talk over gloo (RCCL refuses two ranks on one device), so everything except the collective's transport is the
production path -- flat gradient buffer, HIP kernels writing into it (sinks), bucket launches from the backward
thread, the comm stream and its events, the supernet pack/unpack kernel with the table in the kernel arguments."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden import cases

pytestmark = pytest.mark.gpu
WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _entry(rank, fn, port, world=2):
    global WORLD
    WORLD = world          # (spawned children import this module afresh: the rank count travels as an argument)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        globals()[fn](rank)
    finally:
        dist.destroy_process_group()


def _build_full(c):
    from mmnas.model.full_vqa import Net_Full
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = Net_Full(c['cfg'], init)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in c['P'].items()})
    return net.cuda().train()


def _loss(net, c):
    inp = tuple(torch.from_numpy(a).cuda() for a in c['inputs'])
    tgt = torch.from_numpy(c['target']).cuda()
    return torch.nn.functional.binary_cross_entropy_with_logits(net(inp), tgt, reduction='sum')


def _w_full(rank):
    from mmnas_amd import dp
    cs = [cases.net_case('vqa', 'mmnas_vqa', 500 + r, HSIZE=64, B=3, Sx=6, Sy=9) for r in range(WORLD)]
    for c in cs:
        c['cfg'].DROPOUT_R = 0.0
        c['P'] = cs[0]['P']                     # same weights on both ranks, different batches
    net = _build_full(cs[rank])
    dp.broadcast_parameters(net)
    red = dp.GradReducer(list(net.parameters()), bucket_mb=0.05)
    assert len(red.buckets) >= 3 and red.comm_stream is not None
    # the word embedding's gradient travels as (token, dy row) pairs, not as a dense table (dp.RowExchange)
    assert red.row_exchange is not None and red.fg.params[red.row_exchange.i] is net.embedding.weight
    assert all(red.row_exchange.i not in idxs for _, _, idxs in red.buckets)
    # expected: mean over ranks of the plain autograd gradients
    ref = _build_full(cs[0])
    want = None
    for r in range(WORLD):
        ref.zero_grad()
        _loss(ref, cs[r]).backward()
        g = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in ref.named_parameters()}
        want = g if want is None else {k: want[k] + g[k] for k in g}
    scale = max(float(v.abs().max()) for v in want.values()) / WORLD
    for step in range(6):   # (a bucket reduced too early shows up intermittently: repeat)
        red.begin_step()
        _loss(net, cs[rank]).backward()
        assert all(red._seen), 'a parameter never reported its gradient'
        red.finish()
        torch.cuda.synchronize()
        assert red.row_exchange.done                       # the rows went through the exchange, not the dense fallback
        both = [torch.empty_like(net.embedding.weight.grad) for _ in range(WORLD)]
        dist.all_gather(both, net.embedding.weight.grad)
        assert all(torch.equal(both[0], b) for b in both[1:])   # bitwise equal on the ranks, as an all-reduce's result is
        bad = []
        for k, p in net.named_parameters():
            err = float((p.grad - want[k] / WORLD).abs().max())
            if not err <= 1e-4 * max(float(want[k].abs().max()) / WORLD, 1e-3 * scale):
                bad.append((step, k, err, float(want[k].abs().max()) / WORLD))
        assert not bad, bad[:8]


def _w_supernet(rank):
    from mmnas_amd import dp
    from mmnas_amd.model import mixed
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas.model.mixed import MixedOp
    cs = [cases.net_case('vqa', None, 700 + r, search=True, HSIZE=64, B=2) for r in range(WORLD)]
    init = {'token_size': cs[0]['token_size'], 'ans_size': cs[0]['ans_size'],
            'pretrained_emb': np.zeros((cs[0]['token_size'], cs[0]['cfg'].WORD_EMBED_SIZE), np.float32)}
    cs[rank]['cfg'].DROPOUT_R = 0.0
    torch.manual_seed(1)
    net = Net_Search(cs[rank]['cfg'], init).cuda().train()
    dp.broadcast_parameters(net)
    red = dp.SupernetReducer(net)
    assert red.row_exchange is not None and red.fg.params[red.row_exchange.i] is net.embedding.weight
    mixed.seed_arch_sampler(321)                 # same samples on both ranks
    MixedOp.MODE = None
    net.reset_binary_gates()
    assert dp.check_same_architecture(net)
    red.begin_weight_step()
    net.unused_modules_off()
    _loss(net, cs[rank]).backward()
    local = red.fg.flat.clone()
    red.finish_weight_step()
    net.unused_modules_back()
    torch.cuda.synchronize()
    # every rank's flat buffer must now hold the mean of the two local buffers on the exchanged (active) segments
    both = [torch.zeros_like(local) for _ in range(WORLD)]
    dist.all_gather(both, local)
    mean = sum(both[1:], both[0]) / WORLD
    segs = red.exchanged_segments()
    # the word embedding is not part of any bucket: its rows were exchanged inside backward (so `local` above may already
    # hold them -- it is checked separately below)
    rx = red.row_exchange
    eo, en = red.fg.offsets[rx.i], net.embedding.weight.numel()
    assert rx.done and all(not (o < eo + en and eo < o + n) for o, n in segs)
    emb_avg = red.fg.flat[eo:eo + en].clone()
    segs = segs + [(eo, en)]
    mean[eo:eo + en] = emb_avg
    local[eo:eo + en] = emb_avg
    assert sum(n for _, n in segs) >= sum(p.numel() for p in red._active)
    covered = 0
    for o, n in segs:
        assert torch.allclose(red.fg.flat[o:o + n], mean[o:o + n], rtol=2e-5, atol=2e-7)
        covered += n
    assert covered < red.fg.total                # unsampled candidates were not exchanged ...
    mask = torch.ones(red.fg.total, dtype=torch.bool, device='cuda')
    for o, n in segs:
        mask[o:o + n] = False
    assert torch.equal(red.fg.flat[mask], local[mask])   # ... and stay untouched
    # the exchanged embedding gradient = the mean of the ranks' dense embedding gradients: the same step once more with
    # the row exchange switched off (the table's gradient then stays local: it is in no bucket)
    red.row_exchange = None
    red.begin_weight_step()
    net.unused_modules_off()
    _loss(net, cs[rank]).backward()
    torch.cuda.synchronize()
    mine = red.fg.flat[eo:eo + en].clone()
    red.finish_weight_step()
    net.unused_modules_back()
    parts = [torch.zeros_like(mine) for _ in range(WORLD)]
    dist.all_gather(parts, mine)
    want = sum(parts[1:], parts[0]) / WORLD
    assert float(want.abs().max()) > 0
    assert float((emb_avg - want).abs().max()) <= 1e-5 * float(want.abs().max())
    both_e = [torch.zeros_like(emb_avg) for _ in range(WORLD)]
    dist.all_gather(both_e, emb_avg)
    assert all(torch.equal(both_e[0], b) for b in both_e[1:])   # bitwise equal on the ranks


def _w_arch_then_weight(rank):
    """the row exchange of the embedding gradient runs on the communication stream, which only the
    weight step joins (begin_weight_step .. finish_weight_step).  An arch step in between (its weight gradients are
    never read; search_vqa.py:331 steps alpha_optim only) must not take that path: nothing of it may still be writing
    into the flat buffer when the next weight step zeroes it.  The weight step after an arch step must give exactly the
    embedding gradient of the same weight step without one."""
    from mmnas_amd.harness import SearchLoop
    from mmnas_amd.model import mixed
    from mmnas.model.hygr_vqa import Net_Search
    cs = [cases.net_case('vqa', None, 900 + r, search=True, HSIZE=64, B=2) for r in range(WORLD)]
    init = {'token_size': cs[0]['token_size'], 'ans_size': cs[0]['ans_size'],
            'pretrained_emb': np.zeros((cs[0]['token_size'], cs[0]['cfg'].WORD_EMBED_SIZE), np.float32)}
    cs[rank]['cfg'].DROPOUT_R = 0.0
    torch.manual_seed(2)
    net = Net_Search(cs[rank]['cfg'], init).cuda().train()
    loop = SearchLoop(net, epoch_steps=10)
    red, rx = loop.reducer, loop.reducer.row_exchange
    assert rx is not None and not rx.active
    inp = tuple(torch.from_numpy(a).cuda() for a in cs[rank]['inputs'])
    tgt = torch.from_numpy(cs[rank]['target']).cuda()
    eo, en = red.fg.offsets[rx.i], net.embedding.weight.numel()
    mixed.seed_arch_sampler(99)
    net.reset_binary_gates()
    plan = [(list(m.active_index), list(m.inactive_index)) for m in net.redundant_modules]
    loop.weight_step(inp, tgt, optimize=False, plan=plan)
    torch.cuda.synchronize()
    assert rx.done and not rx.active
    want = red.fg.flat[eo:eo + en].clone()
    assert float(want.abs().max()) > 0
    for _ in range(4):
        calls = []
        orig = rx.exchange
        rx.exchange = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        loop.arch_step(inp, tgt, optimize=False, plan=plan)
        assert not calls and not rx._keep and not rx._idx and not rx.active      # local scatter-add, main stream
        loop.weight_step(inp, tgt, optimize=False, plan=plan)
        rx.exchange = orig
        assert len(calls) == 1
        torch.cuda.synchronize()
        assert torch.equal(red.fg.flat[eo:eo + en], want)
    both = [torch.zeros_like(want) for _ in range(WORLD)]
    dist.all_gather(both, want)
    assert all(torch.equal(both[0], b) for b in both[1:])


def _w_itm_triplet(rank):
    """the ITM triplet step runs three forwards before one backward, so three BackboneFn and three
    HeadFn nodes add into the same gradient views.  A bucket must be all-reduced after the LAST of them, not the first
    (which holds a third of the gradient): two ranks, small buckets, against the mean of plain per-rank gradients."""
    from mmnas_amd import dp, ops
    from mmnas_amd.harness import BCE_Loss, itm_triplet_step
    from mmnas.model.full_itm import Net_Full
    pos = [cases.net_case('itm', 'mmnas_itm', 600 + r, HSIZE=64, B=3, Sx=6, Sy=9) for r in range(WORLD)]
    neg = [cases.net_case('itm', 'mmnas_itm', 650 + r, HSIZE=64, B=3, Sx=6, Sy=9) for r in range(WORLD)]
    c0 = pos[0]
    c0['cfg'].DROPOUT_R = 0.0
    init = {'token_size': c0['token_size'], 'ans_size': c0['ans_size'],
            'pretrained_emb': np.zeros((c0['token_size'], c0['cfg'].WORD_EMBED_SIZE), np.float32)}

    def build():
        net = Net_Full(c0['cfg'], init)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in c0['P'].items()})
        return net.cuda().train()

    tup = lambda c: tuple(torch.from_numpy(a).cuda() for a in c['inputs'])
    ref = build()
    want = None
    for r in range(WORLD):       # plain autograd, per-operator path, no reducer
        ref.zero_grad()
        itm_triplet_step(ref, BCE_Loss(), tup(pos[r]), tup(neg[r]))
        g = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in ref.named_parameters()}
        want = g if want is None else {k: want[k] + g[k] for k in g}
    scale = max(float(v.abs().max()) for v in want.values()) / WORLD
    net = build()
    red = dp.GradReducer(list(net.parameters()), bucket_mb=0.05)
    assert len(red.buckets) >= 3
    early = []
    launch = red._launch

    def checked_launch(b):       # every launch from the backward thread: no live section node may still hold the bucket
        if not red._launched[b]:
            early.extend((b, i) for i in red.buckets[b][2] if red.fg.params[i]._mmnas_sink.live() > 0)
        launch(b)
    red._launch = checked_launch
    for step in range(4):
        itm_triplet_step(net, BCE_Loss(), tup(pos[rank]), tup(neg[rank]), reducer=red)
        torch.cuda.synchronize()
        assert not early, early[:6]
        bad = []
        for k, p in net.named_parameters():
            err = float((p.grad - want[k] / WORLD).abs().max())
            if not err <= 1e-4 * max(float(want[k].abs().max()) / WORLD, 1e-3 * scale):
                bad.append((step, k, err, float(want[k].abs().max()) / WORLD))
        assert not bad, bad[:8]
    assert getattr(red, 'marks_made', 0) > 0     # the chain path ran and placed its bucket events (once per step)


@pytest.mark.parametrize('fn', ['_w_full', '_w_supernet', '_w_itm_triplet', '_w_arch_then_weight'])
def test_two_ranks_on_one_gpu(fn):
    mp.spawn(_entry, args=(fn, _free_port()), nprocs=WORLD, join=True)


@pytest.mark.parametrize('fn', ['_w_full', '_w_supernet', '_w_arch_then_weight'])
def test_two_ranks_on_one_gpu_ragged_decoder_stream(fn, monkeypatch):
    """The same exchanges with the ragged decoder stream on (MMNAS_UNPAD=1 in the ranks): every rank packs its own batch
    (different row counts per rank), the gradients that travel are the same."""
    monkeypatch.setenv('MMNAS_UNPAD', '1')
    mp.spawn(_entry, args=(fn, _free_port()), nprocs=WORLD, join=True)


# World 8 on ONE device is opt-in (MMNAS_TEST_WORLD8=1): with eight HIP processes sharing the box's one GPU a process is
# aborted by the runtime with HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION in about one run of three -- with this round's kernels
# switched off as well as on (profiles/r06_world8_on_one_gpu.txt: MMNAS_MHA_BWD_B16=0/1, MMNAS_LSTM=0), never with four
# processes and never with one -- and on another box of the pool not once in 28 runs of eight: a property of the box (eight
# processes' queues oversubscribe one device and are time-sliced by wave save / restore, which is not how eight ranks on eight
# GPUs run), so the default suite must not depend on it.  When it does not abort, world 8 passes every assertion below.
WORLDS = [4, 8] if os.environ.get('MMNAS_TEST_WORLD8') == '1' else [4]


@pytest.mark.parametrize('world', WORLDS)
@pytest.mark.parametrize('fn', ['_w_full', '_w_supernet', '_w_arch_then_weight'])
def test_four_and_eight_ranks_on_one_gpu(fn, world):
    """the N = 4 and N = 8 LOGIC -- the reference runs DistributedDataParallel over all 8 GPUs of a node
    (search_vqa.py:58-59, 210, 279-337) and no multi-GPU box is reachable from here, so the world-4 / world-8 code paths (the
    row exchange of 8 ranks' embedding rows, the bucket marks with 8 ranks, 8 flat buffers reduced to one mean) run as 4 / 8
    fresh processes sharing the box's one GPU over gloo: averaged gradients equal the mean of the ranks' plain gradients,
    embedding tables are bitwise equal across all ranks, every rank samples the same architecture."""
    mp.spawn(_entry, args=(fn, _free_port(), world), nprocs=world, join=True)

"""Fused optimizer step and direct gradient sinks (GPU)."""
import numpy as np
import pytest
import torch

from tests.golden import cases
from tests.util import rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda'
T = torch.from_numpy


def test_flat_adam_matches_torch_adam_with_clipping_and_skipped_params():
    """FlatAdam == clip_grad_norm_ + torch.optim.Adam (search_vqa.py:296-300), including a parameter whose
    gradient is None on some steps (absent_grads='skip': the parameter is frozen, its moments and step count too)."""
    from mmnas_amd.optim import FlatAdam, WarmupOptimizer
    rs = np.random.RandomState(0)
    shapes = [(64, 33), (7,), (128, 128), (5, 3, 2)]
    init = [rs.standard_normal(s).astype(np.float32) for s in shapes]
    mine = [torch.nn.Parameter(T(a.copy()).to(DEV)) for a in init]
    ref = [torch.nn.Parameter(T(a.copy())) for a in init]
    opt = WarmupOptimizer(4e-4, FlatAdam(mine, lr=0, betas=(0.9, 0.98), eps=1e-9, absent_grads='skip'), epoch_steps=2, warmup=True, max_norm=1.0)
    ropt = torch.optim.Adam(ref, lr=0, betas=(0.9, 0.98), eps=1e-9)
    fg = opt.optimizer.fg
    for step in range(6):
        skip = {2} if step in (1, 3) else set()
        grads = [rs.standard_normal(s).astype(np.float32) * (3.0 if step % 2 else 0.01) for s in shapes]
        fg.zero()
        live = [p for i, p in enumerate(mine) if i not in skip]
        fg.attach(live)
        for i, p in enumerate(mine):
            if i in skip:
                continue
            if i == 1:
                p.grad = T(grads[i]).to(DEV)        # a gradient produced outside the flat buffer is adopted
            else:
                p.grad.copy_(T(grads[i]).to(DEV))
        for i, p in enumerate(ref):
            p.grad = None if i in skip else T(grads[i].copy())
        torch.nn.utils.clip_grad_norm_([p for p in ref if p.grad is not None], 1.0)
        lr = opt.rate(step + 1)
        for g in ropt.param_groups:
            g['lr'] = lr
        ropt.step()
        opt.step()
        assert abs(opt._rate - lr) < 1e-12
    for a, b in zip(mine, ref):
        assert rel_err(a.detach().cpu().numpy(), b.detach().numpy()) < 1e-5
    assert opt.optimizer.steps == [6, 6, 4, 6]
    assert [opt.rate(k) for k in (1, 3, 5, 7)] == pytest.approx([1e-4, 2e-4, 3e-4, 4e-4])   # optimizer.py:26-36


def test_gradient_sinks_equal_autograd_accumulation():
    """Backward kernels writing straight into the flat gradient buffer (dp.FlatGrads sinks) give the same
    gradients as the plain autograd path, for every parameter of a fixed-architecture network."""
    from mmnas.model.full_vqa import Net_Full
    from mmnas_amd import dp
    c = cases.net_case('vqa', 'mmnas_vqa', 4242)
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = Net_Full(c['cfg'], init)
    net.load_state_dict({k: T(v) for k, v in c['P'].items()})
    net = net.to(DEV).train()
    inp = tuple(T(a).to(DEV) for a in c['inputs'])
    tgt = T(c['target']).to(DEV)

    def run():
        loss = torch.nn.functional.binary_cross_entropy_with_logits(net(inp), tgt, reduction='sum')
        loss.backward()

    run()
    plain = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    red = dp.GradReducer(list(net.parameters()))
    n_sink = sum(hasattr(p, '_mmnas_sink') for p in net.parameters())
    assert n_sink == len(list(net.parameters()))
    for _ in range(2):          # twice: the buffer is re-zeroed, nothing accumulates across steps
        red.begin_step()
        run()
        red.finish()
    for k, p in net.named_parameters():
        if k not in plain:
            assert not torch.any(p.grad != 0), k
            continue
        assert p.grad.data_ptr() == red.fg.views[red.fg.index[id(p)]].data_ptr(), k
        # (the two paths cut their reductions differently -- pair launches, stream-K pieces, float atomics -- and twelve
        #  encoder operators of 10 rows amplify the round-off: 1.4e-5 seen once on an encoder projection)
        assert rel_err(p.grad.cpu().numpy(), plain[k].cpu().numpy()) < 3e-5, k
    red.fg.disable_sinks()


def test_supernet_step_with_reducer_and_fused_optimizer():
    """One weight step + one arch step of the bilevel loop (search_vqa.py:279-337) on the supernet with the
    single-GPU reducer and the fused optimizer: runs, updates only sampled candidates, alphas move."""
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas.model.mixed import MixedOp
    from mmnas_amd import dp
    from mmnas_amd.model import mixed
    from mmnas_amd.optim import FlatAdam, WarmupOptimizer
    c = cases.net_case('vqa', None, 77, search=True, HSIZE=64)
    c['cfg'].DROPOUT_R = 0.1
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = Net_Search(c['cfg'], init).to(DEV).train()
    red = dp.SupernetReducer(net)
    net_optim = WarmupOptimizer(4e-4, FlatAdam(red.fg.params, betas=(0.9, 0.98), eps=1e-9, grads=red.fg, absent_grads='skip'), 10, True, max_norm=1.0)
    alpha_optim = torch.optim.Adam(list(net.alpha_prob_parameters()), 0.1, betas=(0.0, 0.999))
    inp = tuple(T(a).to(DEV) for a in c['inputs'])
    tgt = T(c['target']).to(DEV)
    loss_fn = torch.nn.BCEWithLogitsLoss(reduction='sum')
    mixed.seed_arch_sampler(888)
    before = {k: p.detach().clone() for k, p in net.named_parameters()}
    # weight step
    MixedOp.MODE = None
    net.reset_binary_gates()
    red.begin_weight_step()
    net.unused_modules_off()
    loss = loss_fn(net(inp), tgt)
    loss.backward()
    red.finish_weight_step()
    net_optim.step()
    net.unused_modules_back()
    sampled = {id(p) for m in net.redundant_modules for p in m.candidate_ops[m.active_index[0]].parameters()}
    cand = {id(p) for m in net.redundant_modules for p in m.candidate_ops.parameters()}
    moved = 0
    for k, p in net.named_parameters():
        changed = bool(torch.any(p.detach() != before[k]))
        if id(p) in cand and id(p) not in sampled:
            assert not changed, k                      # unsampled candidates untouched (mixed.py:160-163)
        elif 'alpha' not in k and p.grad is not None and float(p.grad.abs().sum()) > 0:
            moved += int(changed)
    assert moved > 20 and np.isfinite(float(loss))
    # arch step
    MixedOp.MODE = 'full'
    try:
        net.reset_binary_gates()
        net.unused_modules_off()
        loss = loss_fn(net(inp), tgt)
        net.zero_grad()
        loss.backward()
        red.reduce_alpha_gate_grads()
        net.set_arch_param_grad()
        a0 = torch.stack([torch.nn.functional.pad(p.detach(), (0, 4 - p.numel())) for p in net.alpha_prob_parameters()]).clone()
        alpha_optim.step()
        net.unused_modules_back()
        a1 = torch.stack([torch.nn.functional.pad(p.detach(), (0, 4 - p.numel())) for p in net.alpha_prob_parameters()])
        assert float((a1 - a0).abs().max()) > 0
    finally:
        MixedOp.MODE = None
        red.fg.disable_sinks()


@pytest.mark.parametrize('mode', ['zero', 'skip'])
def test_flat_adam_checkpoint_interchanges_with_torch_adam(mode):
    """state_dict() is torch.optim.Adam's format (what the reference saves as 'net_optim', search_vqa.py:342-346): a
    torch Adam loads it and continues exactly as the FlatAdam does, and the other way round."""
    import io
    from mmnas_amd.optim import FlatAdam
    rs = np.random.RandomState(5)
    shapes = [(7, 5), (13,), (4, 3, 2), (64,)]
    init = [rs.standard_normal(sh).astype(np.float32) for sh in shapes]
    grads = [[rs.standard_normal(sh).astype(np.float32) for sh in shapes] for _ in range(4)]

    def mk():
        return [torch.nn.Parameter(torch.from_numpy(a.copy()).to(DEV)) for a in init]

    def give(ps, gs, skip_last):
        for j, (p, g) in enumerate(zip(ps, gs)):
            if skip_last and j == len(ps) - 1:
                p.grad = None if mode == 'skip' else torch.zeros_like(p)
            else:
                p.grad = torch.from_numpy(g).to(DEV)

    # two steps on the FlatAdam, checkpoint, then two more on (a) the same FlatAdam, (b) a torch Adam loaded from the
    # file, (c) a fresh FlatAdam loaded from what that torch Adam saves
    pa = mk()
    fa = FlatAdam(pa, lr=0.01, betas=(0.9, 0.98), eps=1e-9, absent_grads=mode)
    for t in range(2):
        fa.zero_grad()
        give(pa, grads[t], skip_last=(t == 1 and mode == 'skip'))
        fa.step()
    buf = io.BytesIO()
    torch.save(fa.state_dict(), buf)
    buf.seek(0)
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    ta = torch.optim.Adam(pb, lr=0.01, betas=(0.9, 0.98), eps=1e-9)
    ta.load_state_dict(torch.load(buf))
    pc = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    fc = FlatAdam(pc, lr=0.5, betas=(0.5, 0.5), eps=1.0, absent_grads=mode)     # (hyper-parameters come from the file)
    fc.load_state_dict(ta.state_dict())
    for t in (2, 3):
        fa.zero_grad(); give(pa, grads[t], False); fa.step()
        ta.zero_grad(); give(pb, grads[t], False); ta.step()
        fc.zero_grad(); give(pc, grads[t], False); fc.step()
    for a, b, c in zip(pa, pb, pc):
        assert rel_err(a.detach().cpu().numpy(), b.detach().cpu().numpy()) < 2e-6
        assert rel_err(c.detach().cpu().numpy(), b.detach().cpu().numpy()) < 2e-6



@pytest.mark.parametrize('mode', ['zero', 'skip'])
def test_conv_seq_follows_the_weights_across_a_flat_adam_step(mode):
    """the re-arranged StdConv weights were cached on the tensor's version counter, which a raw-pointer
    optimizer step (mmnas_adam_step on the flat buffer) never moves: after the first FlatAdam step the operator kept
    computing with its INITIAL weights.  conv_seq -> FlatAdam.step -> conv_seq must equal torch's conv1d on the parameter
    as it stands, forward and all gradients, both before and after the step."""
    import torch.nn.functional as F
    from mmnas_amd import ops
    from mmnas_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(3)
    B, S, d, k = 3, 100, 64, 7
    x = torch.randn(B, S, d, generator=g).to(DEV)
    dy = torch.randn(B, S, d, generator=g).to(DEV)
    w = torch.nn.Parameter((torch.randn(d, d, k, generator=g) * 0.1).to(DEV))
    b = torch.nn.Parameter(torch.randn(d, generator=g).to(DEV))
    opt = FlatAdam([w, b], lr=0.05, absent_grads=mode)

    def check():
        xr = x.double().requires_grad_(True)
        wr, br = w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
        yr = F.conv1d(xr.transpose(1, 2), wr, br, padding=k // 2).transpose(1, 2)
        yr.backward(dy.double())
        opt.zero_grad()
        if mode == 'skip':
            opt.fg.attach()
        xg = x.clone().requires_grad_(True)
        y = ops.conv_seq(xg, w, b)
        y.backward(dy)
        for got, ref in ((y, yr), (xg.grad, xr.grad), (w.grad, wr.grad), (b.grad, br.grad)):
            assert float((got.detach().double() - ref.detach()).abs().max()) <= 3e-6 * float(ref.abs().max())

    check()
    w0 = w.detach().clone()
    opt.step()
    torch.cuda.synchronize()
    assert float((w.detach() - w0).abs().max()) > 1e-3       # the step moved the weights (through the raw pointer) ...
    check()                                                  # ... and the operator sees them
    opt.step()
    check()

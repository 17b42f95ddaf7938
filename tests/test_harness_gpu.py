"""The step harness on the GPU against goldens made by the reference's own loop / losses:
  * capture (v) (tests/golden/traj.npz): two clip+Adam weight steps and one 'full' arch step with injected samples,
    replayed through SearchLoop = SupernetReducer + FlatAdam(absent_grads='zero') + WarmupOptimizer + ArchAdam;
  * the ITM triplet step with BCE_Loss and the VGD loss (tests/golden/losses.npz);
  * the kernels under them (gated sum of a MixedOp, fused alpha update, dense Adam) against torch."""
import numpy as np
import pytest
import torch

from tests.golden import cases
from tests.util import TOL, load, rel_err, REL_PATH_SELF_TOL, is_rel_path

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = torch.from_numpy


def _build(cls, c):
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = cls(c['cfg'], init)
    net.load_state_dict({k: T(v) for k, v in c['P'].items()})
    return net.to(DEV).train()


def _plan_list(plan):
    return plan['enc'] + plan['dec']


@pytest.mark.parametrize('full64', [False, True], ids=['small', 'B64_production_dimensions'])
def test_bilevel_trajectory_vs_reference_loop(full64):
    """full64 (round 6, tests/golden/traj64.npz): the same statements at BASELINE configs[2]'s own dimensions and batch -- HSIZE
    256, B = 64, 100 regions, 14 tokens, 3129 answers -- against the reference's own loop run on the CPU."""
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas_amd.harness import SearchLoop
    from tests.test_oracle_golden2 import check_trajectory
    fname = 'traj64.npz' if full64 else 'traj.npz'
    c, c2, plans = cases.traj_setup(full64=full64)
    H = cases.TRAJ_HYPER
    net = _build(Net_Search, c)
    loop = SearchLoop(net, net_lr=H['net_lr'], net_betas=H['net_betas'], net_eps=H['net_eps'], clip=H['clip'],
                      epoch_steps=H['epoch_steps'], warmup=True, alpha_lr=H['alpha_lr'], alpha_betas=H['alpha_betas'])
    try:
        inp = tuple(T(a).to(DEV) for a in c['inputs']); tgt = T(c['target']).to(DEV)
        inp2 = tuple(T(a).to(DEV) for a in c2['inputs']); tgt2 = T(c2['target']).to(DEV)
        net_keys = [k for k, _ in net.named_parameters() if 'alpha' not in k]
        named = dict(net.named_parameters())
        res = {'losses': [], 'gnorms': [], 'snap': {}, 'P0': {k: T(c['P'][k]) for k in net_keys}}

        def snap(tag):
            res['snap'][tag] = {k: named[k].detach().cpu().clone() for k in net_keys}

        for i in (0, 1):
            loss = loop.weight_step(inp, tgt, plan=_plan_list(plans[i]))
            res['losses'].append(float(loss.detach()))
            res['gnorms'].append(loop.net_optim.optimizer.grad_norm())
            snap('w%d' % (i + 1))
        loss = loop.arch_step(inp2, tgt2, plan=_plan_list(plans[2]))
        res['losses'].append(float(loss.detach()))
        gg, pg = net._flat_grads
        res['gate_grads'] = gg.cpu().numpy()
        res['prob_grads'] = pg.cpu().numpy()
        res['alpha_after'] = np.stack([np.pad(m.alpha_prob.detach().cpu().numpy(), (0, 4 - m.n_choices))
                                       for m in net.redundant_modules])
        snap('a')
        loss = loop.weight_step(inp, tgt, optimize=False, plan=_plan_list(plans[3]))
        res['losses'].append(float(loss.detach()))
        assert loop.net_optim._step == 2 and abs(loop.net_optim._rate - load(fname)['traj|lr'][1]) < 1e-12
        check_trajectory(res, fname=fname)
    finally:
        loop.reducer.fg.disable_sinks()


@pytest.mark.parametrize('full64', [False, True], ids=['small', 'B64_production_dimensions'])
def test_training_loop_trajectory_vs_reference_loop(full64):
    """tests/golden/train_traj.npz: the reference's own train_vqa.py statements (Net_Full + WarmupOptimizer + torch Adam +
    clip_grad_norm_), five steps with a decay before the last, replayed through harness.TrainLoop = GradReducer + FlatAdam.
    full64 (round 6, train_traj64.npz): the same at BASELINE configs[1]'s own dimensions and batch (HSIZE 512, B = 64)."""
    from mmnas.model.full_vqa import Net_Full
    from mmnas_amd.harness import TrainLoop
    from tests.test_oracle_golden2 import check_train_trajectory
    c, c2 = cases.train_traj_setup(full64=full64)
    H = cases.TRAIN_HYPER
    net = _build(Net_Full, c)
    loop = TrainLoop(net, lr=H['lr'], betas=H['betas'], eps=H['eps'], clip=H['clip'], epoch_steps=H['epoch_steps'], warmup=True)
    try:
        batches = [(tuple(T(a).to(DEV) for a in c['inputs']), T(c['target']).to(DEV)),
                   (tuple(T(a).to(DEV) for a in c2['inputs']), T(c2['target']).to(DEV))]
        named = dict(net.named_parameters())
        res = {'losses': [], 'gnorms': [], 'rates': [], 'snap': {}, 'P0': {k: T(v) for k, v in c['P'].items()}}
        for i in range(5):
            if i == 4:
                loop.decay(H['decay_r'])
            loss = loop.step(*batches[i % 2])
            res['losses'].append(float(loss.detach()))
            res['gnorms'].append(loop.net_optim.optimizer.grad_norm())
            res['rates'].append(loop.net_optim._rate)
            if i in (0, 3, 4):
                res['snap']['s%d' % (i + 1)] = {k: named[k].detach().cpu().clone() for k in named}
        check_train_trajectory(res, fname='train_traj64.npz' if full64 else 'train_traj.npz', stray=2e-4 if full64 else 0.0)   # (measured: 16 of 262144 coordinates of the one large tensor at step 1, 3 at steps 4 / 5)
    finally:
        loop.reducer.fg.disable_sinks()


def test_default_products_follow_the_fp32_mfma_through_a_bilevel_run(monkeypatch):
    """Twelve optimizer steps of the bilevel loop (10 weight steps with clip + Adam, 2 'full' arch steps, dropout on with
    fixed seeds) with the default 6-product arithmetic and with the fp32 MFMA, from the same initial state: the loss
    sequences stay together (the products agree to fp32 rounding; what differs is summation order, which Adam's
    normalisation amplifies slowly)."""
    import mmnas_amd._lib as L
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas_amd import ops
    from mmnas_amd.harness import SearchLoop
    c = cases.net_case('vqa', None, 1234, search=True, HSIZE=256, B=8, Sx=14, Sy=20)
    inp = tuple(T(a).to(DEV) for a in c['inputs']); tgt = T(c['target']).to(DEV)
    plans = [cases.search_plan(np.random.RandomState(100 + i), 'full' if i % 6 == 5 else None) for i in range(12)]
    runs = {}
    for mode in (6, 0):
        monkeypatch.setenv('MMNAS_GEMM_SPLIT', str(mode))
        L.lib().mmnas_gemm_reload_tuning()
        c['cfg'].DROPOUT_R = 0.1
        net = _build(Net_Search, c)
        loop = SearchLoop(net, net_lr=1e-4, alpha_lr=0.05)
        ops.manual_seed(4321)
        losses = []
        try:
            for i, pl in enumerate(plans):
                step = loop.arch_step if i % 6 == 5 else loop.weight_step
                losses.append(float(step(inp, tgt, plan=_plan_list(pl)).detach()))
            torch.cuda.synchronize()
            runs[mode] = (np.array(losses), {k: p.detach().cpu().numpy().copy() for k, p in net.named_parameters()})
        finally:
            loop.reducer.fg.disable_sinks()
    monkeypatch.delenv('MMNAS_GEMM_SPLIT')
    L.lib().mmnas_gemm_reload_tuning()
    la, lb = runs[6][0], runs[0][0]
    assert np.all(np.isfinite(la)) and np.abs(la - lb).max() <= 2e-4 * np.abs(lb).max(), (la, lb)
    # parameters: L2 over everything (a maximum would pick the parameters whose gradient is mathematically zero --
    # the softmax-shift-invariant logit biases -- which Adam moves by +-lr per step in the direction of rounding noise)
    keys = [k for k in c['P'] if 'alpha' not in k]
    moved = np.sqrt(sum(float(((runs[0][1][k] - c['P'][k]).astype(np.float64) ** 2).sum()) for k in keys))
    apart = np.sqrt(sum(float(((runs[6][1][k] - runs[0][1][k]).astype(np.float64) ** 2).sum()) for k in keys))
    assert moved > 0 and apart <= 0.05 * moved, (moved, apart)     # (measured 1.8 %: two runs of ONE arithmetic with
                                                                   #  different summation orders part as much)


def test_arch_step_fused_path_matches_per_module_path():
    """SearchLoop.arch_step (gated-sum kernel + gate-gradient block + fused alpha Adam) against the reference-shaped
    per-module sequence on the same net: MixedOp.set_arch_param_grad + torch.optim.Adam."""
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas.model.mixed import MixedOp
    from mmnas_amd.harness import SearchLoop
    c = cases.net_case('vqa', None, 4141, search=True)
    plan = _plan_list(cases.search_plan(np.random.RandomState(5), 'full'))
    inp = tuple(T(a).to(DEV) for a in c['inputs']); tgt = T(c['target']).to(DEV)
    loss_fn = torch.nn.BCEWithLogitsLoss(reduction='sum')
    # (a) per-module path, plain autograd gate gradients
    net = _build(Net_Search, c)
    opt = torch.optim.Adam(list(net.alpha_prob_parameters()), 0.1, betas=(0.0, 0.999))
    MixedOp.MODE = 'full'
    try:
        net.set_sampled(plan)
        loss_a = loss_fn(net(inp), tgt)
        net.zero_grad()
        loss_a.backward()
        gate_a = np.stack([np.pad(m.alpha_gate.grad.cpu().numpy(), (0, 4 - m.n_choices)) for m in net.redundant_modules])
        net.set_arch_param_grad()
        opt.step()
        alpha_a = np.stack([np.pad(m.alpha_prob.detach().cpu().numpy(), (0, 4 - m.n_choices)) for m in net.redundant_modules])
    finally:
        MixedOp.MODE = None
    # (b) fused path
    net2 = _build(Net_Search, c)
    loop = SearchLoop(net2)
    try:
        loss_b = loop.arch_step(inp, tgt, plan=plan)
        gg, _ = net2._flat_grads
        alpha_b = np.stack([np.pad(m.alpha_prob.detach().cpu().numpy(), (0, 4 - m.n_choices)) for m in net2.redundant_modules])
        assert abs(float(loss_a) - float(loss_b)) < 1e-5 * abs(float(loss_a))
        assert rel_err(gg.cpu().numpy(), gate_a) < 1e-4
        assert rel_err(alpha_b, alpha_a) < 1e-4
        # the sampling cache sees the update
        p1 = net2._probs_cpu(net2._flat_alphas()[0]).numpy()
        assert rel_err(p1[12:], torch.softmax(T(alpha_b[12:]), 1).numpy()) < 1e-5
    finally:
        loop.reducer.fg.disable_sinks()


def test_mixed_sum_kernels_vs_torch():
    from mmnas_amd import ops
    g = torch.Generator().manual_seed(3)
    for n, shape, active in ((4, (64, 100, 256), 2), (2, (3, 14, 64), 0), (4, (5, 7, 36), 3)):
        outs = [torch.randn(shape, generator=g).to(DEV) for _ in range(n)]
        gate = torch.randn(n, generator=g).to(DEV).requires_grad_(True)
        a = outs[active].clone().requires_grad_(True)
        lst = list(outs); lst[active] = a
        if n == 4:
            lst[(active + 1) % n] = None                  # a candidate that takes no part (mode 'two')
        y = ops.mixed_sum(gate, lst, active)
        dout = torch.randn(shape, generator=g).to(DEV)
        y.backward(dout)
        gate64 = gate.detach().double()
        want = sum(gate64[j] * (lst[j].detach().double()) for j in range(n) if lst[j] is not None)
        assert rel_err(y.detach().cpu().numpy(), want.cpu().numpy()) < 1e-6
        wg = torch.stack([(dout.double() * lst[j].detach().double()).sum() if lst[j] is not None else torch.zeros((), dtype=torch.float64, device=DEV)
                          for j in range(n)])
        assert rel_err(gate.grad.cpu().numpy(), wg.cpu().numpy()) < 1e-5
        assert rel_err(a.grad.cpu().numpy(), (gate64[active] * dout.double()).cpu().numpy()) < 1e-6


def test_alpha_full_step_kernel_vs_torch_adam():
    from mmnas_amd import ops
    g = torch.Generator().manual_seed(9)
    rows, width = 30, 4
    a = torch.randn(rows, width, generator=g)
    a[:12, 2:] = float('-inf')                          # encoder nodes have two candidates
    prob = a.clone().to(DEV)
    m = torch.zeros_like(prob); v = torch.zeros_like(prob); pg = torch.zeros_like(prob)
    ps = [torch.nn.Parameter(a[i, :(2 if i < 12 else 4)].clone().double()) for i in range(rows)]
    opt = torch.optim.Adam(ps, 0.1, betas=(0.0, 0.999))
    for step in (1, 2, 3):
        gg = torch.randn(rows, width, generator=g)
        gg[:12, 2:] = 0
        for i, p in enumerate(ps):
            gi = gg[i, :p.numel()].double()
            pr = torch.softmax(p.detach(), 0)
            p.grad = pr * (gi - (gi * pr).sum())
        opt.step()
        ops.alpha_full_step(prob, gg.to(DEV), m, v, pg, 0.1, (0.0, 0.999), 1e-8, step)
        for i, p in enumerate(ps):
            n = p.numel()
            assert rel_err(pg[i, :n].cpu().numpy(), p.grad.numpy()) < 1e-5
            assert rel_err(prob[i, :n].cpu().numpy(), p.detach().numpy()) < 1e-5
            assert bool(torch.all(torch.isinf(prob[i, n:]))), 'padding columns must stay -inf'


def test_alpha_optimizer_checkpoint_is_torch_adams():
    """ArchAdam.state_dict() loads into torch.optim.Adam(net.alpha_prob_parameters()) -- what search_vqa.py:350 saves --
    and back; both continue with the same update."""
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas_amd.harness import SearchLoop
    c = cases.net_case('vqa', None, 77, search=True, HSIZE=64)
    inp = tuple(T(a).to(DEV) for a in c['inputs']); tgt = T(c['target']).to(DEV)
    net = _build(Net_Search, c)
    loop = SearchLoop(net)
    try:
        plans = [cases.search_plan(np.random.RandomState(40 + i), 'full') for i in range(3)]
        for pl in plans[:2]:
            loop.arch_step(inp, tgt, plan=_plan_list(pl))
        sd = loop.alpha_optim.state_dict()
        alphas = [p.detach().clone().requires_grad_(True) for p in net.alpha_prob_parameters()]
        ta = torch.optim.Adam(alphas, lr=0.1, betas=(0.0, 0.999))
        ta.load_state_dict(sd)
        assert ta.state_dict()['state'][0]['step'] == 2
        # third step on both: the torch side takes the gradients the fused step exposes as alpha_prob.grad
        loop.arch_step(inp, tgt, plan=_plan_list(plans[2]))
        for a, p in zip(alphas, net.alpha_prob_parameters()):
            a.grad = p.grad.detach().clone()
        ta.step()
        for a, p in zip(alphas, net.alpha_prob_parameters()):
            assert rel_err(p.detach().cpu().numpy(), a.detach().cpu().numpy()) < 1e-5
        # and back: a fresh ArchAdam resumes from the torch optimizer's file
        from mmnas_amd.harness import ArchAdam
        fresh = ArchAdam(net, lr=0.5, betas=(0.5, 0.5))
        fresh.load_state_dict(ta.state_dict())
        assert fresh.steps == 3 and fresh.lr == 0.1 and fresh.betas == (0.0, 0.999)
        assert rel_err(fresh.m.cpu().numpy(), loop.alpha_optim.m.cpu().numpy()) < 1e-5
        assert rel_err(fresh.v.cpu().numpy(), loop.alpha_optim.v.cpu().numpy()) < 1e-5
    finally:
        loop.reducer.fg.disable_sinks()


def test_flat_adam_dense_mode_is_torch_adam_with_zero_gradients():
    """absent_grads='zero': what the reference loop amounts to (search_vqa.py:285-300) -- parameters without a gradient
    are stepped with a zero gradient, one global step count; 'skip' mode freezes them."""
    from mmnas_amd.optim import FlatAdam
    g = torch.Generator().manual_seed(1)
    shapes = [(33, 17), (5,), (64, 64), (7, 3)]
    init = [torch.randn(s, generator=g) for s in shapes]
    for mode in ('zero', 'skip'):
        ps = [torch.nn.Parameter(t.clone().to(DEV)) for t in init]
        ref = [torch.nn.Parameter(t.clone().double()) for t in init]
        opt = FlatAdam(ps, lr=1e-2, betas=(0.9, 0.98), eps=1e-9, absent_grads=mode)
        ropt = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.98), eps=1e-9)
        for step in range(4):
            live = [0, 1, 2, 3] if step == 0 else ([0, 2] if step % 2 else [1, 2, 3])
            opt.zero_grad()
            grads = {i: torch.randn(shapes[i], generator=g) for i in live}
            for i in range(4):
                if i in grads:
                    if ps[i].grad is None:
                        ps[i].grad = grads[i].to(DEV)                 # a stray gradient outside the flat buffer
                    else:
                        ps[i].grad.copy_(grads[i].to(DEV))
                    ref[i].grad = grads[i].double()
                else:
                    ps[i].grad = None
                    ref[i].grad = torch.zeros_like(ref[i]) if mode == 'zero' else None
            tot = torch.sqrt(sum((gr.double() ** 2).sum() for gr in grads.values()))
            coef = min(1.0, 0.5 / (float(tot) + 1e-6))
            for i in live:
                ref[i].grad.mul_(coef)
            opt.step(max_norm=0.5)
            ropt.step()
            assert abs(opt.grad_norm() - float(tot)) < 1e-4 * float(tot)
            for i in range(4):
                assert rel_err(ps[i].detach().cpu().numpy(), ref[i].detach().numpy()) < 2e-5, (mode, step, i)


@pytest.mark.parametrize('products', [6, 0, 3])
def test_itm_triplet_step_vs_reference(products, monkeypatch):
    """BASELINE configs[4] (train_itm.py:380-391) against the reference's own step.  products: the default 6 bf16-MFMA
    products per fp32 product, the fp32 MFMA, and the 3-product form (16 mantissa bits of every operand kept, fp32
    accumulation), all at the same 1e-3 tolerance.  (The reference itself is fp32 throughout -- no `half` / `amp` anywhere in
    it; "fp16 MFMA" is BASELINE.json's wording for configs[4].  The 3-product form is this library's experiment for
    that label -- finer than fp16's 11 bits -- not a restatement of something the reference does; there is no fp16-operand
    path.)"""
    import mmnas_amd._lib as L
    from mmnas.model.full_itm import Net_Full
    from mmnas.utils.itm_loss import BCE_Loss
    from mmnas_amd.harness import itm_triplet_step
    monkeypatch.setenv('MMNAS_GEMM_SPLIT', str(products))
    L.lib().mmnas_gemm_reload_tuning()
    try:
        _itm_triplet_check(Net_Full, BCE_Loss, itm_triplet_step)
    finally:
        monkeypatch.delenv('MMNAS_GEMM_SPLIT')
        L.lib().mmnas_gemm_reload_tuning()


def test_itm_triplet_step_single_pass_bf16_products(monkeypatch):
    """BASELINE configs[4] as written ("fp16 MFMA"): MMNAS_GEMM_SPLIT=1, every projection / FFN product as ONE bf16 MFMA pass
    on bf16-rounded operands with fp32 accumulation (attention cores, LayerNorm, softmax, loss stay fp32).  Reduced
    precision by construction (8 mantissa bits per operand; the reference's fp16 would keep 11): NOT held to the 1e-3 parity
    bar -- the stated tolerance against the reference's fp32 step is 3e-2 on loss, scores and gradient norms -- and never
    the default or a headline."""
    import mmnas_amd._lib as L
    from mmnas.model.full_itm import Net_Full
    from mmnas.utils.itm_loss import BCE_Loss
    from mmnas_amd.harness import itm_triplet_step
    monkeypatch.setenv('MMNAS_GEMM_SPLIT', '1')
    L.lib().mmnas_gemm_reload_tuning()
    try:
        _itm_triplet_check(Net_Full, BCE_Loss, itm_triplet_step, tol=3e-2, gtol=3e-2)
        with pytest.raises(AssertionError):          # ... and it really is the reduced-precision path
            _itm_triplet_check(Net_Full, BCE_Loss, itm_triplet_step, tol=1e-5, gtol=1e-5)
    finally:
        monkeypatch.delenv('MMNAS_GEMM_SPLIT')
        L.lib().mmnas_gemm_reload_tuning()


def _itm_triplet_check(Net_Full, BCE_Loss, itm_triplet_step, tol=TOL, gtol=2e-3, full64=False):
    npz = load('losses64.npz' if full64 else 'losses.npz')
    c, neg, _ = cases.losses_cases(full64)
    net = _build(Net_Full, c)
    pos = tuple(T(a).to(DEV) for a in c['inputs']); ng = tuple(T(a).to(DEV) for a in neg['inputs'])
    loss = itm_triplet_step(net, BCE_Loss(), pos, ng)
    assert abs(float(loss) - float(npz['itm|loss'])) < tol * float(npz['itm|loss'])
    sp = net(pos)
    assert rel_err(sp.detach().cpu().numpy(), npz['itm|scores'][0]) < tol
    keys = [str(k) for k in npz['itm|gradnorm_keys']]
    named = dict(net.named_parameters())
    top = float(np.max(npz['itm|gradnorms']))
    for k, n in zip(keys, npz['itm|gradnorms']):
        mine = 0.0 if named[k].grad is None else float(named[k].grad.double().norm())
        assert abs(mine - n) <= gtol * n + 1e-5 * top * (gtol / 2e-3), (k, mine, n)
    assert rel_err(named['proj.weight'].grad.cpu().numpy(), npz['itm|g:proj.weight']) < tol


def test_itm_triplet_step_at_the_full_batch_vs_reference():
    """BASELINE configs[4] at its own dimensions and batch: three forwards of B = 160 (50 tokens, 36 regions, HSIZE 512), BCE_Loss,
    one backward -- against the reference's own step on the CPU (tests/golden/losses64.npz, make_golden.gen_losses64; round 6)."""
    from mmnas.model.full_itm import Net_Full
    from mmnas.utils.itm_loss import BCE_Loss
    from mmnas_amd.harness import itm_triplet_step
    _itm_triplet_check(Net_Full, BCE_Loss, itm_triplet_step, full64=True)


@pytest.mark.parametrize('full64', [False, True], ids=['small', 'B64_production_dimensions'])
def test_vgd_loss_vs_reference(full64):
    from mmnas.model.full_vgd import Net_Full
    from mmnas_amd.harness import vgd_loss
    npz = load('losses64.npz' if full64 else 'losses.npz')
    c = cases.losses_cases(full64)[2]
    t = {k: T(v).to(DEV) for k, v in cases.vgd_targets(c, 9204).items()}
    net = _build(Net_Full, c)
    ps, pr = net(tuple(T(a).to(DEV) for a in c['inputs']))
    loss = vgd_loss(ps, pr, t['scores'], t['scores_mask'], t['bbox'], t['bbox_mask'])
    loss.backward()
    assert rel_err(ps.detach().cpu().numpy(), npz['vgd|pred_scores']) < TOL
    assert rel_err(pr.detach().cpu().numpy(), npz['vgd|pred_reg']) < TOL
    assert abs(float(loss) - float(npz['vgd|loss_parts'][2])) < TOL * abs(float(npz['vgd|loss_parts'][2]))
    keys = [str(k) for k in npz['vgd|gradnorm_keys']]
    named = dict(net.named_parameters())
    top = float(np.max(npz['vgd|gradnorms']))
    for k, n in zip(keys, npz['vgd|gradnorms']):
        mine = 0.0 if named[k].grad is None else float(named[k].grad.double().norm())
        assert abs(mine - n) <= 2e-3 * n + 1e-5 * top, (k, mine, n)


def test_operator_backward_twice_raises_a_clear_error():
    from mmnas.utils.ops_adapter import OpsAdapter
    cfg = cases.small_cfg(HSIZE=128)
    op = OpsAdapter().OPS['feed_forward'](cfg, norm=True, residual=True).to(DEV)
    x = torch.randn(2, 5, 128, device=DEV, requires_grad=True)
    y = op(x, None, None, None, None).sum()
    y.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match='second time'):
        y.backward()


def test_node_mix_kernels_vs_torch():
    """mmnas_node_mix_fwd/bwd (one pass per supernet node of the architecture step): out = sum_j gate_j LN_j(z_j) with
    the reference's LayerNorm (unbiased std, eps on the std; modules.py:52-56), candidates without LayerNorm passed
    through, non-binary gates; backward: every gate's <dout, LN_j(z_j)> added onto the gate-gradient row and
    d_active = gate[active] * dout."""
    import ctypes as C
    from mmnas_amd import _lib as L
    g = torch.Generator().manual_seed(7)
    for n, M, d, active, plain in ((4, 6400, 256, 2, ()), (2, 896, 256, 0, (1,)), (4, 37, 512, 3, (0, 2)), (3, 5, 36, 1, ()), (1, 9, 1024, 0, ())):
        z = [torch.randn(M, d, generator=g).mul_(1.5).add_(0.3).to(DEV) for _ in range(n)]
        la = [None if j in plain else (torch.rand(d, generator=g) + 0.5).to(DEV) for j in range(n)]
        lb = [None if j in plain else torch.randn(d, generator=g).to(DEV) for j in range(n)]
        gate = torch.randn(n, generator=g).to(DEV)
        dout = torch.randn(M, d, generator=g).to(DEV)
        eps = 1e-6

        def ln(x, a, b):
            x = x.double()
            mu = x.mean(-1, keepdim=True)
            sd = x.std(-1, keepdim=True)          # unbiased
            return a.double() * (x - mu) / (sd + eps) + b.double()
        outs = [zj.double() if a is None else ln(zj, a, b) for zj, a, b in zip(z, la, lb)]
        want = sum(gate[j].double() * outs[j] for j in range(n))
        arr = lambda ts: (C.c_void_p * n)(*[L.fptr(t) for t in ts])
        out = torch.empty(M, d, device=DEV)
        L.check(L.lib().mmnas_node_mix_fwd(arr(z), arr(la), arr(lb), n, L.fptr(gate), L.fptr(out), M, d, eps, L.stream()))
        assert rel_err(out.cpu().numpy(), want.cpu().numpy()) < 2e-6, (n, M, d)
        dgate = torch.full((n,), 0.25, device=DEV)
        dact = torch.empty(M, d, device=DEV)
        ws = torch.empty(L.lib().mmnas_mixed_sum_ws_floats(), device=DEV)
        L.check(L.lib().mmnas_node_mix_bwd(arr(z), arr(la), arr(lb), n, L.fptr(gate), L.fptr(dout), L.fptr(dact), active, L.fptr(dgate),
                                           L.fptr(ws), M, d, eps, L.stream()))
        wg = torch.stack([(dout.double() * o).sum() for o in outs]) + 0.25
        assert rel_err(dgate.cpu().numpy(), wg.cpu().numpy()) < 1e-5, (n, M, d)
        assert rel_err(dact.cpu().numpy(), (gate[active] * dout).cpu().numpy()) < 1e-6


@pytest.mark.parametrize('mode', ['full', 'two'])
def test_arch_step_through_the_mixed_chain_equals_the_per_candidate_path(mode, monkeypatch):
    """The architecture step as ONE native call per direction (every evaluated candidate, node epilogues fused) against
    the per-candidate path (one autograd node per candidate + the gated-sum kernels, MMNAS_MIXED_CHAIN=0): loss, the gate
    gradients, the alpha update and every weight gradient of the sampled candidates, stem and head."""
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas_amd.harness import SearchLoop
    c = cases.net_case('vqa', None, 4242, search=True, B=3)
    plan = _plan_list(cases.search_plan(np.random.RandomState(6), mode))
    inp = tuple(T(a).to(DEV) for a in c['inputs']); tgt = T(c['target']).to(DEV)
    res = {}
    for chain in ('0', '1'):
        monkeypatch.setenv('MMNAS_MIXED_CHAIN', chain)
        net = _build(Net_Search, c)
        loop = SearchLoop(net, arch_mode=mode)
        try:
            loss = loop.arch_step(inp, tgt, plan=plan)
            torch.cuda.synchronize()
            gg, _ = net._flat_grads
            res[chain] = dict(loss=float(loss), gg=gg.cpu().numpy().copy(),
                              alpha=np.stack([np.pad(m.alpha_prob.detach().cpu().numpy(), (0, 4 - m.n_choices)) for m in net.redundant_modules]),
                              grads={k: p.grad.detach().cpu().numpy().copy() for k, p in net.named_net_parameters() if p.grad is not None})
        finally:
            loop.reducer.fg.disable_sinks()
    a, b = res['0'], res['1']
    assert abs(a['loss'] - b['loss']) < 1e-5 * abs(a['loss'])
    assert float(np.abs(a['gg']).max()) > 0 and rel_err(b['gg'], a['gg']) < 1e-4
    assert rel_err(b['alpha'], a['alpha']) < 1e-5
    top = max(float(np.abs(v).max()) for v in a['grads'].values())
    for k, v in a['grads'].items():
        assert float(np.abs(b['grads'][k] - v).max()) <= (REL_PATH_SELF_TOL if is_rel_path(k) else 1e-4) * max(float(np.abs(v).max()), 1e-3 * top), k
    nz = sum(float(np.abs(v).max()) > 0 for v in a['grads'].values())
    assert nz > 60          # the sampled candidates, stem and head carry gradients on both paths


def test_search_loop_mode_two_follows_the_per_module_statements():
    """SearchLoop(arch_mode='two') (search_vqa.py:317-334 with ALPHA_BINARY_MODE 'two'): the sampled pair's gate gradients,
    set_arch_param_grad over the pair, torch Adam and rescale_updated_arch_param -- against the same statements written out
    on a second net through the per-module path."""
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas.model.mixed import MixedOp
    from mmnas_amd.harness import SearchLoop
    c = cases.net_case('vqa', None, 4343, search=True)
    plan = _plan_list(cases.search_plan(np.random.RandomState(8), 'two'))
    inp = tuple(T(a).to(DEV) for a in c['inputs']); tgt = T(c['target']).to(DEV)
    loss_fn = torch.nn.BCEWithLogitsLoss(reduction='sum')
    net = _build(Net_Search, c)
    opt = torch.optim.Adam(list(net.alpha_prob_parameters()), 0.1, betas=(0.0, 0.999))
    MixedOp.MODE = 'two'
    try:
        net.set_sampled(plan)
        loss_a = loss_fn(net(inp), tgt)
        net.zero_grad()
        loss_a.backward()
        net.set_arch_param_grad()
        pg_a = np.stack([np.pad(m.alpha_prob.grad.cpu().numpy(), (0, 4 - m.n_choices)) for m in net.redundant_modules])
        opt.step()
        net.rescale_updated_arch_param()
        alpha_a = np.stack([np.pad(m.alpha_prob.detach().cpu().numpy(), (0, 4 - m.n_choices)) for m in net.redundant_modules])
    finally:
        MixedOp.MODE = None
    net2 = _build(Net_Search, c)
    loop = SearchLoop(net2, arch_mode='two')
    try:
        loss_b = loop.arch_step(inp, tgt, plan=plan)
        pg_b = np.stack([np.pad(m.alpha_prob.grad.cpu().numpy(), (0, 4 - m.n_choices)) for m in net2.redundant_modules])
        alpha_b = np.stack([np.pad(m.alpha_prob.detach().cpu().numpy(), (0, 4 - m.n_choices)) for m in net2.redundant_modules])
        assert abs(float(loss_a) - float(loss_b)) < 1e-5 * abs(float(loss_a))
        assert float(np.abs(pg_a).max()) > 0 and rel_err(pg_b, pg_a) < 1e-4
        assert rel_err(alpha_b, alpha_a) < 1e-5
        assert MixedOp.MODE is None
    finally:
        loop.reducer.fg.disable_sinks()


@pytest.mark.parametrize('full64', [False, True], ids=['small', 'B64_production_dimensions'])
@pytest.mark.parametrize('which', ['bilevel', 'train'])
def test_reference_loop_trajectories_with_the_ragged_decoder_stream(which, full64):
    """The reference's OWN loops (traj.npz: search_vqa.py:279-337; train_traj.npz: train_vqa.py:291-311) replayed with the ragged decoder stream on (ops.set_unpad): losses, gradient norms, per-tensor parameter
    motion and post-step alphas of the padded reference computation are met on the valid rows alone -- and the chain really
    ran packed."""
    from mmnas_amd import ops
    seen = []
    orig = ops.BackboneFn.apply
    ops.BackboneFn.apply = lambda *a: (seen.append(a[10] is not None), orig(*a))[1]
    prev = ops.set_unpad(True)
    try:
        if which == 'bilevel':
            test_bilevel_trajectory_vs_reference_loop(full64)
        else:
            test_training_loop_trajectory_vs_reference_loop(full64)
    finally:
        ops.set_unpad(prev)
        ops.BackboneFn.apply = orig
    assert seen and all(seen), seen

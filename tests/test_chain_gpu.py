"""The backbone chain (mmnas_chain_fwd/bwd behind ops.BackboneFn: every cell operator in one native call per direction,
weight gradients on a side stream) against the per-operator path on the same network, weights, batch and dropout
seeds: identical operators and kernels in a different issue order, so outputs and data gradients agree to round-off
and parameter gradients to the float-atomic summation order of the split-K weight-gradient products."""
import numpy as np
import pytest
import torch

from tests.golden import cases
from tests.util import REL_PATH_SELF_TOL, is_rel_path, rel_err

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = torch.from_numpy


def _run(task, arch, search, chain, side, monkeypatch, dropout=0.1, plan=None, B=3, glimpse1=True):
    import importlib
    from mmnas_amd import dp, ops
    from mmnas.model.mixed import MixedOp
    monkeypatch.setenv('MMNAS_CHAIN', '1' if chain else '0')
    monkeypatch.setenv('MMNAS_SIDE_STREAM', '1' if side else '0')
    # (0: AttFlat's glimpse-logit layer as GEMM launches instead of the matrix-vector kernels -- in both paths)
    monkeypatch.setenv('MMNAS_HEAD_GLIMPSE1', '1' if glimpse1 else '0')
    c = cases.net_case(task, arch, 31337, search=search, B=B, Sx=6, Sy=9)
    c['cfg'].DROPOUT_R = dropout
    mod = importlib.import_module('mmnas.model.%s_%s' % ('hygr' if search else 'full', task))
    cls = mod.Net_Search if search else mod.Net_Full
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = cls(c['cfg'], init)
    net.load_state_dict({k: T(v) for k, v in c['P'].items()})
    net = net.to(DEV).train()
    calls = []
    orig = ops.BackboneFn.apply
    monkeypatch.setattr(ops, 'backbone_chain', lambda *a: (calls.append(1), orig(*a))[1])
    inp = tuple(T(a).to(DEV) for a in c['inputs'])
    tgt = T(c['target']).to(DEV)
    ops.manual_seed(99)
    if search:
        MixedOp.MODE = None
        net.set_sampled(plan)
        red = dp.SupernetReducer(net)
        red.begin_weight_step()
    else:
        red = dp.GradReducer(list(net.parameters()))
        red.begin_step()
    try:
        pred = net(inp)
        if task == 'vgd':
            loss = (pred[0] * tgt).sum() + 0.5 * (pred[1] ** 2).sum()
            out = torch.cat([pred[0].reshape(-1), pred[1].reshape(-1)])
        elif task == 'itm':
            loss = torch.nn.functional.binary_cross_entropy(pred, tgt, reduction='sum')
            out = pred
        else:
            loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, tgt, reduction='sum')
            out = pred
        loss.backward()
        if search:
            red.finish_weight_step()
        else:
            red.finish()
        torch.cuda.synchronize()
        grads = {k: (p.grad.detach().cpu().numpy().copy() if p.grad is not None else None) for k, p in net.named_parameters()}
        return out.detach().cpu().numpy(), grads, len(calls)
    finally:
        red.fg.disable_sinks()


def _compare(a, b):
    out_a, g_a, _ = a
    out_b, g_b, _ = b
    assert rel_err(out_a, out_b) < 1e-6
    top = max(float(np.abs(g).max()) for g in g_b.values() if g is not None)
    for k in g_b:
        if g_b[k] is None:
            assert g_a[k] is None or not np.any(g_a[k]), k
            continue
        assert g_a[k] is not None, k
        diff = float(np.abs(g_a[k] - g_b[k]).max())
        # (relation path: the chain computes all relation operators' bias on the fp32 MFMA in one launch per direction
        #  (relmulti.hip), the per-operator path one vector-pipe / MFMA launch each (relfused.hip) -- two summation orders
        #  over 1/r-weighted terms of random sign.  Both are held to the reference's float64 value elsewhere
        #  (tests/util.py::check_grad_samples); against each other they get the bound two fp32 orders can meet)
        tol = REL_PATH_SELF_TOL if is_rel_path(k) else 2e-5
        assert diff <= tol * max(float(np.abs(g_b[k]).max()), 1e-3 * top), (k, diff)


@pytest.mark.parametrize('task,arch', [('vqa', 'mmnas_vqa'), ('vqa', 'mcan'), ('vgd', 'mmnas_vgd'), ('itm', 'mmnas_itm')])
def test_chain_equals_per_operator_path_net_full(task, arch, monkeypatch):
    ref = _run(task, arch, False, False, False, monkeypatch)
    assert ref[2] == 0
    for side in (False, True):
        got = _run(task, arch, False, True, side, monkeypatch)
        assert got[2] == 1, 'the backbone chain was not taken'
        _compare(got, ref)


@pytest.mark.parametrize('task,arch,search', [('vqa', 'mmnas_vqa', False), ('itm', 'mmnas_itm', False), ('vqa', None, True)])
def test_head_one_glimpse_kernels_equal_the_gemm_form(task, arch, search, monkeypatch):
    """ATTFLAT_GLIMPSES = 1: the native head computes the glimpse logits as a matrix-vector product and their backward as an
    outer product + column sums reduced by the next pair launch (head.hip glimpse1_*), instead of GEMM launches with one
    live column / one K step.  Same logits; every parameter gradient to round-off (the relation projections amplify the
    head's last bits the most: 6e-5 of their largest entry seen)."""
    plan = None
    if search:
        pl = cases.search_plan(np.random.RandomState(1), None)
        plan = pl['enc'] + pl['dec']
    ref = _run(task, arch, search, True, False, monkeypatch, plan=plan, glimpse1=False)
    got = _run(task, arch, search, True, False, monkeypatch, plan=plan, glimpse1=True)
    assert ref[2] == 1 and got[2] == 1
    out_a, g_a, _ = got
    out_b, g_b, _ = ref
    assert rel_err(out_a, out_b) < 2e-6
    top = max(float(np.abs(g).max()) for g in g_b.values() if g is not None)
    for k in g_b:
        if g_b[k] is None:
            assert g_a[k] is None or not np.any(g_a[k]), k
            continue
        diff = float(np.abs(g_a[k] - g_b[k]).max())
        assert diff <= 3e-4 * max(float(np.abs(g_b[k]).max()), 1e-3 * top), (k, diff)


@pytest.mark.parametrize('task', ['vqa', 'vgd', 'itm'])
def test_chain_equals_per_operator_path_supernet_weight_step(task, monkeypatch):
    for seed in (1, 2):
        plan = cases.search_plan(np.random.RandomState(seed), None)
        flat = plan['enc'] + plan['dec']
        ref = _run(task, None, True, False, False, monkeypatch, plan=flat)
        got = _run(task, None, True, True, True, monkeypatch, plan=flat)
        assert ref[2] == 0 and got[2] == 1
        _compare(got, ref)


@pytest.mark.parametrize('task,search', [('vqa', True), ('vqa', False), ('vgd', True)])
def test_chain_encoder_decoder_overlap_gives_the_same_result(task, search, monkeypatch):
    """mmnas_set_chain_overlap(1): the language stream's operators on their own stream beside the image stream's leading
    operators (fork / join by events) -- the same kernels on the same data."""
    from mmnas_amd import _lib as L
    lib = L.lib()
    for seed in (3, 4, 5):      # sampled architectures with 0, 1, ... image operators in front of the first guided one
        plan = None
        if search:
            pl = cases.search_plan(np.random.RandomState(seed), None)
            plan = pl['enc'] + pl['dec']
        arch = None if search else 'mmnas_' + task
        ref = _run(task, arch, search, True, False, monkeypatch, plan=plan)
        prev = lib.mmnas_set_chain_overlap(1)
        try:
            got = _run(task, arch, search, True, False, monkeypatch, plan=plan)
        finally:
            lib.mmnas_set_chain_overlap(prev)
        assert ref[2] == 1 and got[2] == 1
        _compare(got, ref)
        if not search:
            break


def _plain_autograd(task, arch, search, monkeypatch, autograd_chain, plan=None, dropout=0.1, mode=None):
    """One forward + backward of a whole net in PLAIN autograd use (no reducer, no flat buffer): -> (outputs, gradients,
    number of backbone-chain calls)."""
    import importlib
    from mmnas_amd import ops
    from mmnas.model.mixed import MixedOp
    monkeypatch.setenv('MMNAS_AUTOGRAD_CHAIN', '1' if autograd_chain else '0')
    c = cases.net_case(task, arch, 31337, search=search, B=3, Sx=6, Sy=9)
    c['cfg'].DROPOUT_R = dropout
    mod = importlib.import_module('mmnas.model.%s_%s' % ('hygr' if search else 'full', task))
    net = (mod.Net_Search if search else mod.Net_Full)(c['cfg'], {'token_size': c['token_size'], 'ans_size': c['ans_size'],
           'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)})
    net.load_state_dict({k: T(v) for k, v in c['P'].items()})
    net = net.to(DEV).train()
    calls = []
    orig = ops.BackboneFn.apply
    monkeypatch.setattr(ops.BackboneFn, 'apply', lambda *a: (calls.append(1), orig(*a))[1])
    ops.manual_seed(99)
    try:
        if search:
            MixedOp.MODE = mode
            net.set_sampled(plan)
            net.unused_modules_off()
        pred = net(tuple(T(a).to(DEV) for a in c['inputs']))
        tgt = T(c['target']).to(DEV)
        if task == 'vgd':
            loss = (pred[0] * tgt).sum() + 0.5 * (pred[1] ** 2).sum()
            out = torch.cat([pred[0].reshape(-1), pred[1].reshape(-1)])
        elif task == 'itm':
            loss, out = torch.nn.functional.binary_cross_entropy(pred, tgt, reduction='sum'), pred
        else:
            loss, out = torch.nn.functional.binary_cross_entropy_with_logits(pred, tgt, reduction='sum'), pred
        loss += 0 * sum(p.sum() for p in net.parameters())           # (the scripts' line: every parameter gets a gradient)
        net.zero_grad()
        loss.backward()
        if search:
            net.unused_modules_back()
        torch.cuda.synchronize()
        grads = {k: (p.grad.detach().cpu().numpy().copy() if p.grad is not None else None) for k, p in net.named_parameters()}
        return out.detach().cpu().numpy(), grads, len(calls)
    finally:
        MixedOp.MODE = None
        monkeypatch.setattr(ops.BackboneFn, 'apply', orig)


@pytest.mark.parametrize('task,arch', [('vqa', 'mmnas_vqa'), ('vqa', 'mcan'), ('vgd', 'mmnas_vgd'), ('itm', 'mmnas_itm')])
def test_plain_autograd_takes_the_chain_and_equals_the_per_operator_nodes(task, arch, monkeypatch):
    """Round 5: without a flat gradient buffer the backbone is still ONE autograd node per direction -- its parameters are
    autograd inputs, the kernels accumulate into a per-call buffer whose views backward returns (ops.autograd_chain_enabled).
    Same dropout masks (the seeds are drawn in operator order on both routes), same logits, every parameter gradient equal to
    the per-operator nodes' (the `0 * sum(p.sum())` line of the scripts included: every parameter ends with a gradient)."""
    ref = _plain_autograd(task, arch, False, monkeypatch, False)
    got = _plain_autograd(task, arch, False, monkeypatch, True)
    assert ref[2] == 0 and got[2] == 1
    assert all(g is not None for g in got[1].values())
    _compare(got, ref)


@pytest.mark.parametrize('task', ['vqa', 'itm'])
def test_plain_autograd_chain_supernet_weight_step_and_arch_mode(task, monkeypatch):
    pl = cases.search_plan(np.random.RandomState(2), None)
    flat = pl['enc'] + pl['dec']
    ref = _plain_autograd(task, None, True, monkeypatch, False, plan=flat)
    got = _plain_autograd(task, None, True, monkeypatch, True, plan=flat)
    assert ref[2] == 0 and got[2] == 1
    _compare(got, ref)
    # MODE 'full' (the arch step) needs the gate blocks of a flat buffer: plain autograd keeps one node per candidate
    pf = cases.search_plan(np.random.RandomState(3), 'full')
    arch = _plain_autograd(task, None, True, monkeypatch, True, plan=pf['enc'] + pf['dec'], mode='full', dropout=0.0)
    assert arch[2] == 0


def test_plain_autograd_chain_is_skipped_without_grad(monkeypatch):
    """Opt-in (MMNAS_AUTOGRAD_CHAIN=1; off by default: measured neutral on the host-bound unchanged loop); under no_grad the
    per-operator path serves the call either way."""
    from mmnas_amd import ops
    monkeypatch.setenv('MMNAS_AUTOGRAD_CHAIN', '1')
    from mmnas.model.full_vqa import Net_Full
    calls = []
    orig = ops.BackboneFn.apply
    monkeypatch.setattr(ops.BackboneFn, 'apply', lambda *a: (calls.append(1), orig(*a))[1])
    c = cases.net_case('vqa', 'mmnas_vqa', 5)
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = Net_Full(c['cfg'], init).to(DEV).eval()
    with torch.no_grad():
        out = net(tuple(T(a).to(DEV) for a in c['inputs']))
    assert not calls and bool(torch.isfinite(out).all())
    net.train()
    net(tuple(T(a).to(DEV) for a in c['inputs'])).sum().backward()
    assert len(calls) == 1 and net.proj.weight.grad is not None
    monkeypatch.setenv('MMNAS_AUTOGRAD_CHAIN', '0')          # the default: one autograd node per operator
    net(tuple(T(a).to(DEV) for a in c['inputs'])).sum().backward()
    assert len(calls) == 1


def test_chain_full_size_side_stream_repeatable(monkeypatch):
    """B = 64, 100 regions, d = 512 (BASELINE configs[1]): the side-stream run equals the single-stream run, three
    times in a row (a missing dependency between the two streams shows up as a changing result)."""
    import json, os
    from types import SimpleNamespace
    from mmnas_amd import dp, ops
    from mmnas.model.full_vqa import Net_Full
    from tests.util import REPO
    g = json.load(open(os.path.join(REPO, 'arch', 'mmnas_vqa.json')))
    cfg = SimpleNamespace(DROPOUT_R=0.1, REL_SIZE=64, OPS_NORM=True, OPS_RESIDUAL=True, LAYERS=1, NODES={'enc': 12, 'dec': 18},
                          ATTFLAT_GLIMPSES=1, ATTFLAT_MLP_SIZE=512, FRCNFEAT_SIZE=2048, BBOX_FEATURE=False, BBOXFEAT_EMB_SIZE=1024,
                          WORD_EMBED_SIZE=300, ALPHA_INIT_TYPE='normal', SCORES_LOSS='kld', HSIZE=512, ATTFLAT_OUT_SIZE=1024,
                          GENOTYPE=g[sorted(g)[-1]])
    V, ANS, B = 2000, 3129, 64
    torch.manual_seed(0)
    init = {'token_size': V, 'ans_size': ANS, 'pretrained_emb': torch.randn(V, 300).numpy()}
    net = Net_Full(cfg, init).to(DEV).train()
    gen = torch.Generator().manual_seed(4)
    frcn = torch.relu(torch.randn(B, 100, 2048, generator=gen)); frcn[:, 80:] = 0
    inp = (frcn.to(DEV), torch.zeros(B, 100, 5, device=DEV), torch.randn(B, 100, 100, 4, generator=gen).to(DEV),
           torch.randint(1, V, (B, 14), generator=gen).to(DEV), torch.randn(B, 14, 14, 3, generator=gen).to(DEV))
    tgt = (torch.rand(B, ANS, generator=gen) * (torch.rand(B, ANS, generator=gen) < 0.003)).to(DEV)
    red = dp.GradReducer(list(net.parameters()))
    results = []
    try:
        for side in (False, True, True, True):
            monkeypatch.setenv('MMNAS_SIDE_STREAM', '1' if side else '0')
            ops.manual_seed(7)
            red.begin_step()
            loss = torch.nn.functional.binary_cross_entropy_with_logits(net(inp), tgt, reduction='sum')
            loss.backward()
            red.finish()
            torch.cuda.synchronize()
            results.append((float(loss.detach()), red.fg.flat.clone()))
    finally:
        red.fg.disable_sinks()
    ref_loss, ref = results[0]
    scale = float(ref.abs().max())
    for loss, flat in results[1:]:
        assert loss == ref_loss
        assert float((flat - ref).abs().max()) <= 2e-5 * scale


# ----------------------------------------------------------------------------------------------------------------------
# Ragged decoder stream (MMNAS_UNPAD / ops.set_unpad): the chain on the valid region rows only
# ----------------------------------------------------------------------------------------------------------------------
def _run_unpad(task, arch, search, unpad, mode=None, plan=None, B=5, Sy=23, HSIZE=128):
    """One forward + backward of a whole net through the harness-style flat gradient buffer; returns (outputs, flat
    gradients by parameter name, whether the chain ran ragged)."""
    import importlib
    from mmnas_amd import dp, ops
    from mmnas.model.mixed import MixedOp
    c = cases.net_case(task, arch, 4242, search=search, B=B, Sx=6, Sy=Sy, HSIZE=HSIZE)
    c['cfg'].DROPOUT_R = 0.0
    mod = importlib.import_module('mmnas.model.%s_%s' % ('hygr' if search else 'full', task))
    cls = mod.Net_Search if search else mod.Net_Full
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = cls(c['cfg'], init)
    net.load_state_dict({k: T(v) for k, v in c['P'].items()})
    net = net.to(DEV).train()
    inp = tuple(T(a).to(DEV) for a in c['inputs'])
    tgt = T(c['target']).to(DEV)
    seen = []
    orig = ops.BackboneFn.apply
    prev = ops.set_unpad(unpad)
    ops.BackboneFn.apply = lambda *a: (seen.append(a[10] is not None), orig(*a))[1]
    red = None
    try:
        if search:
            MixedOp.MODE = mode
            net.set_sampled(plan)
            red = dp.SupernetReducer(net)
            if mode is None:
                red.begin_weight_step()
            else:
                net.begin_arch_step()
                red.fg.zero()
                red.fg.attach()
        else:
            red = dp.GradReducer(list(net.parameters()))
            red.begin_step()
        pred = net(inp)
        if task == 'itm':
            loss = torch.nn.functional.binary_cross_entropy(pred, tgt, reduction='sum')
        else:
            loss = torch.nn.functional.binary_cross_entropy_with_logits(pred, tgt, reduction='sum')
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: (p.grad.detach().cpu().numpy().copy() if p.grad is not None else None) for k, p in net.named_parameters()}
        return pred.detach().cpu().numpy(), grads, seen
    finally:
        MixedOp.MODE = None
        ops.BackboneFn.apply = orig
        ops.set_unpad(prev)
        if red is not None:
            red.fg.disable_sinks()


def _same(a, b, gtol=4e-5):   # (summation order: packed and padded batches cut their weight-gradient reductions differently)
    out_a, g_a, _ = a
    out_b, g_b, _ = b
    assert rel_err(out_a, out_b) < 2e-6
    top = max(float(np.abs(g).max()) for g in g_b.values() if g is not None)
    for k in g_b:
        if g_b[k] is None:
            assert g_a[k] is None or not np.any(g_a[k]), k
            continue
        assert g_a[k] is not None, k
        diff = float(np.abs(g_a[k] - g_b[k]).max())
        t = max(gtol, REL_PATH_SELF_TOL) if is_rel_path(k) else gtol
        # (floor: gradients that are small against the largest one in the net are compared on that scale -- 1e-3 of it, 3e-3
        #  for the cancellation-limited relation path)
        floor = (3e-3 if is_rel_path(k) else 1e-3) * top
        assert diff <= t * max(float(np.abs(g_b[k]).max()), floor), (k, diff, float(np.abs(g_b[k]).max()))


@pytest.mark.parametrize('task,arch', [('vqa', 'mmnas_vqa'), ('vqa', 'mcan'), ('itm', 'mmnas_itm')])
def test_ragged_decoder_stream_equals_the_padded_computation_net_full(task, arch):
    """Net_Full: logits and EVERY parameter gradient of the chain on the valid region rows only (packed rows, attention over
    each sample's own rows, relation bias over its n_b x n_b corner) equal those of the padded computation -- the padding
    rows are masked keys and AttFlat-masked outputs in the reference (hygr_vqa.py:113-122, modules.py:195-196)."""
    padded = _run_unpad(task, arch, False, False)
    ragged = _run_unpad(task, arch, False, True)
    assert padded[2] == [False] and ragged[2] == [True]
    # (round 5: the head pools over packed rows too -- AttFlat's glimpse-logit weight gradient, a cancelling sum over the
    #  region rows, is reduced in another order: 9e-8 of the net's largest gradient entry apart)
    _same(ragged, padded, gtol=1e-4)


@pytest.mark.parametrize('mode,B,Sy', [(None, 5, 23), ('full', 5, 23), (None, 1, 40), ('full', 2, 128), (None, 3, 33)])
def test_ragged_decoder_stream_equals_the_padded_computation_supernet(mode, B, Sy):
    """Net_Search weight step (MODE None) and architecture step (MODE 'full': every candidate forward, gate gradients); a
    single sample, the longest supported sequences (128 rows), odd row counts."""
    plan = cases.search_plan(np.random.RandomState(5), mode)
    flat = plan['enc'] + plan['dec']
    padded = _run_unpad('vqa', None, True, False, mode, flat, B=B, Sy=Sy)
    ragged = _run_unpad('vqa', None, True, True, mode, flat, B=B, Sy=Sy)
    assert padded[2] == [False] and ragged[2] == [True]
    _same(ragged, padded)
    if mode == 'full':      # the gate gradients live in the alpha_gate parameters' gradients: compared above by name
        assert any('alpha_gate' in k and g is not None and np.any(g) for k, g in ragged[1].items())


def test_grounding_head_keeps_the_padded_computation():
    """VGD scores every region row, padding included (full_vgd.py:105-114): the ragged stream must not engage."""
    from mmnas_amd import ops
    prev = ops.set_unpad(True)
    try:
        out = _run_unpad_vgd()
    finally:
        ops.set_unpad(prev)
    assert out == [False]


def _run_unpad_vgd():
    from mmnas_amd import dp, ops
    from mmnas.model.full_vgd import Net_Full
    c = cases.net_case('vgd', 'mmnas_vgd', 77, B=3, Sx=6, Sy=11)
    c['cfg'].DROPOUT_R = 0.0
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = Net_Full(c['cfg'], init)
    net.load_state_dict({k: T(v) for k, v in c['P'].items()})
    net = net.to(DEV).train()
    seen = []
    orig = ops.BackboneFn.apply
    ops.BackboneFn.apply = lambda *a: (seen.append(a[10] is not None), orig(*a))[1]
    red = dp.GradReducer(list(net.parameters()))
    try:
        red.begin_step()
        ps, pr = net(tuple(T(a).to(DEV) for a in c['inputs']))
        (ps.sum() + pr.sum()).backward()
        torch.cuda.synchronize()
    finally:
        ops.BackboneFn.apply = orig
        red.fg.disable_sinks()
    return seen


def test_ragged_info_sources_and_refusals():
    """ops.ragged_info_for: lengths from the mask (one device-to-host copy per distinct features tensor, cached on the
    tensor object + version) or from `_mmnas_lengths` (the data pipeline's region counts); None -- i.e. the padded
    computation -- when the valid rows are not a prefix, a sample is empty, or nothing is padded."""
    from mmnas_amd import ops
    from mmnas_amd.model.nets import make_mask
    prev = ops.set_unpad(True)
    try:
        g = torch.Generator().manual_seed(3)
        feat = torch.rand(4, 9, 16, generator=g).to(DEV) + 0.1
        lens = [9, 3, 1, 6]
        for b, n in enumerate(lens):
            feat[b, n:] = 0
        a = ops.ragged_info_for(feat, make_mask(feat))
        assert a is not None and a.lengths == tuple(lens) and a.N == 19 and a.off.tolist() == [0, 9, 12, 13, 19]
        assert a.tile_off.tolist() == [0, 3, 4, 5, 7] and a.ntiles == 7
        assert ops.ragged_info_for(feat, make_mask(feat)) is a                      # cached
        feat.mul_(2.0)                                                               # written: looked at again
        assert ops.ragged_info_for(feat, make_mask(feat)) is not a
        f2 = feat.clone()
        f2._mmnas_lengths = torch.tensor(lens)
        b2 = ops.ragged_info_for(f2, None)
        assert b2.off.tolist() == a.off.tolist()
        hole = feat.clone(); hole[0, 2] = 0                                          # a zero row in the middle: not a prefix
        assert ops.ragged_info_for(hole, make_mask(hole)) is None
        empty = feat.clone(); empty[1] = 0
        assert ops.ragged_info_for(empty, make_mask(empty)) is None
        full = torch.rand(2, 5, 8, generator=g).to(DEV) + 0.1
        assert ops.ragged_info_for(full, make_mask(full)) is None
        ops.set_unpad(False)
        assert ops.ragged_info_for(feat, make_mask(feat)) is None
    finally:
        ops.set_unpad(prev)


# ----------------------------------------------------------------------------------------------------------------------
# Relation bias of all relation operators of a stream in one launch per direction (relmulti.hip; mmnas_set_rel_hoist)
# ----------------------------------------------------------------------------------------------------------------------
def _rel_heavy_plan(mode, n_rel):
    """A sampled architecture whose decoder takes the relation operator (candidate 1 of dec_safe) at n_rel nodes."""
    plan = cases.search_plan(np.random.RandomState(9), mode)
    dec = []
    for k, (act, inact) in enumerate(plan['dec']):
        a = 1 if k < n_rel else (act[0] if act[0] != 1 else 0)
        dec.append(([a], [i for i in range(4) if i != a]))
    return plan['enc'] + dec


@pytest.mark.parametrize('mode,n_rel,unpad', [(None, 6, False), (None, 18, False), (None, 1, False), ('full', 5, False),
                                              (None, 7, True), ('full', 4, True)])
def test_hoisted_relation_bias_equals_the_per_operator_launches(mode, n_rel, unpad):
    """Supernet weight and architecture steps with the relation bias of every (sampled / evaluated) relation operator
    computed by ONE mmnas_rel_multi launch per direction against one mmnas_rel_fused launch per operator: same logits,
    every parameter gradient to round-off -- padded and ragged decoder streams; 18 relation operators = 3 row tiles
    forward, 3 launches backward at 4 heads ... here HSIZE 128: 2 heads, 36 rows."""
    from mmnas_amd import _lib as L
    lib = L.lib()
    flat = _rel_heavy_plan(mode, n_rel)
    outs = []
    for hoist in (0, 1):
        prev = lib.mmnas_set_rel_hoist(hoist)
        try:
            outs.append(_run_unpad('vqa', None, True, unpad, mode, flat, B=4, Sy=19))
        finally:
            lib.mmnas_set_rel_hoist(prev)
    assert outs[0][2] == [unpad] and outs[1][2] == [unpad]
    # (forward: a row's bias is the same bit for bit whichever rows share its launch; backward: other summation orders in
    #  the relation-path gradients only)
    assert np.array_equal(outs[1][0], outs[0][0])
    _same(outs[1], outs[0])
    assert any('linear_r.weight' in k and g is not None and np.any(g) for k, g in outs[1][1].items())
    assert np.any(outs[1][1]['linear_y_rel.weight'])


@pytest.mark.parametrize('task,arch', [('vqa', 'mmnas_vqa'), ('vgd', 'mmnas_vgd'), ('itm', 'mmnas_itm')])
def test_hoisted_relation_bias_net_full(task, arch, monkeypatch):
    """The fixed architectures of arch/*.json (4-5 relation operators each) through the chain with and without the hoist."""
    from mmnas_amd import _lib as L
    lib = L.lib()
    outs = []
    for hoist in (0, 1):
        prev = lib.mmnas_set_rel_hoist(hoist)
        try:
            outs.append(_run(task, arch, False, True, False, monkeypatch, dropout=0.1))
        finally:
            lib.mmnas_set_rel_hoist(prev)
    assert outs[0][2] == 1 and outs[1][2] == 1
    out_a, g_a, _ = outs[1]
    out_b, g_b, _ = outs[0]
    assert rel_err(out_a, out_b) < 2e-6
    top = max(float(np.abs(g).max()) for g in g_b.values() if g is not None)
    for k in g_b:
        if g_b[k] is None:
            continue
        diff = float(np.abs(g_a[k] - g_b[k]).max())
        assert diff <= (REL_PATH_SELF_TOL if is_rel_path(k) else 2e-5) * max(float(np.abs(g_b[k]).max()), 1e-3 * top), (k, diff)


# ----------------------------------------------------------------------------------------------------------------------
# Key / value projections of all guided operators in grouped launches (mmnas_set_guided_hoist)
# ----------------------------------------------------------------------------------------------------------------------
def _guided_heavy_plan(mode, n_guided, first=2):
    """A sampled architecture whose decoder takes the guided operator (candidate 2 of dec_safe) at n_guided nodes, the first
    of them at node `first`."""
    plan = cases.search_plan(np.random.RandomState(11), mode)
    dec = []
    for k, (act, inact) in enumerate(plan['dec']):
        a = 2 if first <= k < first + n_guided else (act[0] if act[0] != 2 else 3)
        dec.append(([a], [i for i in range(4) if i != a]))
    return plan['enc'] + dec


@pytest.mark.parametrize('mode,n_guided,first,unpad', [(None, 5, 2, False), (None, 9, 0, False), (None, 16, 1, False), (None, 1, 3, False),
                                                       ('full', 4, 2, False), (None, 6, 1, True), ('full', 3, 0, True)])
def test_hoisted_guided_projections_equal_the_per_operator_launches(mode, n_guided, first, unpad):
    """The key / value projections of all guided operators of a chain as grouped launches behind the encoder, their gradients
    as grouped gradient-pair launches behind the last guided operator's backward + ONE sum of the key / value source
    gradients -- against one launch set per operator.  Forward: the same products in other launches (bit-equal logits);
    backward: the language state's gradient is summed in another order."""
    from mmnas_amd import _lib as L
    lib = L.lib()
    flat = _guided_heavy_plan(mode, n_guided, first)
    outs = []
    for hoist in (0, 1):
        prev = lib.mmnas_set_guided_hoist(hoist)
        try:
            outs.append(_run_unpad('vqa', None, True, unpad, mode, flat, B=4, Sy=19))
        finally:
            lib.mmnas_set_guided_hoist(prev)
    assert outs[0][2] == [unpad] and outs[1][2] == [unpad]
    assert rel_err(outs[1][0], outs[0][0]) < 1e-6
    _same(outs[1], outs[0])
    assert sum(1 for k, g in outs[1][1].items() if 'linear_k.weight' in k and g is not None and np.any(g)) >= n_guided


@pytest.mark.parametrize('task,arch', [('vqa', 'mmnas_vqa'), ('vqa', 'mcan'), ('itm', 'mmnas_itm')])
def test_hoisted_guided_projections_net_full(task, arch, monkeypatch):
    from mmnas_amd import _lib as L
    lib = L.lib()
    outs = []
    for hoist in (0, 1):
        prev = lib.mmnas_set_guided_hoist(hoist)
        try:
            outs.append(_run(task, arch, False, True, False, monkeypatch, dropout=0.1))
        finally:
            lib.mmnas_set_guided_hoist(prev)
    _compare(outs[1], outs[0])


@pytest.mark.parametrize('task,arch,search,mode', [('vqa', 'mmnas_vqa', False, None), ('itm', 'mmnas_itm', False, None),
                                                   ('vqa', None, True, None), ('vqa', None, True, 'full')])
def test_ragged_stream_runs_stem_and_head_on_packed_rows(task, arch, search, mode, monkeypatch):
    """Round 5: with the ragged decoder stream on, the stem projects the valid region rows only (packed), the chain takes and
    returns packed rows, and AttFlat pools over each sample's own rows (mmnas_attflat_side.off) -- no pack / unpack launch,
    no padded row computed anywhere; logits and every parameter gradient equal the padded computation's (the imgfeat_linear
    and AttFlat parameters among them)."""
    from mmnas_amd import ops
    packs, heads, chains = [], [], []
    o_pack, o_unpack, o_head, o_chain = ops.pack_rows, ops.unpack_rows, ops.HeadFn.apply, ops.BackboneFn.apply
    monkeypatch.setattr(ops, 'pack_rows', lambda x, rg: (packs.append(tuple(x.shape)), o_pack(x, rg))[1])
    monkeypatch.setattr(ops, 'unpack_rows', lambda *a: (packs.append('unpack'), o_unpack(*a))[1])
    monkeypatch.setattr(ops.HeadFn, 'apply', lambda *a: (heads.append(len(a) > 6 and a[6] is not None), o_head(*a))[1])
    plan = None
    if search:
        pl = cases.search_plan(np.random.RandomState(5), mode)
        plan = pl['enc'] + pl['dec']
    padded = _run_unpad(task, arch, search, False, mode, plan)
    assert heads == [False] and not packs
    ragged = _run_unpad(task, arch, search, True, mode, plan)
    assert heads == [False, True], heads                       # the head took packed image rows
    assert len(packs) == 1 and packs[0][-1] == 32, packs       # ONE pack: the raw region features (FRCNFEAT_SIZE 32 here)
    assert ragged[2] == [True]
    _same(ragged, padded, gtol=1e-4)
    for k in ('imgfeat_linear.weight', 'imgfeat_linear.bias', 'attflat_y.mlp.fc.linear.weight', 'attflat_y.linear_merge.weight'):
        assert ragged[1][k] is not None and np.any(ragged[1][k]), k


@pytest.mark.parametrize('mode,unpad', [(None, False), ('full', False), (None, True)])
def test_relation_launches_beside_the_encoder_give_the_same_result(mode, unpad):
    """mmnas_set_rel_overlap: the image stream's relation-bias launches on the chain's side stream beside the language
    stream's operators (fork / join by events) against everything on one stream -- the same kernels on the same data, so
    every number is bit-equal; three repetitions (a missing dependency between the streams shows as a changing result)."""
    from mmnas_amd import _lib as L
    lib = L.lib()
    flat = _rel_heavy_plan(mode, 6)
    outs = []
    for on in (0, 1, 1, 1):
        prev = lib.mmnas_set_rel_overlap(on)
        try:
            outs.append(_run_unpad('vqa', None, True, unpad, mode, flat, B=4, Sy=19))
        finally:
            lib.mmnas_set_rel_overlap(prev)
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0])
        for k, g in outs[0][1].items():
            if g is None:
                continue
            # (bit-equal except where split-K weight-gradient pieces add with float atomics in hardware order)
            assert np.allclose(o[1][k], g, rtol=0, atol=2e-5 * max(float(np.abs(g).max()), 1e-12)), k
            if 'linear_r' in k or '_rel.' in k:
                assert np.array_equal(o[1][k], g), k

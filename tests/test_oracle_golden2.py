"""Pin the CPU oracle's step harness (optimizer trajectory of the bilevel loop, task losses) and its loader-function
restatements against golden vectors produced by the imported reference (tests/golden/make_golden.py, captures
traj / losses / loader), and guard the golden recipe itself: when the reference tree is present (build container
only) a subset is regenerated and compared bit for bit with the committed files."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import mmnas_oracle as O
from tests import oracle_runner as R
from tests.golden import cases
from tests.util import GOLDEN, REPO, esample, load, rel_err

T = torch.from_numpy


# ---------------------------------------------------------------------------------------------------------------------
# capture (v): two Adam weight steps + one 'full' arch step (search_vqa.py:279-337)
# ---------------------------------------------------------------------------------------------------------------------
def oracle_trajectory():
    c, c2, plans = cases.traj_setup()
    H = cases.TRAJ_HYPER
    cfg = c['cfg']
    P = {k: T(v).clone() for k, v in c['P'].items()}
    net_keys = [k for k in P if 'alpha' not in k]
    gate_keys = [k for k in P if k.endswith('alpha_gate')]
    prob_keys = [k.replace('alpha_gate', 'alpha_prob') for k in gate_keys]
    net_adam = O.Adam({k: P[k] for k in net_keys}, H['net_betas'], H['net_eps'])
    alpha_adam = O.Adam({k: P[k] for k in prob_keys}, H['alpha_betas'], 1e-8)
    inp = tuple(T(a) for a in c['inputs']); tgt = T(c['target'])
    inp2 = tuple(T(a) for a in c2['inputs']); tgt2 = T(c2['target'])
    res = {'losses': [], 'gnorms': [], 'snap': {}}

    def fwd_bwd(plan, inputs, target):
        flat = plan['enc'] + plan['dec']
        for k, (act, _) in zip(gate_keys, flat):
            P[k].zero_()
            P[k][act[0]] = 1.0
        Q = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
        loss = O.bce_with_logits_sum(O.net_forward('vqa', Q, cfg, inputs, search=plan), target)
        loss.backward()
        res['losses'].append(float(loss.detach()))
        return {k: q.grad for k, q in Q.items()}

    step = 0
    for i in (0, 1):
        g = fwd_bwd(plans[i], inp, tgt)
        grads = {k: g[k] for k in net_keys if g[k] is not None}
        res['gnorms'].append(O.clip_grad_norm(list(grads.values()), H['clip']))
        step += 1
        net_adam.step(grads, O.warmup_rate(H['net_lr'], step, H['epoch_steps']))
        res['snap']['w%d' % (i + 1)] = {k: P[k].clone() for k in net_keys}
    g = fwd_bwd(plans[2], inp2, tgt2)
    res['gate_grads'] = np.stack([np.pad(g[k].numpy(), (0, 4 - g[k].numel())) for k in gate_keys])
    pg = {pk: O.alpha_prob_grad_full(P[pk], g[k]) for k, pk in zip(gate_keys, prob_keys)}
    res['prob_grads'] = np.stack([np.pad(pg[pk].numpy(), (0, 4 - pg[pk].numel())) for pk in prob_keys])
    alpha_adam.step(pg, H['alpha_lr'])
    res['alpha_after'] = np.stack([np.pad(P[pk].numpy(), (0, 4 - P[pk].numel())) for pk in prob_keys])
    res['snap']['a'] = {k: P[k].clone() for k in net_keys}
    fwd_bwd(plans[3], inp, tgt)
    res['P0'] = {k: T(c['P'][k]) for k in net_keys}
    return res


# AttFlat's glimpse-logit bias sits in front of a softmax over the sequence (modules.py:78-82): its gradient is
# mathematically zero, what arrives is round-off, and Adam (eps 1e-9) turns round-off into full +-lr steps whose
# direction depends on summation order.  The softmax ignores the shift, so nothing downstream sees it.
SHIFT_INVARIANT = ('attflat_x.mlp.linear.bias', 'attflat_y.mlp.linear.bias')
# The hidden layer in front of those logits inherits a weaker form of it: a unit that is active on every unmasked row
# has the bias gradient W2[j] * sum_rows(dlogits) = 0 up to round-off (a softmax's input gradients sum to zero), and
# for |g| ~ eps = 1e-9 Adam's step length itself depends on the round-off.  A few of the 64 units are of that kind:
# the tensor's motion is compared at 10 % instead of 2 %.
NEAR_INVARIANT = ('attflat_x.mlp.fc.linear.bias', 'attflat_y.mlp.fc.linear.bias')


def check_trajectory(res, tol_loss=2e-4, tol_delta=2e-2, fname='traj.npz'):
    """Shared with the GPU replay (tests/test_harness_gpu.py::test_bilevel_trajectory_vs_reference_loop).  Adam normalises every coordinate's first step to +-lr,
    so coordinates whose gradient is round-off-sized move by a full step in a direction the summation order decides:
    parameter motion is compared as per-tensor delta norms and, for the listed small tensors, element-wise."""
    npz = load(fname)
    ref = npz['traj|losses']
    for i, (a, b) in enumerate(zip(res['losses'], ref)):
        assert abs(a - b) <= tol_loss * abs(b), ('loss', i, a, b)
    assert rel_err(np.array(res['gnorms']), npz['traj|grad_norms']) < 1e-3
    assert rel_err(res['gate_grads'], npz['traj|arch|gate_grads']) < 1e-3
    assert rel_err(res['prob_grads'], npz['traj|arch|prob_grads']) < 1e-3
    assert rel_err(res['alpha_after'], npz['traj|arch|alpha_after']) < 1e-3
    for tag in ('w1', 'w2', 'a'):
        keys = [str(k) for k in npz['traj|%s|keys' % tag]]
        dn = npz['traj|%s|delta_norm' % tag]
        snap = res['snap'][tag]
        assert set(keys) == set(snap.keys())
        for k, n in zip(keys, dn):
            if k in SHIFT_INVARIANT:
                continue
            mine = float((snap[k].double() - res['P0'][k].double()).norm())
            tol = 0.1 if k in NEAR_INVARIANT else tol_delta
            assert abs(mine - n) <= tol * n + 1e-7, (tag, k, mine, n)
        # element-wise anchors: strided samples of every tensor's motion.  A coordinate whose clipped gradient is
        # round-off-sized takes its +-lr Adam step in a direction the summation order decides, so single coordinates may
        # differ: per tensor at most a tenth of the samples, over all tensors at most 1 %
        off = npz['traj|%s|delta_off' % tag]
        ds = npz['traj|%s|delta_sample' % tag]
        bad = total = 0
        for i, k in enumerate(keys):
            if k in SHIFT_INVARIANT or k in NEAR_INVARIANT:
                continue
            want = ds[off[i]:off[i + 1]]
            mine = esample((snap[k].double() - res['P0'][k].double()).numpy())
            miss = int(np.sum(np.abs(mine - want) > 5e-2 * np.abs(want).max() + 1e-7))
            assert miss <= max(1, want.size // 10), (tag, k, miss, want.size)
            bad += miss
            total += want.size
        assert total > 5000 and bad <= 0.01 * total, (tag, bad, total)
        for k in cases.TRAJ_FULL_KEYS:
            want = npz['traj|%s|P:%s' % (tag, k)]
            d0 = np.abs(want - res['P0'][k].numpy()).max()
            # (5 % of the largest step: a coordinate whose clipped gradient is within ~50 eps of zero takes a step whose
            #  LENGTH depends on the summation order of that gradient)
            assert np.abs(snap[k].numpy() - want).max() <= 5e-2 * d0 + 1e-7, (tag, k)
    # the arch step leaves the network weights alone
    for k in res['snap']['a']:
        assert torch.equal(res['snap']['a'][k], res['snap']['w2'][k]), k


def oracle_train_trajectory():
    """The oracle's restatement of train_vqa.py:291-311 (zero_grad, forward, BCE sum, backward, clip, warm-up Adam step),
    five steps with a decay(0.2) before the last -- see make_golden.gen_train_traj."""
    c, c2 = cases.train_traj_setup()
    H = cases.TRAIN_HYPER
    cfg = c['cfg']
    P = {k: T(v).clone() for k, v in c['P'].items()}
    adam = O.Adam(P, H['betas'], H['eps'])
    batches = [(tuple(T(a) for a in c['inputs']), T(c['target'])), (tuple(T(a) for a in c2['inputs']), T(c2['target']))]
    res = {'losses': [], 'gnorms': [], 'rates': [], 'snap': {}, 'P0': {k: T(v) for k, v in c['P'].items()}}
    lr_base = H['lr']
    for i in range(5):
        if i == 4:
            lr_base *= H['decay_r']
        inp, tgt = batches[i % 2]
        Q = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
        loss = O.bce_with_logits_sum(O.net_forward('vqa', Q, cfg, inp, genotype=c['genotype']), tgt)
        loss.backward()
        res['losses'].append(float(loss.detach()))
        grads = {k: q.grad for k, q in Q.items() if q.grad is not None}
        res['gnorms'].append(O.clip_grad_norm(list(grads.values()), H['clip']))
        rate = O.warmup_rate(lr_base, i + 1, H['epoch_steps'])
        res['rates'].append(rate)
        adam.step(grads, rate)
        if i in (0, 3, 4):
            res['snap']['s%d' % (i + 1)] = {k: v.clone() for k, v in P.items()}
    return res


def check_train_trajectory(res, tol_loss=2e-4, tol_delta=2e-2, tol_near=0.3, fname='train_traj.npz', stray=0.0):
    """Shared with the GPU replay (tests/test_harness_gpu.py); tolerances as for check_trajectory, except that the
    round-off-driven AttFlat hidden biases (NEAR_INVARIANT) have five steps instead of two to drift: 13 % measured for
    the fp64-free oracle against the reference at step 4."""
    npz = load(fname)
    for i, (a, b) in enumerate(zip(res['losses'], npz['train|losses'])):
        assert abs(a - b) <= tol_loss * abs(b), ('loss', i, a, b)
    assert rel_err(np.array(res['gnorms']), npz['train|grad_norms']) < 1e-3
    assert np.allclose(np.array(res['rates']), npz['train|rates'], rtol=1e-12, atol=0)
    for tag in ('s1', 's4', 's5'):
        keys = [str(k) for k in npz['train|%s|keys' % tag]]
        dn = npz['train|%s|delta_norm' % tag]
        snap = res['snap'][tag]
        assert set(keys) == set(snap.keys())
        for k, n in zip(keys, dn):
            if k in SHIFT_INVARIANT:
                continue
            mine = float((snap[k].double() - res['P0'][k].double()).norm())
            tol = tol_near if k in NEAR_INVARIANT else tol_delta
            assert abs(mine - n) <= tol * n + 1e-7, (tag, k, mine, n)
        for k in cases.TRAIN_FULL_KEYS:
            want = npz['train|%s|P:%s' % (tag, k)]
            d0 = np.abs(want - res['P0'][k].numpy()).max()
            err = np.abs(snap[k].numpy() - want)
            # `stray` (the B = 64 file only): the fraction of a tensor's coordinates that may miss the 5 % -- Adam normalises every
            # coordinate's step to ~lr, so a coordinate whose clipped gradient is round-off-sized takes a step whose length and sign
            # the summation order decides (check_trajectory's element anchors allow the same); none may be further off than a
            # full step in the other direction
            bad = err > 5e-2 * d0 + 1e-7
            TRAIN_STRAYS[(fname, tag, k)] = (int(bad.sum()), int(bad.size), float(err.max() / max(d0, 1e-30)))
            assert bad.mean() <= stray and err.max() <= 2.1 * d0 + 1e-7, (tag, k, int(bad.sum()), bad.size, float(err.max()), float(d0))


TRAIN_STRAYS = {}


def test_oracle_training_loop_vs_reference_trajectory():
    check_train_trajectory(oracle_train_trajectory())


def test_reference_loop_moves_unsampled_candidates():
    """The property the ADVICE review pointed at: under the reference loop a candidate that was sampled at step 1 but
    not at step 2 still moves at step 2 (zero gradient, stale momentum) -- its delta norm grows between w1 and w2."""
    npz = load('traj.npz')
    p0, p1 = npz['traj|plan0'], npz['traj|plan1']
    node = next(i for i in range(12) if p0[i] != p1[i])
    key = 'backnone.cells_enc.0.dag.%d.0.candidate_ops.%d.ln.a_2' % (node, p0[node])
    keys = [str(k) for k in npz['traj|w1|keys']]
    i = keys.index(key)
    assert npz['traj|w2|delta_norm'][i] > 1.2 * npz['traj|w1|delta_norm'][i] > 0


def test_optimizer_trajectory_vs_reference_loop():
    check_trajectory(oracle_trajectory())


# ---------------------------------------------------------------------------------------------------------------------
# task losses
# ---------------------------------------------------------------------------------------------------------------------
def _gradnorm_check(npz, tag, grads, P):
    keys = [str(k) for k in npz[tag + 'gradnorm_keys']]
    norms = npz[tag + 'gradnorms']
    assert set(keys) == set(P.keys())
    for k, n in zip(keys, norms):
        mine = 0.0 if grads[k] is None else float(np.linalg.norm(np.asarray(grads[k], np.float64)))
        assert abs(mine - n) <= 2e-3 * n + 1e-6 * float(np.max(norms)), (k, mine, n)


def test_itm_triplet_step():
    npz = load('losses.npz')
    c, neg, _ = cases.losses_cases()
    P = {k: T(v).clone().requires_grad_(True) for k, v in c['P'].items()}
    pos = tuple(T(a) for a in c['inputs']); ng = tuple(T(a) for a in neg['inputs'])
    f = lambda inp: O.net_forward('itm', P, c['cfg'], inp, genotype=c['genotype'])
    sp, sc, si = f(pos), f((pos[0], pos[1], pos[2], ng[3], ng[4])), f((ng[0], ng[1], ng[2], pos[3], pos[4]))
    loss = O.itm_bce_loss(sp, sc, si)
    loss.backward()
    assert rel_err(torch.stack([sp, sc, si]).detach().numpy(), npz['itm|scores']) < 1e-4
    assert abs(float(loss.detach()) - float(npz["itm|loss"])) < 1e-4 * float(npz['itm|loss'])
    _gradnorm_check(npz, 'itm|', {k: (p.grad.numpy() if p.grad is not None else None) for k, p in P.items()}, P)
    assert rel_err(P['proj.weight'].grad.numpy(), npz['itm|g:proj.weight']) < 1e-3


def test_vgd_loss():
    npz = load('losses.npz')
    c = cases.losses_cases()[2]
    t = cases.vgd_targets(c, 9204)
    P = {k: T(v).clone().requires_grad_(True) for k, v in c['P'].items()}
    ps, pr = O.net_forward('vgd', P, c['cfg'], tuple(T(a) for a in c['inputs']), genotype=c['genotype'])
    loss, ls, lr = O.vgd_loss(ps, pr, T(t['scores']), T(t['scores_mask']), T(t['bbox']), T(t['bbox_mask']))
    loss.backward()
    assert rel_err(ps.detach().numpy(), npz['vgd|pred_scores']) < 1e-4
    assert rel_err(np.array([float(ls.detach()), float(lr.detach()), float(loss.detach())]), npz['vgd|loss_parts']) < 1e-4
    _gradnorm_check(npz, 'vgd|', {k: (p.grad.numpy() if p.grad is not None else None) for k, p in P.items()}, P)


# ---------------------------------------------------------------------------------------------------------------------
# loader functions (load_data_vqa.py:7-58, 252-296)
# ---------------------------------------------------------------------------------------------------------------------
def test_relation_embedding_pinned():
    npz = load('loader.npz')
    for i in range(4):
        out = O.relation_embedding(T(npz['rel|%d|bbox' % i])).numpy()
        assert out.shape == npz['rel|%d|out' % i].shape
        assert np.allclose(out, npz['rel|%d|out' % i], rtol=1e-6, atol=1e-6)


def test_pad_and_bbox_features_pinned():
    npz = load('loader.npz')
    for i in range(3):
        want = npz['pad|%d|out' % i]
        assert np.array_equal(O.pad_rows(npz['pad|%d|in' % i], want.shape[0]), want)
    for i in range(2):
        out = O.bbox_features(npz['bboxfeat|%d|bbox' % i], tuple(npz['bboxfeat|%d|shape' % i]))
        assert np.array_equal(out, npz['bboxfeat|%d|out' % i])


def test_answer_targets_pinned():
    npz = load('loader.npz')
    a2i = {a: i for i, a in enumerate(cases.LOADER_ANSWERS)}
    for i, answers in enumerate(cases.LOADER_ANSWER_SETS):
        assert np.array_equal(O.answer_target(answers, a2i), npz['ans|%d|out' % i]), i


def test_tokenize_and_semantic_embedding_pinned():
    npz = load('loader.npz')
    tok = {w: i for i, w in enumerate(cases.LOADER_VOCAB)}
    emb = npz['sem|emb']
    for i, q in enumerate(cases.LOADER_QUESTIONS):
        ix, nwords = O.tokenize(q, tok, 14)
        assert np.array_equal(ix, npz['sem|%d|ques_ix' % i])
        size = min(nwords, 14)
        out = O.semantic_embedding(ix, emb, size).numpy()
        assert out.shape == npz['sem|%d|out' % i].shape == (size, size, 3)
        assert np.allclose(out, npz['sem|%d|out' % i], rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------------------------------------------------------------
# the golden recipe runs at HEAD and reproduces the committed files
# ---------------------------------------------------------------------------------------------------------------------
REF = os.environ.get('MMNAS_REFERENCE', '/root/reference')


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'mmnas')), reason='reference tree absent (GPU box)')
def test_golden_recipe_regenerates_bit_exact(tmp_path):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import tests.golden.make_golden as mg\n"
            "mg.HERE = %r\n"
            "for w in ('prims', 'mixed', 'traj', 'train_traj', 'loader', 'losses', 'nets_full'):\n"
            "    getattr(mg, 'gen_' + w)()\n" % (REPO, str(tmp_path)))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, '-c', code], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for w in ('prims', 'mixed', 'traj', 'train_traj', 'loader', 'losses', 'nets_full'):
        new = np.load(os.path.join(str(tmp_path), w + '.npz'))
        old = np.load(os.path.join(GOLDEN, w + '.npz'))
        assert sorted(new.files) == sorted(old.files), w
        for k in new.files:
            assert np.array_equal(new[k], old[k]), (w, k)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'mmnas')) or os.environ.get('MMNAS_REGEN_FULL64') != '1',
                    reason='opt-in (MMNAS_REGEN_FULL64=1, needs the reference tree): the reference at B = 64 on the CPU, two minutes')
def test_full_batch_golden_regenerates_bit_exact(tmp_path):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import tests.golden.make_golden as mg\n"
            "mg.HERE = %r\n"
            "mg.gen_nets_full64()\n"
            "mg.gen_traj64()\n"
            "mg.gen_train_traj64()\n"
            "mg.gen_losses64()\n" % (REPO, str(tmp_path)))
    r = subprocess.run([sys.executable, '-c', code], cwd=str(tmp_path), env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'),
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stderr[-2000:]
    for w in ('nets_full64.npz', 'traj64.npz', 'train_traj64.npz', 'losses64.npz'):
        new = np.load(os.path.join(str(tmp_path), w))
        old = np.load(os.path.join(GOLDEN, w))
        assert sorted(new.files) == sorted(old.files), w
        for k in new.files:
            assert np.array_equal(new[k], old[k]), (w, k)


# ---------------------------------------------------------------------------------------------------------------------
# whole networks at the entry scripts' own dimensions (nets_full.npz)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('spec', cases.FULL_CASES, ids=[cases.full_case_tag(s).rstrip('|') for s in cases.FULL_CASES])
def test_oracle_nets_at_production_dimensions(spec):
    """The CPU restatement against the reference at HSIZE 512 / 256, 100 regions + 14 tokens (ITM 36 + 50), 3129 answers,
    B = 2-4 (configs[0] literally: arch/mcan.json, B = 4, 36 regions): logits, loss, every parameter's gradient norm, strided
    element samples of every gradient (the relation-path ones against the reference's float64 run)."""
    _oracle_full_case(spec, 'nets_full.npz')


def test_oracle_supernet_weight_step_at_the_full_batch():
    """BASELINE configs[2] at its own batch: the supernet weight step at B = 64 (HSIZE 256, 100 regions, 14 tokens, 3129 answers)
    -- the CPU restatement against the reference's own run of exactly that (tests/golden/nets_full64.npz; round 6)."""
    _oracle_full_case(cases.FULL64_CASES[0], 'nets_full64.npz')


def _oracle_full_case(spec, fname):
    from tests import oracle_runner as R
    from tests.util import check_grad_samples
    npz = load(fname)
    kind, task, arch, d, B, Sx, Sy, mode = spec
    tag = cases.full_case_tag(spec)
    c = cases.net_case_full(spec, int(npz[tag + 'seed']))
    assert abs(cases.checksum(dict(c['P'], frcn=c['inputs'][0], yrel=c['inputs'][2], q=c['inputs'][3], xrel=c['inputs'][4]))
               - float(npz[tag + 'insum'])) <= 1e-9 * abs(float(npz[tag + 'insum']))
    search = None
    if kind == 'search':
        search = c['plan']
        flat = search['enc'] + search['dec']
        assert [a[0] for a, _ in flat] == list(npz[tag + 'plan_act'])
        gates = [k for k in c['P'] if k.endswith('alpha_gate')]
        for k, (act, _) in zip(gates, flat):
            c['P'][k][:] = 0
            c['P'][k][act[0]] = 1.0
    pred, loss, grads = R.run_oracle_net(c, search=search)
    if task == 'vgd':
        assert rel_err(pred[0].detach().numpy(), npz[tag + 'scores']) < 1e-4
        assert rel_err(pred[1].detach().numpy(), npz[tag + 'reg']) < 1e-4
    else:
        assert rel_err(pred.detach().numpy(), npz[tag + 'pred']) < 1e-4
    assert abs(loss - float(npz[tag + 'loss'])) < 1e-4 * abs(float(npz[tag + 'loss']))
    keys = [str(k) for k in npz[tag + 'gradnorm_keys']]
    assert set(keys) == set(c['P'].keys())
    top = float(np.max(npz[tag + 'gradnorms']))
    for k, n in zip(keys, npz[tag + 'gradnorms']):
        if 'alpha' in k:
            continue
        mine = 0.0 if grads[k] is None else float(np.linalg.norm(grads[k].astype(np.float64)))
        assert abs(mine - n) <= 2e-3 * n + 1e-6 * top, (k, mine, n)
    assert check_grad_samples(npz, tag, grads, skip=lambda k: 'alpha' in k) > 100
    if mode is not None:
        gg = np.stack([np.pad(grads[k], (0, 4 - grads[k].size)) for k in gates])
        assert rel_err(gg, npz[tag + 'gate_grads']) < 1e-3

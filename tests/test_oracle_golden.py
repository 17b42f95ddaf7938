"""Pin the CPU oracle against the golden vectors produced by the imported reference
(tests/golden/make_golden.py).  Runs without a GPU."""
import numpy as np
import pytest
import torch

from oracle import mmnas_oracle as O
from tests import oracle_runner as R
from tests.golden import cases
from tests.util import check_grad_samples, TOL, golden_err, has, load, rel_err

T = torch.from_numpy
OTOL = 2e-5  # oracle vs reference: same fp32 arithmetic, only summation order differs


def _check_case(npz, tag, case, res, tol=OTOL):
    insum = cases.checksum(dict(x=case['x'], y=case['y'], rel=case['rel'], **case['P']))
    assert abs(insum - float(npz[tag + '|insum'])) <= 1e-6 * max(1.0, abs(insum)), 'input generator drifted'
    checked = 0
    for k, v in res.items():
        key = tag + '|' + k
        if not has(npz, key):
            # the reference leaves grads of unused inputs as None; ours must then be zero
            assert not np.any(v), key
            continue
        e = golden_err(npz, key, v)
        assert e <= tol, (key, e)
        checked += 1
    assert checked >= 2


@pytest.mark.parametrize('name', O.ALL_OP_NAMES)
@pytest.mark.parametrize('nr', [(True, True), (False, False)])
def test_every_registry_op(name, nr):
    npz = load('ops.npz')
    tag = '%s|%d%d' % (name, int(nr[0]), int(nr[1]))
    case = cases.op_case(name, nr[0], nr[1], int(npz[tag + '|seed']))
    # rel-bias gradients go through 1/r with r arbitrarily close to the clamp; fp32 re-association
    # of the 64-term dot moves those entries by ~1e-4 relative
    tol = 5e-4 if 'rel_' in name else OTOL
    _check_case(npz, tag, case, R.run_oracle_op(case), tol)


def _shape_tags():
    npz = load('ops_shapes.npz')
    return sorted({k.rsplit('|', 1)[0] for k in npz.files if k.endswith('|seed')})


@pytest.mark.parametrize('tag', _shape_tags())
def test_shape_variants(tag):
    npz = load('ops_shapes.npz')
    name, dims = tag.split('|')
    B, Sx, Sy, d = (int(v) for v in dims.split('_'))
    case = cases.op_case(name, True, True, int(npz[tag + '|seed']), dict(B=B, Sx=Sx, Sy=Sy, HSIZE=d))
    tol = 5e-4 if 'rel_' in name else 5e-5
    _check_case(npz, tag, case, R.run_oracle_op(case), tol)


def test_layernorm_forward_and_closed_form_backward():
    npz = load('prims.npz')
    for d in (128, 256, 1024):
        x, a, b, g = (T(npz['ln%d|%s' % (d, k)]) for k in 'xabg')
        y = O.layer_norm(x, a, b)
        assert rel_err(y.numpy(), npz['ln%d|y' % d]) < OTOL
        dx, da, db = O.layer_norm_backward(x, a, g)
        assert rel_err(dx.numpy(), npz['ln%d|dx' % d]) < 1e-4
        assert rel_err(da.numpy(), npz['ln%d|da' % d]) < 1e-4
        assert rel_err(db.numpy(), npz['ln%d|db' % d]) < 1e-4


def test_attflat_mask_lstm():
    npz = load('prims.npz')
    for G in (1, 2):
        pre = 'af%d|' % G
        P = {k[len(pre) + 2:]: T(npz[k]).requires_grad_(True) for k in npz.files if k.startswith(pre + 'P:')}
        x = T(npz[pre + 'x']).requires_grad_(True)
        y = O.att_flat(P, x, T(npz[pre + 'mask']), G)
        assert rel_err(y.detach().numpy(), npz[pre + 'y']) < OTOL
        (y * T(npz[pre + 'g'])).sum().backward()
        assert rel_err(x.grad.numpy(), npz[pre + 'dx']) < 1e-4
        for k, p in P.items():
            assert rel_err(p.grad.numpy(), npz[pre + 'g:' + k]) < 1e-4, k
    assert np.array_equal(O.make_mask(T(npz['mask|f'])).numpy(), npz['mask|m'])
    P = {k[len('lstm|P:'):]: T(npz[k]) for k in npz.files if k.startswith('lstm|P:')}
    y = O.lstm_forward(P, '', T(npz['lstm|x']))
    assert rel_err(y.numpy(), npz['lstm|y']) < OTOL


@pytest.mark.parametrize('mode', [None, 'full', 'two'])
@pytest.mark.parametrize('kind', ['enc_safe', 'dec_safe'])
def test_mixed_op(mode, kind):
    npz = load('mixed.npz')
    tag = 'mx|%s|%s|' % (mode, kind)
    c = cases.mixed_case(mode, kind, int(npz[tag + 'seed']))
    insum = cases.checksum(dict(c['P'], s=c['s'], pre=c['pre'], rel=c['rel']))
    assert abs(insum - float(npz[tag + 'insum'])) <= 1e-6 * abs(insum)
    P = {k: T(v).requires_grad_(True) for k, v in c['P'].items()}
    s = T(c['s']).requires_grad_(True)
    names = O.USED_OPS[kind]
    involved = c['act'] + (c['inact'] if mode else [])
    outs = [None] * len(names)
    for i in involved:
        sub = {k[len('candidate_ops.%d.' % i):]: v for k, v in P.items() if k.startswith('candidate_ops.%d.' % i)}
        outs[i] = O.op_forward(names[i], sub, c['cfg'], s, T(c['pre']), T(c['sm']), T(c['pm']), T(c['rel']))
    o = O.mixed_forward(mode, outs, P['alpha_gate'], c['act'], c['inact'])
    assert rel_err(o.detach().numpy(), npz[tag + 'out']) < OTOL
    (o * T(c['g'])).sum().backward()
    assert rel_err(s.grad.numpy(), npz[tag + 'ds']) < 1e-4
    if mode is None:
        return
    gg = P['alpha_gate'].grad
    assert rel_err(gg.numpy(), npz[tag + 'gate_grad']) < 1e-4
    if mode == 'full':
        pg = O.alpha_prob_grad_full(P['alpha_prob'].detach(), gg)
    else:
        pg = O.alpha_prob_grad_two(P['alpha_prob'].detach(), gg, c['act'] + c['inact'])
        rs = O.rescale_two(T(npz[tag + 'alpha_stepped']), [float(npz[tag + 'alpha_old'][i]) for i in c['act'] + c['inact']],
                           c['act'] + c['inact'])
        assert rel_err(rs.numpy(), npz[tag + 'alpha_rescaled']) < 1e-5
    assert rel_err(pg.numpy(), npz[tag + 'prob_grad']) < 1e-4


def test_alpha_algebra_vectors():
    npz = load('mixed.npz')
    for n in (2, 4, 5):
        a = T(npz['alg|%d|alpha' % n]); g = T(npz['alg|%d|gate_grad' % n])
        assert rel_err(O.alpha_prob_grad_full(a, g).numpy(), npz['alg|%d|prob_grad' % n]) < 1e-5
        assert rel_err(torch.softmax(a, 0).numpy(), npz['alg|%d|probs' % n]) < 1e-6
        assert int(torch.argmax(a)) == int(npz['alg|%d|chosen' % n])


@pytest.mark.parametrize('task,arch', [('vqa', 'mcan'), ('vqa', 'mmnas_vqa'), ('vgd', 'mmnas_vgd'), ('itm', 'mmnas_itm')])
def test_net_full(task, arch):
    npz = load('nets.npz')
    tag = 'full|%s|%s|' % (task, arch)
    c = cases.net_case(task, arch, int(npz[tag + 'seed']))
    pred, loss, grads = R.run_oracle_net(c)
    if task == 'vgd':
        assert rel_err(pred[0].detach().numpy(), npz[tag + 'scores']) < 1e-4
        assert rel_err(pred[1].detach().numpy(), npz[tag + 'reg']) < 1e-4
    else:
        assert rel_err(pred.detach().numpy(), npz[tag + 'pred']) < 1e-4
    assert abs(loss - float(npz[tag + 'loss'])) < 1e-4 * abs(float(npz[tag + 'loss']))
    keys = [str(k) for k in npz[tag + 'gradnorm_keys']]
    norms = npz[tag + 'gradnorms']
    assert set(keys) == set(c['P'].keys())  # state_dict key compatibility
    for k, n in zip(keys, norms):
        mine = 0.0 if grads[k] is None else float(np.linalg.norm(grads[k].astype(np.float64)))
        assert abs(mine - n) <= 2e-3 * n + 1e-6 * float(np.max(npz[tag + 'gradnorms'])), (k, mine, n)
    assert check_grad_samples(npz, tag, grads) > 100      # every gradient tensor, element-wise on strided samples


@pytest.mark.parametrize('task,mode', [('vqa', None), ('vqa', 'full'), ('vqa', 'two'), ('vgd', None),
                                       ('vgd', 'full'), ('itm', None), ('itm', 'full')])
def test_net_search_steps(task, mode):
    npz = load('nets.npz')
    tag = 'search|%s|%s|' % (task, mode)
    seed = int(npz[tag + 'seed'])
    c = cases.net_case(task, None, seed, search=True)
    plan = cases.search_plan(np.random.RandomState(seed + 50000), mode)
    flat = plan['enc'] + plan['dec']
    assert [a[0] for a, _ in flat] == list(npz[tag + 'plan_act'])
    # set the gates as binarize() would (mixed.py:133,144/158)
    keys = [k for k in c['P'] if k.endswith('alpha_gate')]
    assert len(keys) == 30
    for k, (act, _) in zip(keys, flat):
        c['P'][k][:] = 0
        c['P'][k][act[0]] = 1.0
    pred, loss, grads = R.run_oracle_net(c, search=plan)
    if task == 'vgd':
        assert rel_err(pred[0].detach().numpy(), npz[tag + 'scores']) < 1e-4
    else:
        assert rel_err(pred.detach().numpy(), npz[tag + 'pred']) < 1e-4
    assert abs(loss - float(npz[tag + 'loss'])) < 1e-4 * abs(float(npz[tag + 'loss']))
    gkeys = [str(k) for k in npz[tag + 'gradnorm_keys']]
    assert set(gkeys) == set(c['P'].keys())
    for k, n in zip(gkeys, npz[tag + 'gradnorms']):
        if 'alpha' in k:
            continue
        mine = 0.0 if grads[k] is None else float(np.linalg.norm(grads[k].astype(np.float64)))
        assert abs(mine - n) <= 2e-3 * n + 1e-6 * float(np.max(npz[tag + 'gradnorms'])), (k, mine, n)
    assert check_grad_samples(npz, tag, grads, skip=lambda k: 'alpha' in k) > 100
    if mode is not None:
        gg = np.stack([np.pad(grads[k], (0, 4 - grads[k].size)) for k in keys])
        assert rel_err(gg, npz[tag + 'gate_grads']) < 1e-3
        pkeys = [k.replace('alpha_gate', 'alpha_prob') for k in keys]
        pgs = []
        for k, pk, (act, inact) in zip(keys, pkeys, flat):
            a = T(c['P'][pk]); g = T(grads[k])
            pg = O.alpha_prob_grad_full(a, g) if mode == 'full' else O.alpha_prob_grad_two(a, g, act + inact)
            pgs.append(np.pad(pg.numpy(), (0, 4 - pg.numel())))
        assert rel_err(np.stack(pgs), npz[tag + 'prob_grads']) < 1e-3


def test_genotype_and_init_prior():
    npz = load('nets.npz')
    seed = int(npz['search|vqa|None|seed'])
    c = cases.net_case('vqa', None, seed, search=True)
    pk = [k for k in c['P'] if k.endswith('alpha_prob')]
    enc = [T(c['P'][k]) for k in pk if 'enc' in k]
    dec = [T(c['P'][k]) for k in pk if 'dec' in k]
    g = O.genotype_from_alphas(enc, dec)
    assert [n[0] for n in g['enc']] == [str(s) for s in npz['search|vqa|genotype_enc']]
    assert [n[0] for n in g['dec']] == [str(s) for s in npz['search|vqa|genotype_dec']]
    w = np.stack([torch.softmax(a, -1).numpy() for a in dec])
    assert rel_err(w, npz['search|vqa|w_dec']) < 1e-6
    ia = npz['search|vqa|init_alpha']
    # hygr_vqa.py:142-156: +1 on the MCAN-style prior op, -1 elsewhere
    assert ia.shape == (30, 4)
    assert np.all(np.abs(ia[:12, :2]) == 1) and np.all(ia[:12].sum(1) == 0)
    assert np.all(ia[12:].sum(1) == -2)

"""Generate the golden vectors under tests/golden/ by running the *imported reference*
(/root/reference, available in the build container only) on the deterministic cases of
cases.py.  Only plain arrays (outputs, gradients, checksums) are written -- no reference
source, bytecode or pickled module travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Files written (npz, float32 unless noted):
    ops.npz          every registry operator x {(norm,res)=(T,T),(F,F)}: out, dx, dy, drel, param grads
    ops_shapes.npz   head-dim / production-shape spot checks (out + input grads + grad checksums)
    prims.npz        LayerNorm, AttFlat, make_mask, LSTM stand-alone
    mixed.npz        MixedOp algebra: forward modes, alpha-gradient, rescale, genotype
    nets.npz         Net_Full(arch/*.json) for vqa/vgd/itm and Net_Search weight/arch steps
"""
import os
import sys

os.environ.setdefault('PYTHONDONTWRITEBYTECODE', '1')
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = os.environ.get('MMNAS_REFERENCE', '/root/reference')

import numpy as np
import torch

from tests.golden import cases
from oracle import mmnas_oracle as O

torch.set_num_threads(4)


def _import_reference():
    # the repo ships an alias package also called `mmnas`; make sure the reference wins here
    for k in [k for k in sys.modules if k == 'mmnas' or k.startswith('mmnas.')]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    import mmnas.model.modules as rm
    import mmnas.model.mixed as rmix
    import mmnas.utils.ops_adapter as roa
    assert rm.__file__.startswith(REF), rm.__file__
    return rm, rmix, roa


RM, RMIX, ROA = _import_reference()
from mmnas.model import hygr_vqa, hygr_vgd, hygr_itm, full_vqa, full_vgd, full_itm  # noqa: E402

T = torch.from_numpy


def load_state(mod, P):
    sd = mod.state_dict()
    assert set(sd.keys()) == set(P.keys()), (sorted(set(sd) ^ set(P)))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), (k, sd[k].shape, P[k].shape)
    mod.load_state_dict({k: T(v) for k, v in P.items()})


def summarize(out, key, g):
    """Store a gradient fully when small, else a strided sample + L2 norm + weighted checksum."""
    g = np.ascontiguousarray(g, dtype=np.float32)
    io = key.split('|')[-1] in ('out', 'dx', 'dy')
    if g.size <= 4096 or (io and g.size <= 30000):
        out[key] = g
    else:
        flat = g.reshape(-1)
        out[key + '#sample'] = flat[::(7 if io else 53)].copy()
        out[key + '#norm'] = np.float64(np.sqrt(np.sum(flat.astype(np.float64) ** 2)))
        out[key + '#shape'] = np.array(g.shape, np.int64)


def run_ref_op(case):
    name, cfg = case['name'], case['cfg']
    op = ROA.OpsAdapter().OPS[name](cfg, norm=cfg.OPS_NORM, residual=cfg.OPS_RESIDUAL)
    op.train()  # DROPOUT_R = 0 -> deterministic
    if case['P']:
        load_state(op, case['P'])
    x = T(case['x']).requires_grad_(True)
    y = T(case['y']).requires_grad_(True)
    rel = T(case['rel']).requires_grad_(True)
    if case['kind'] in ('relu', 'leakyrelu'):
        out = op(x)  # plain nn modules take one positional argument (SURVEY 8a note)
    else:
        out = op(x, y, T(case['x_mask']), T(case['y_mask']), rel)
    loss = (out * T(case['gout'])).sum()
    loss.backward()
    res = {'out': out.detach().numpy()}
    res['dx'] = x.grad.numpy() if x.grad is not None else np.zeros_like(case['x'])
    if y.grad is not None:
        res['dy'] = y.grad.numpy()
    if rel.grad is not None:
        res['drel'] = rel.grad.numpy()
    for k, p in op.named_parameters():
        res['g:' + k] = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    return res


def gen_ops():
    out = {}
    seed = 1000
    for name in O.ALL_OP_NAMES:
        for (norm, res) in ((True, True), (False, False)):
            seed += 1
            case = cases.op_case(name, norm, res, seed)
            r = run_ref_op(case)
            tag = '%s|%d%d' % (name, int(norm), int(res))
            out[tag + '|seed'] = np.int64(seed)
            out[tag + '|insum'] = np.float64(cases.checksum(
                dict(x=case['x'], y=case['y'], rel=case['rel'], **case['P'])))
            for k, v in r.items():
                summarize(out, tag + '|' + k, v)
    np.savez_compressed(os.path.join(HERE, 'ops.npz'), **out)
    print('ops.npz', len(out), 'arrays')


def gen_ops_shapes():
    """head-dim variants at HSIZE=256 and production-shape spot checks (B=2)."""
    out = {}
    specs = []
    for nm in ('self_att_16', 'self_att_32', 'self_att_128', 'self_att_64_2', 'guided_att_64_2',
               'rel_self_att_16', 'rel_self_att_128', 'uniimg_att_32'):
        specs.append((nm, dict(B=2, Sx=9, Sy=6, HSIZE=256)))
    for nm in ('self_att_64', 'rel_self_att_64', 'guided_att_64', 'feed_forward'):
        specs.append((nm, dict(B=2, Sx=100, Sy=14, HSIZE=512)))   # VQA/VGD decoder shapes
        specs.append((nm, dict(B=2, Sx=36, Sy=50, HSIZE=512)))    # ITM decoder shapes
    specs.append(('self_att_64', dict(B=2, Sx=14, Sy=100, HSIZE=512)))  # encoder shape
    specs.append(('self_att_64', dict(B=2, Sx=130, Sy=4, HSIZE=128)))   # > 128 keys (two key blocks)
    seed = 5000
    for nm, dims in specs:
        seed += 1
        case = cases.op_case(nm, True, True, seed, dims)
        r = run_ref_op(case)
        tag = '%s|%d_%d_%d_%d' % (nm, dims['B'], dims['Sx'], dims['Sy'], dims['HSIZE'])
        out[tag + '|seed'] = np.int64(seed)
        out[tag + '|insum'] = np.float64(cases.checksum(
            dict(x=case['x'], y=case['y'], rel=case['rel'], **case['P'])))
        for k, v in r.items():
            summarize(out, tag + '|' + k, v)
    np.savez_compressed(os.path.join(HERE, 'ops_shapes.npz'), **out)
    print('ops_shapes.npz', len(out), 'arrays')


def gen_prims():
    out = {}
    rs = np.random.RandomState(77)
    # LayerNorm (modules.py:44-56)
    for d in (128, 256, 1024):
        x = rs.standard_normal((3, 5, d)).astype(np.float32) * 2 + 0.5
        a = (1 + 0.2 * rs.standard_normal(d)).astype(np.float32)
        b = (0.1 * rs.standard_normal(d)).astype(np.float32)
        g = rs.standard_normal((3, 5, d)).astype(np.float32)
        ln = RM.LayerNorm(d)
        ln.load_state_dict({'a_2': T(a), 'b_2': T(b)})
        xt = T(x).requires_grad_(True)
        y = ln(xt)
        (y * T(g)).sum().backward()
        out['ln%d|x' % d] = x; out['ln%d|a' % d] = a; out['ln%d|b' % d] = b; out['ln%d|g' % d] = g
        out['ln%d|y' % d] = y.detach().numpy(); out['ln%d|dx' % d] = xt.grad.numpy()
        out['ln%d|da' % d] = ln.a_2.grad.numpy(); out['ln%d|db' % d] = ln.b_2.grad.numpy()
    # AttFlat (modules.py:59-85), glimpses 1 and 2
    for G in (1, 2):
        cfg = cases.small_cfg(HSIZE=128, ATTFLAT_GLIMPSES=G)
        af = RM.AttFlat(cfg)
        shapes = {k: tuple(v.shape) for k, v in af.state_dict().items()}
        P = cases.rand_params(shapes, rs)
        load_state(af, P)
        x = rs.standard_normal((3, 7, 128)).astype(np.float32)
        m = cases.masks(rs, 3, 7)
        xt = T(x).requires_grad_(True)
        y = af(xt, T(m))
        g = rs.standard_normal(tuple(y.shape)).astype(np.float32)
        (y * T(g)).sum().backward()
        pre = 'af%d|' % G
        out[pre + 'x'] = x; out[pre + 'mask'] = m; out[pre + 'g'] = g
        for k, v in P.items():
            out[pre + 'P:' + k] = v
        out[pre + 'y'] = y.detach().numpy(); out[pre + 'dx'] = xt.grad.numpy()
        for k, p in af.named_parameters():
            out[pre + 'g:' + k] = p.grad.numpy()
    # make_mask (hygr_vqa.py:121-122) through a Net method needs a net; restate call on tensors
    f = rs.standard_normal((3, 6, 8)).astype(np.float32)
    f[1, 4:] = 0; f[2] = 0
    mm = (torch.sum(torch.abs(T(f)), dim=-1) == 0).unsqueeze(1).unsqueeze(2)
    out['mask|f'] = f; out['mask|m'] = mm.numpy()
    # LSTM stem (hygr_vqa.py:64-69)
    lstm = torch.nn.LSTM(input_size=24, hidden_size=32, num_layers=1, batch_first=True)
    shapes = {k: tuple(v.shape) for k, v in lstm.state_dict().items()}
    P = cases.rand_params(shapes, rs)
    lstm.load_state_dict({k: T(v) for k, v in P.items()})
    x = rs.standard_normal((3, 5, 24)).astype(np.float32)
    y, _ = lstm(T(x))
    out['lstm|x'] = x; out['lstm|y'] = y.detach().numpy()
    for k, v in P.items():
        out['lstm|P:' + k] = v
    np.savez_compressed(os.path.join(HERE, 'prims.npz'), **out)
    print('prims.npz', len(out), 'arrays')


def gen_mixed():
    """MixedOp algebra with injected indices (mixed.py:59-208)."""
    out = {}
    rs = np.random.RandomState(4242)
    cfg = cases.small_cfg(HSIZE=128)
    MixedOp = RMIX.MixedOp
    seed = 7000
    for mode in (None, 'full', 'two'):
        for kind in ('enc_safe', 'dec_safe'):
            seed += 1
            tag = 'mx|%s|%s|' % (mode, kind)
            c = cases.mixed_case(mode, kind, seed)
            m = MixedOp(c['cfg'], kind)
            m.train()
            load_state(m, c['P'])
            m.active_index, m.inactive_index = list(c['act']), list(c['inact'])
            MixedOp.MODE = mode
            st = T(c['s']).requires_grad_(True)
            o = m(st, T(c['pre']), T(c['sm']), T(c['pm']), T(c['rel']))
            (o * T(c['g'])).sum().backward()
            out[tag + 'seed'] = np.int64(seed)
            out[tag + 'insum'] = np.float64(cases.checksum(dict(c['P'], s=c['s'], pre=c['pre'], rel=c['rel'])))
            out[tag + 'out'] = o.detach().numpy(); out[tag + 'ds'] = st.grad.numpy()
            if mode is not None:
                out[tag + 'gate_grad'] = m.alpha_gate.grad.numpy().copy()
                m.alpha_prob.grad = None
                m.set_arch_param_grad()
                out[tag + 'prob_grad'] = m.alpha_prob.grad.numpy().copy()
                if mode == 'two':
                    old = m.alpha_prob.data.clone()
                    m.alpha_prob.data -= 0.1 * m.alpha_prob.grad  # a stand-in optimizer step
                    out[tag + 'alpha_stepped'] = m.alpha_prob.data.numpy().copy()
                    m.rescale_updated_arch_param()
                    out[tag + 'alpha_rescaled'] = m.alpha_prob.data.numpy().copy()
                    out[tag + 'alpha_old'] = old.numpy()
            MixedOp.MODE = None
    # pure alpha algebra on random vectors (no operator evaluation), incl. chosen_index / probs
    for n in (2, 4, 5):
        a = rs.standard_normal(n).astype(np.float32)
        gg = rs.standard_normal(n).astype(np.float32)
        m = MixedOp(cfg, 'enc_safe' if n == 2 else ('dec_safe' if n == 4 else 'dec'))
        m.alpha_prob.data.copy_(T(a))
        m.alpha_gate.grad = T(gg.copy())
        MixedOp.MODE = 'full'
        m.active_index, m.inactive_index = [0], list(range(1, n))
        m.set_arch_param_grad()
        out['alg|%d|alpha' % n] = a; out['alg|%d|gate_grad' % n] = gg
        out['alg|%d|prob_grad' % n] = m.alpha_prob.grad.numpy().copy()
        out['alg|%d|probs' % n] = m.probs_over_ops.detach().numpy()
        out['alg|%d|chosen' % n] = np.int64(m.chosen_index[0])
        MixedOp.MODE = None
    np.savez_compressed(os.path.join(HERE, 'mixed.npz'), **out)
    print('mixed.npz', len(out), 'arrays')


def _net_loss(task, pred, target):
    if task == 'vqa':
        return torch.nn.functional.binary_cross_entropy_with_logits(pred, T(target), reduction='sum')
    if task == 'itm':
        return torch.nn.functional.binary_cross_entropy(pred, T(target), reduction='sum')
    scores, reg = pred
    return (scores * T(target)).sum() + 0.5 * (reg ** 2).sum()


def gen_nets():
    out = {}
    full = {'vqa': full_vqa.Net_Full, 'vgd': full_vgd.Net_Full, 'itm': full_itm.Net_Full}
    hygr = {'vqa': hygr_vqa.Net_Search, 'vgd': hygr_vgd.Net_Search, 'itm': hygr_itm.Net_Search}
    seed = 9000
    for task, arch in (('vqa', 'mcan'), ('vqa', 'mmnas_vqa'), ('vgd', 'mmnas_vgd'), ('itm', 'mmnas_itm')):
        seed += 1
        c = cases.net_case(task, arch, seed)
        init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
                'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
        net = full[task](c['cfg'], init)
        net.train()
        load_state(net, c['P'])
        inp = tuple(T(a) for a in c['inputs'])
        pred = net(inp)
        loss = _net_loss(task, pred, c['target'])
        loss.backward()
        tag = 'full|%s|%s|' % (task, arch)
        out[tag + 'seed'] = np.int64(seed)
        out[tag + 'insum'] = np.float64(cases.checksum(dict(c['P'], frcn=c['inputs'][0], yrel=c['inputs'][2],
                                                           q=c['inputs'][3], xrel=c['inputs'][4])))
        if task == 'vgd':
            out[tag + 'scores'] = pred[0].detach().numpy(); out[tag + 'reg'] = pred[1].detach().numpy()
        else:
            out[tag + 'pred'] = pred.detach().numpy()
        out[tag + 'loss'] = np.float64(loss.item())
        gn = {}
        for k, p in net.named_parameters():
            gn[k] = 0.0 if p.grad is None else float(p.grad.double().norm())
        keys = sorted(gn)
        out[tag + 'gradnorm_keys'] = np.array(keys)
        out[tag + 'gradnorms'] = np.array([gn[k] for k in keys], np.float64)
        out[tag + 'g:imgfeat_linear.bias'] = net.imgfeat_linear.bias.grad.numpy()
        if net.linear_y_rel.weight.grad is not None:
            out[tag + 'g:linear_y_rel.weight'] = net.linear_y_rel.weight.grad.numpy()

    # supernet: weight step (MODE None) and arch steps ('full', 'two') with injected samples
    MixedOp = RMIX.MixedOp
    for task in ('vqa', 'vgd', 'itm'):
        for mode in (None, 'full', 'two'):
            if task != 'vqa' and mode == 'two':
                continue
            seed += 1
            c = cases.net_case(task, None, seed, search=True)
            rs = np.random.RandomState(seed + 50000)
            plan = cases.search_plan(rs, mode)
            init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
                    'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
            net = hygr[task](c['cfg'], init)
            net.train()
            load_state(net, c['P'])
            mops = net.redundant_modules
            assert len(mops) == 30
            flat = plan['enc'] + plan['dec']
            MixedOp.MODE = mode
            for m, (act, inact) in zip(mops, flat):
                m.alpha_gate.data.zero_()
                m.alpha_gate.data[act[0]] = 1.0
                m.active_index, m.inactive_index = list(act), list(inact)
            net.unused_modules_off()
            inp = tuple(T(a) for a in c['inputs'])
            pred = net(inp)
            loss = _net_loss(task, pred, c['target'])
            net.zero_grad()
            loss.backward()
            tag = 'search|%s|%s|' % (task, mode)
            out[tag + 'seed'] = np.int64(seed)
            out[tag + 'plan_act'] = np.array([a[0] for a, _ in flat], np.int64)
            out[tag + 'plan_inact'] = np.array([(list(i) + [-1] * 3)[:3] for _, i in flat], np.int64)
            if task == 'vgd':
                out[tag + 'scores'] = pred[0].detach().numpy(); out[tag + 'reg'] = pred[1].detach().numpy()
            else:
                out[tag + 'pred'] = pred.detach().numpy()
            out[tag + 'loss'] = np.float64(loss.item())
            if mode is not None:
                out[tag + 'gate_grads'] = np.stack(
                    [np.pad(m.alpha_gate.grad.numpy(), (0, 4 - m.n_choices)) for m in mops])
                net.set_arch_param_grad()
                out[tag + 'prob_grads'] = np.stack(
                    [np.pad(m.alpha_prob.grad.numpy(), (0, 4 - m.n_choices)) for m in mops])
            net.unused_modules_back()
            gn = {}
            for k, p in net.named_parameters():
                gn[k] = 0.0 if p.grad is None else float(p.grad.double().norm())
            keys = sorted(gn)
            out[tag + 'gradnorm_keys'] = np.array(keys)
            out[tag + 'gradnorms'] = np.array([gn[k] for k in keys], np.float64)
            MixedOp.MODE = None
            if task == 'vqa' and mode is None:
                # genotype / genotype_weights for the loaded alphas (hygr_vqa.py:242-297)
                g = net.genotype()
                out['search|vqa|genotype_enc'] = np.array([n[0] for n in g['enc']])
                out['search|vqa|genotype_dec'] = np.array([n[0] for n in g['dec']])
                gw = net.genotype_weights()
                out['search|vqa|w_enc'] = np.stack(gw['w_enc']); out['search|vqa|w_dec'] = np.stack(gw['w_dec'])
    # init_arch prior (hygr_vqa.py:124-156): alpha_prob after construction
    cfg = cases.small_cfg(HSIZE=128)
    init = {'token_size': 40, 'ans_size': 13, 'pretrained_emb': np.zeros((40, cfg.WORD_EMBED_SIZE), np.float32)}
    net = hygr_vqa.Net_Search(cfg, init)
    out['search|vqa|init_alpha'] = np.stack(
        [np.pad(p.detach().numpy(), (0, 4 - p.numel())) for p in net.alpha_prob_parameters()])
    np.savez_compressed(os.path.join(HERE, 'nets.npz'), **out)
    print('nets.npz', len(out), 'arrays')


if __name__ == '__main__':
    which = sys.argv[1:] or ['ops', 'ops_shapes', 'prims', 'mixed', 'nets']
    for w in which:
        globals()['gen_' + w]()

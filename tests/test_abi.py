"""CPU-side checks of the C-ABI boundary and the host logic (no GPU, no compute calls):
the shared library loads and exports every symbol include/mmnas_hip.h declares, the ctypes
binding covers exactly those symbols, the product path refuses CPU tensors, and the MixedOp /
supernet bookkeeping (sampling, alpha-gradient algebra, genotype) matches the reference goldens."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.golden import cases
from tests.util import REPO, load, rel_err

T = torch.from_numpy


def _header_functions():
    src = open(os.path.join(REPO, 'include', 'mmnas_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mmnas_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from mmnas_amd import _lib as L
    names = _header_functions()
    assert len(names) >= 25
    assert os.path.exists(L.LIB_PATH), 'run __graft_entry__.build() first'
    raw = ctypes.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), 'libmmnas_hip.so does not export %s' % n
    assert sorted(L.SYMBOLS.keys()) == names  # binding and header agree
    assert L.lib().mmnas_abi_version() == 1


def test_struct_sizes_are_plain_c_layouts():
    from mmnas_amd import _lib as L
    # spot-check the layouts the C side assumes (LP64): catches a field added on one side only
    assert ctypes.sizeof(L.Plan) == 24
    assert ctypes.sizeof(L.Segment) == 24
    assert ctypes.sizeof(L.GemmGroup) == 8 + 8 * 12
    assert ctypes.sizeof(L.GemmDesc) % 8 == 0 and ctypes.sizeof(L.AttOp) % 8 == 0


def test_plan_functions_run_on_host():
    from mmnas_amd import _lib as L
    op = L.AttOp()
    op.B, op.Sq, op.Sk, op.d, op.di, op.H, op.dh, op.R = 64, 100, 100, 512, 512, 8, 64, 64
    op.flags = L.F_NORM | L.F_RESIDUAL | L.F_MASK | L.F_REL | L.F_SELF | L.F_TRAIN
    op.drop_p = 0.1
    p = L.Plan()
    L.check(L.lib().mmnas_att_op_plan(ctypes.byref(op), ctypes.byref(p)))
    M = 6400
    assert p.save_bytes >= (4 * M * 512 + M * 512 + 64 * 8 * 100 * 100 + 64 * 8 * 100 * 2) * 4
    assert p.ws_bwd_bytes >= (6 * M * 512) * 4
    op.H = 7  # inconsistent head split -> shape error with a message
    rc = L.lib().mmnas_att_op_plan(ctypes.byref(op), ctypes.byref(p))
    assert rc == -1 and b'H*dh' in L.lib().mmnas_last_error()
    m = L.MlpOp()
    m.M, m.nl = 6400, 2
    m.dims[0], m.dims[1], m.dims[2] = 512, 2048, 512
    m.flags = L.F_NORM | L.F_RESIDUAL
    L.check(L.lib().mmnas_mlp_op_plan(ctypes.byref(m), ctypes.byref(p)))
    assert p.save_bytes >= (6400 * 2048 + 6400 * 512) * 4


def test_host_only_entry_points_of_the_product_library():
    """Chain / mixed-chain / head planners, scratch-size helpers and the argument checks with their messages
    (tests/host_plan_driver.py; tests/test_asan.py runs the same driver against the AddressSanitizer build)."""
    from tests import host_plan_driver
    out = host_plan_driver.run()
    assert out['mixed_chain_bytes'] > out['chain_bytes'] > 0 and out['head_bytes'] > 0


def test_product_path_refuses_cpu_tensors():
    from mmnas_amd import _lib as L
    from mmnas.utils.ops_adapter import OpsAdapter
    cfg = cases.small_cfg()
    ad = OpsAdapter()
    assert len(ad.OPS) == 41
    assert ad.Used_OPS['dec'] == ['self_att_64', 'rel_self_att_64', 'guided_att_64', 'feed_forward', 'none']
    x = torch.zeros(2, 3, cfg.HSIZE)
    for name in ('feed_forward', 'self_att_64', 'none', 'gelu', 'sep_conv_3'):
        with pytest.raises(L.MMNasHipError):
            ad.OPS[name](cfg, True, True)(x, x, None, None, None)


def test_registry_state_dict_keys_match_reference():
    """Parameter names/shapes of every registry operator equal the reference's (via the oracle's table,
    itself asserted against the reference's state_dict() in make_golden.load_state)."""
    from mmnas.utils.ops_adapter import OpsAdapter
    from oracle import mmnas_oracle as O
    for name in O.ALL_OP_NAMES:
        d = 256 if '256' in name else 128
        cfg = cases.small_cfg(HSIZE=d)
        op = OpsAdapter().OPS[name](cfg, True, True)
        sd = {k: tuple(v.shape) for k, v in op.state_dict().items()}
        assert sd == {k: tuple(v) for k, v in O.op_param_shapes(name, cfg).items()}, name


def test_mixed_op_alpha_algebra_on_host():
    from mmnas.model.mixed import MixedOp
    npz = load('mixed.npz')
    cfg = cases.small_cfg()
    for n, kind in ((2, 'enc_safe'), (4, 'dec_safe'), (5, 'dec')):
        m = MixedOp(cfg, kind)
        assert m.n_choices == n
        m.alpha_prob.data.copy_(T(npz['alg|%d|alpha' % n]))
        m.alpha_gate.grad = T(npz['alg|%d|gate_grad' % n].copy())
        MixedOp.MODE = 'full'
        m.set_arch_param_grad()
        MixedOp.MODE = None
        assert rel_err(m.alpha_prob.grad.numpy(), npz['alg|%d|prob_grad' % n]) < 1e-5
        assert rel_err(m.probs_over_ops.detach().numpy(), npz['alg|%d|probs' % n]) < 1e-6
        assert m.chosen_index[0] == int(npz['alg|%d|chosen' % n])
        m.set_chosen_op_active()
        assert m.active_index == [int(npz['alg|%d|chosen' % n])]
    # 'two' mode: gradient over the sampled pair and the mass-preserving rescale (mixed.py:179-208)
    for kind in ('enc_safe', 'dec_safe'):
        tag = 'mx|two|%s|' % kind
        c = cases.mixed_case('two', kind, int(npz[tag + 'seed']))
        m = MixedOp(c['cfg'], kind)
        m.alpha_prob.data.copy_(T(npz[tag + 'alpha_old']))
        m.set_active(c['act'], c['inact'])
        assert m.alpha_gate.data[c['act'][0]] == 1 and m.alpha_gate.data.sum() == 1
        m.alpha_gate.grad = T(npz[tag + 'gate_grad'].copy())
        MixedOp.MODE = 'two'
        m.set_arch_param_grad()
        assert rel_err(m.alpha_prob.grad.numpy(), npz[tag + 'prob_grad']) < 1e-5
        m.alpha_prob.data.copy_(T(npz[tag + 'alpha_stepped']))
        m.rescale_updated_arch_param()
        MixedOp.MODE = None
        assert rel_err(m.alpha_prob.data.numpy(), npz[tag + 'alpha_rescaled']) < 1e-5


def test_sampler_is_seeded_and_mode_aware():
    from mmnas.model import mixed
    probs = torch.tensor([0.1, 0.2, 0.3, 0.4])
    mixed.seed_arch_sampler(5)
    a = [mixed.sample_indices(probs, None) for _ in range(20)]
    mixed.seed_arch_sampler(5)
    b = [mixed.sample_indices(probs, None) for _ in range(20)]
    assert a == b
    for act, inact in a:
        assert len(act) == 1 and sorted(act + inact) == [0, 1, 2, 3]
    mixed.seed_arch_sampler(6)
    for _ in range(20):
        act, inact = mixed.sample_indices(probs, 'two')
        assert len(act) == 1 and len(inact) == 1 and act[0] != inact[0]


def test_supernet_structure_and_prior_on_host():
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas.model.full_vqa import Net_Full
    npz = load('nets.npz')
    c = cases.net_case('vqa', None, 1, search=True)
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = Net_Search(c['cfg'], init)
    assert set(net.state_dict().keys()) == set(c['P'].keys())       # incl. the 'backnone' spelling
    ia = np.stack([np.pad(p.detach().numpy(), (0, 4 - p.numel())) for p in net.alpha_prob_parameters()])
    assert np.array_equal(ia, npz['search|vqa|init_alpha'])
    assert len(net.redundant_modules) == 30
    assert len(list(net.alpha_gate_parameters())) == 30
    n_net = sum(1 for _ in net.net_parameters())
    assert n_net + 60 == sum(1 for _ in net.parameters())
    net.reset_binary_gates()        # CPU parameters: pure bookkeeping, no operator is evaluated
    net.unused_modules_off()
    for m in net.redundant_modules:
        assert sum(op is not None for op in m.candidate_ops) == 1
    net.unused_modules_back()
    for m in net.redundant_modules:
        assert all(op is not None for op in m.candidate_ops)
    cf = cases.net_case('vqa', 'mmnas_vqa', 2)
    full = Net_Full(cf['cfg'], init)
    assert set(full.state_dict().keys()) == set(cf['P'].keys())


def test_sampling_probability_cache_follows_the_alphas():
    """reset_binary_gates() keeps softmax(alpha) on the host between architecture updates (no device->host copy per
    weight step); an in-place optimizer update, a load_state_dict and the rescale step's `.data` write must each
    refresh it (host logic only: CPU parameters)."""
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas.model.mixed import MixedOp
    from mmnas_amd.model import mixed
    c = cases.net_case('vqa', None, 1, search=True)
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = Net_Search(c['cfg'], init)
    mixed.seed_arch_sampler(5)
    net.reset_binary_gates()
    first = net._probs_cache[1]
    net.reset_binary_gates()
    assert net._probs_cache[1] is first                                   # reused
    for m in net.redundant_modules:                                       # gates are one-hot of the sampled index
        g = m.alpha_gate.detach().numpy()
        assert g.sum() == 1.0 and g[m.active_index[0]] == 1.0
    # 1. in-place update through the Parameter (what torch.optim.Adam does): node 0 collapses onto its last choice
    opt = torch.optim.SGD(list(net.alpha_prob_parameters()), lr=1.0)
    p0 = next(iter(net.alpha_prob_parameters()))
    p0.grad = torch.zeros_like(p0)
    p0.grad[-1] = -50.0
    opt.step()
    net.reset_binary_gates()
    assert net._probs_cache[1] is not first
    assert net.redundant_modules[0].active_index == [p0.numel() - 1]
    # 2. the rescale step writes through .data: MixedOp.alpha_version covers it
    MixedOp.MODE = 'two'
    try:
        net.reset_binary_gates()
        before = net._probs_cache[1]
        for m in net.redundant_modules:
            m.alpha_gate.grad = torch.ones_like(m.alpha_gate)
        net.set_arch_param_grad()
        for p in net.alpha_prob_parameters():
            p.data.add_(0.25 * torch.arange(p.numel(), dtype=p.dtype))    # an update that bumps no version counter
        net.rescale_updated_arch_param()
        net.reset_binary_gates()
        assert net._probs_cache[1] is not before
    finally:
        MixedOp.MODE = None
    # 3. load_state_dict
    before = net._probs_cache[1]
    net.load_state_dict(net.state_dict())
    net.reset_binary_gates()
    assert net._probs_cache[1] is not before

"""Deterministic case generators shared by make_golden.py (which runs the imported
reference on them, in the build container only) and by the tests (which run the oracle
and the HIP path on the very same inputs).  No reference import here.

Inputs and parameters are drawn from numpy's legacy MT19937 ``RandomState`` (frozen
stream, NEP 19) so that a fixture only has to store the *expected outputs*; a checksum of
the generated inputs is stored alongside to detect any drift.
"""
import json
import os
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def small_cfg(HSIZE=128, **over):
    c = dict(HSIZE=HSIZE, DROPOUT_R=0.0, REL_SIZE=64, OPS_NORM=True, OPS_RESIDUAL=True, LAYERS=1,
             NODES={'enc': 12, 'dec': 18}, ATTFLAT_GLIMPSES=1, ATTFLAT_OUT_SIZE=2 * HSIZE,
             ATTFLAT_MLP_SIZE=64, FRCNFEAT_SIZE=32, BBOX_FEATURE=False, BBOXFEAT_EMB_SIZE=16,
             WORD_EMBED_SIZE=24, ALPHA_INIT_TYPE='normal', SCORES_LOSS='kld', GENOTYPE=None)
    c.update(over)
    return SimpleNamespace(**c)


def _param(rs, key, shape):
    """Reference-independent synthetic parameter values (well-conditioned, every path exercised)."""
    if key.endswith('a_2'):
        return (1.0 + 0.2 * rs.standard_normal(shape)).astype(np.float32)
    if key.endswith('b_2') or key.endswith('bias') or 'bias_' in key:
        return (0.1 * rs.standard_normal(shape)).astype(np.float32)
    if key.endswith('alpha_prob'):
        return rs.standard_normal(shape).astype(np.float32)
    if key.endswith('alpha_gate'):
        return np.zeros(shape, np.float32)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    return (rs.standard_normal(shape) / np.sqrt(max(fan_in, 1))).astype(np.float32)


def rand_params(shapes, rs):
    return {k: _param(rs, k, tuple(s)) for k, s in shapes.items()}


def checksum(arrs):
    tot = 0.0
    for k in sorted(arrs):
        a = np.asarray(arrs[k])
        if a.dtype == np.bool_:
            a = a.astype(np.float64)
        tot += float(np.sum(a.astype(np.float64) * (1.0 + (np.arange(a.size).reshape(a.shape) % 7))))
    return tot


def masks(rs, B, S, full_pad_last=True):
    """bool [B,1,1,S], True = padded key.  sample 0 unpadded, last sample fully padded, others ragged tails."""
    m = np.zeros((B, 1, 1, S), dtype=np.bool_)
    for b in range(1, B):
        n = int(rs.randint(1, S)) if S > 1 else 1
        m[b, 0, 0, n:] = True
    if full_pad_last and B > 1:
        m[B - 1] = True
    return m


def op_case(name, norm, residual, seed, dims=None):
    """Inputs for one registry operator.  dims: dict(B,Sx,Sy,HSIZE)."""
    from oracle.mmnas_oracle import op_param_shapes, parse_op_name
    kind, kw = parse_op_name(name)
    dm = dict(B=3, Sx=7, Sy=5, HSIZE=128)
    if kw.get('base', 0) == 256:
        dm['HSIZE'] = 256
    if dims:
        dm.update(dims)
    cfg = small_cfg(HSIZE=dm['HSIZE'], OPS_NORM=norm, OPS_RESIDUAL=residual)
    rs = np.random.RandomState(seed)
    B, Sx, Sy, d = dm['B'], dm['Sx'], dm['Sy'], dm['HSIZE']
    P = rand_params(op_param_shapes(name, cfg, norm=norm), rs)
    x = rs.standard_normal((B, Sx, d)).astype(np.float32)
    y = rs.standard_normal((B, Sy, d)).astype(np.float32)
    x_mask = masks(rs, B, Sx)
    y_mask = masks(rs, B, Sy)
    rel = np.maximum(rs.standard_normal((B, Sx, Sx, cfg.REL_SIZE)), 0).astype(np.float32)
    gout = rs.standard_normal((B, Sx, d)).astype(np.float32)
    return dict(name=name, kind=kind, cfg=cfg, P=P, x=x, y=y, x_mask=x_mask, y_mask=y_mask, rel=rel,
                gout=gout, dims=dm)


def load_arch(name):
    """arch/<name>.json -> genotype dict (train_vqa.py:185 reads ['epoch'+str(N)])."""
    with open(os.path.join(REPO, 'arch', name + '.json')) as f:
        d = json.load(f)
    return d[sorted(d.keys())[-1]]


def net_param_shapes(task, cfg, token_size, ans_size, genotype=None, search=False):
    """Full state_dict key -> shape of Net_Full / Net_Search (key order follows module registration)."""
    from oracle.mmnas_oracle import op_param_shapes, USED_OPS
    d = cfg.HSIZE
    sh = {}
    sh['embedding.weight'] = (token_size, cfg.WORD_EMBED_SIZE)
    sh['lstm.weight_ih_l0'] = (4 * d, cfg.WORD_EMBED_SIZE)
    sh['lstm.weight_hh_l0'] = (4 * d, d)
    sh['lstm.bias_ih_l0'] = (4 * d,)
    sh['lstm.bias_hh_l0'] = (4 * d,)
    fs = cfg.FRCNFEAT_SIZE
    if cfg.BBOX_FEATURE:
        sh['bboxfeat_linear.weight'] = (cfg.BBOXFEAT_EMB_SIZE, 5)
        sh['bboxfeat_linear.bias'] = (cfg.BBOXFEAT_EMB_SIZE,)
        fs += cfg.BBOXFEAT_EMB_SIZE
    sh['imgfeat_linear.weight'] = (d, fs)
    sh['imgfeat_linear.bias'] = (d,)
    if search and task == 'itm':   # hygr_itm.py registers the rel linears before the backbone
        sh['linear_x_rel.weight'] = (cfg.REL_SIZE, 3); sh['linear_x_rel.bias'] = (cfg.REL_SIZE,)
        sh['linear_y_rel.weight'] = (cfg.REL_SIZE, 4); sh['linear_y_rel.bias'] = (cfg.REL_SIZE,)
    if (not search) and task == 'itm':  # full_itm.py registers linear_y_rel before the backbone
        sh['linear_y_rel.weight'] = (cfg.REL_SIZE, 4); sh['linear_y_rel.bias'] = (cfg.REL_SIZE,)
    for kind in ('enc', 'dec'):
        for l in range(cfg.LAYERS):
            base = 'backnone.cells_%s.%d.dag.' % (kind, l)
            if search:
                names = USED_OPS[kind + '_safe']
                for ni in range(cfg.NODES[kind]):
                    key = '%s%d.0.' % (base, ni)
                    sh[key + 'alpha_prob'] = (len(names),)
                    sh[key + 'alpha_gate'] = (len(names),)
                    for ci, nm in enumerate(names):
                        for k, s in op_param_shapes(nm, cfg, norm=cfg.OPS_NORM).items():
                            sh['%scandidate_ops.%d.%s' % (key, ci, k)] = s
            else:
                for ni, node in enumerate(genotype[kind]):
                    for j, nm in enumerate(node):
                        for k, s in op_param_shapes(nm, cfg, norm=cfg.OPS_NORM).items():
                            sh['%s%d.%d.%s' % (base, ni, j, k)] = s

    def attflat(pre):
        sh[pre + 'mlp.fc.linear.weight'] = (cfg.ATTFLAT_MLP_SIZE, d)
        sh[pre + 'mlp.fc.linear.bias'] = (cfg.ATTFLAT_MLP_SIZE,)
        sh[pre + 'mlp.linear.weight'] = (cfg.ATTFLAT_GLIMPSES, cfg.ATTFLAT_MLP_SIZE)
        sh[pre + 'mlp.linear.bias'] = (cfg.ATTFLAT_GLIMPSES,)
        sh[pre + 'linear_merge.weight'] = (cfg.ATTFLAT_OUT_SIZE, d * cfg.ATTFLAT_GLIMPSES)
        sh[pre + 'linear_merge.bias'] = (cfg.ATTFLAT_OUT_SIZE,)
    attflat('attflat_x.')
    if task == 'vgd':
        sh['attfc_y.weight'] = (cfg.ATTFLAT_OUT_SIZE, d); sh['attfc_y.bias'] = (cfg.ATTFLAT_OUT_SIZE,)
    else:
        attflat('attflat_y.')
    sh['proj_norm.a_2'] = (cfg.ATTFLAT_OUT_SIZE,)
    sh['proj_norm.b_2'] = (cfg.ATTFLAT_OUT_SIZE,)
    if task == 'vgd':
        sh['proj_scores.weight'] = (1, cfg.ATTFLAT_OUT_SIZE); sh['proj_scores.bias'] = (1,)
        sh['proj_reg.weight'] = (4, cfg.ATTFLAT_OUT_SIZE); sh['proj_reg.bias'] = (4,)
    elif task == 'itm':
        sh['proj.weight'] = (1, cfg.ATTFLAT_OUT_SIZE); sh['proj.bias'] = (1,)
    else:
        sh['proj.weight'] = (ans_size, cfg.ATTFLAT_OUT_SIZE); sh['proj.bias'] = (ans_size,)
    if search and task == 'vqa':
        sh['linear_x_rel.weight'] = (cfg.REL_SIZE, 3); sh['linear_x_rel.bias'] = (cfg.REL_SIZE,)
        sh['linear_y_rel.weight'] = (cfg.REL_SIZE, 4); sh['linear_y_rel.bias'] = (cfg.REL_SIZE,)
    if search and task == 'vgd':
        sh['linear_y_rel.weight'] = (cfg.REL_SIZE, 4); sh['linear_y_rel.bias'] = (cfg.REL_SIZE,)
    if (not search) and task in ('vqa', 'vgd'):
        sh['linear_y_rel.weight'] = (cfg.REL_SIZE, 4); sh['linear_y_rel.bias'] = (cfg.REL_SIZE,)
    return sh


def net_inputs(rs, cfg, B, Sx, Sy, token_size):
    """The 5-tuple of hygr_vqa.py:92-93 with ragged padding (SURVEY 3.1 tensor contract)."""
    frcn = np.maximum(rs.standard_normal((B, Sy, cfg.FRCNFEAT_SIZE)), 0).astype(np.float32)
    y_rel = rs.standard_normal((B, Sy, Sy, 4)).astype(np.float32)
    ques = rs.randint(1, token_size, size=(B, Sx)).astype(np.int64)
    x_rel = rs.standard_normal((B, Sx, Sx, 3)).astype(np.float32)
    for b in range(B):
        ny = int(rs.randint(max(1, Sy // 2), Sy + 1))
        frcn[b, ny:] = 0
        y_rel[b, ny:] = 0
        y_rel[b, :, ny:] = 0
        nx = int(rs.randint(min(3, Sx), Sx + 1))
        ques[b, nx:] = 0
        x_rel[b, nx:] = 0
        x_rel[b, :, nx:] = 0
    bbox = np.zeros((B, Sy, 5), np.float32)
    return frcn, bbox, y_rel, ques, x_rel


def net_case(task, arch, seed, search=False, HSIZE=128, B=2, Sx=5, Sy=7, token_size=40, ans_size=13, cfg_over=None):
    cfg = small_cfg(HSIZE=HSIZE, **(cfg_over or {}))
    genotype = None
    if not search:
        genotype = load_arch(arch)
        cfg.GENOTYPE = genotype
    rs = np.random.RandomState(seed)
    P = rand_params(net_param_shapes(task, cfg, token_size, ans_size, genotype, search), rs)
    inputs = net_inputs(rs, cfg, B, Sx, Sy, token_size)
    if task == 'vqa':
        target = (rs.uniform(size=(B, ans_size)) * (rs.uniform(size=(B, ans_size)) < 0.2)).astype(np.float32)
    elif task == 'itm':
        target = (rs.uniform(size=(B,)) < 0.5).astype(np.float32)
    else:
        target = rs.standard_normal((B, Sy)).astype(np.float32)
    return dict(task=task, arch=arch, cfg=cfg, genotype=genotype, P=P, inputs=inputs, target=target,
                token_size=token_size, ans_size=ans_size)


# ---- whole networks at the entry scripts' own dimensions (make_golden.gen_nets_full -> nets_full.npz) -----------------
# search_vqa.py:87-114 (HSIZE 256, ATTFLAT_OUT 512), train_vqa.py:136-154 / train_vgd.py / train_itm.py:143-154 (HSIZE 512,
# ATTFLAT_OUT 1024); FRCNFEAT 2048, ATTFLAT_MLP 512, GloVe 300, 3129 answers; 100 regions + 14 tokens (VGD 15 tokens; ITM 36
# regions + 50 tokens); BASELINE configs[0] literally: arch/mcan.json, batch 4, 36 regions, 14 tokens.  Only the batch is
# small (2-4: the CPU reference and the CPU oracle run these in seconds) and the vocabulary (2000 rows: the embedding table is
# outside the path).  Dropout 0 (the reference's Philox stream cannot be replayed).
FULL_CASES = (
    # tag-kind, task, arch, HSIZE, B, Sx, Sy, mode
    ('full', 'vqa', 'mcan', 512, 4, 14, 36, None),
    ('full', 'vqa', 'mmnas_vqa', 512, 2, 14, 100, None),
    ('full', 'vgd', 'mmnas_vgd', 512, 2, 15, 100, None),
    ('full', 'itm', 'mmnas_itm', 512, 2, 50, 36, None),
    ('search', 'vqa', None, 256, 2, 14, 100, None),
    ('search', 'vqa', None, 256, 2, 14, 100, 'full'),
)
FULL_SEED0 = 9500
# the same at the FULL batch of BASELINE configs[2] (supernet weight step, HSIZE 256), configs[1] (train_vqa, HSIZE 512), configs[3]
# (train_vgd) and configs[4] (train_itm, B = 160): B = 64 / 160
# (make_golden.gen_nets_full64 -> nets_full64.npz; the reference runs these on the CPU in a minute or two each)
FULL64_CASES = (
    ('search', 'vqa', None, 256, 64, 14, 100, None),
    ('full', 'vqa', 'mmnas_vqa', 512, 64, 14, 100, None),
    ('full', 'vgd', 'mmnas_vgd', 512, 64, 15, 100, None),      # configs[3]: train_vgd at its batch
    ('full', 'itm', 'mmnas_itm', 512, 160, 50, 36, None),      # configs[4]: train_itm at its batch (one forward of the net)
    ('search', 'vqa', None, 256, 64, 14, 100, 'full'),         # configs[2]'s architecture step (MODE 'full': all 96 candidates)
)
FULL64_SEED0 = 9700


def full_case_tag(spec):
    kind, task, arch, d, B, Sx, Sy, mode = spec
    return '%s|%s|%s|%d_%d_%d_%d|' % (kind, task, arch if kind == 'full' else mode, d, B, Sx, Sy)


def net_case_full(spec, seed):
    """One FULL_CASES entry -> the case dict of net_case at production dimensions (+ the injected sample for a supernet)."""
    kind, task, arch, d, B, Sx, Sy, mode = spec
    search = kind == 'search'
    cfg_over = dict(FRCNFEAT_SIZE=2048, ATTFLAT_MLP_SIZE=512, WORD_EMBED_SIZE=300, ATTFLAT_OUT_SIZE=2 * d, BBOXFEAT_EMB_SIZE=1024)
    ans = 3129 if task == 'vqa' else 13
    c = net_case(task, arch, seed, search=search, HSIZE=d, B=B, Sx=Sx, Sy=Sy, token_size=2000, ans_size=ans, cfg_over=cfg_over)
    if search:
        c['plan'] = search_plan(np.random.RandomState(seed + 50000), mode)
        c['mode'] = mode
    return c


def search_plan(rs, mode):
    """Injected (active, inactive) index lists per node, bypassing torch.multinomial (mixed.py:138,151)."""
    plan = {'mode': mode, 'enc': [], 'dec': []}
    for kind, n_nodes, n_choice in (('enc', 12, 2), ('dec', 18, 4)):
        for _ in range(n_nodes):
            if mode == 'two':
                pair = rs.choice(n_choice, size=2, replace=False)
                plan[kind].append(([int(pair[0])], [int(pair[1])]))
            else:
                a = int(rs.randint(0, n_choice))
                plan[kind].append(([a], [i for i in range(n_choice) if i != a]))
    return plan


def mixed_param_shapes(kind, cfg):
    """state_dict of one MixedOp(cfg, kind) (mixed.py:39-55)."""
    from oracle.mmnas_oracle import op_param_shapes, USED_OPS
    names = USED_OPS[kind]
    sh = {'alpha_prob': (len(names),), 'alpha_gate': (len(names),)}
    for ci, nm in enumerate(names):
        for k, s in op_param_shapes(nm, cfg, norm=cfg.OPS_NORM).items():
            sh['candidate_ops.%d.%s' % (ci, k)] = s
    return sh


def mixed_case(mode, kind, seed):
    """One MixedOp forward/backward case with injected active/inactive indices."""
    from oracle.mmnas_oracle import USED_OPS
    cfg = small_cfg(HSIZE=128)
    rs = np.random.RandomState(seed)
    P = rand_params(mixed_param_shapes(kind, cfg), rs)
    n = len(USED_OPS[kind])
    if mode == 'two':
        pair = rs.choice(n, 2, replace=False)
        act, inact = [int(pair[0])], [int(pair[1])]
    else:
        a = int(rs.randint(0, n))
        act, inact = [a], [i for i in range(n) if i != a]
    P['alpha_gate'][:] = 0
    P['alpha_gate'][act[0]] = 1.0
    B, S, S2, d = 2, 6, 4, 128
    s = rs.standard_normal((B, S, d)).astype(np.float32)
    pre = rs.standard_normal((B, S2, d)).astype(np.float32)
    sm = masks(rs, B, S, full_pad_last=False)
    pm = masks(rs, B, S2, full_pad_last=False)
    rel = np.maximum(rs.standard_normal((B, S, S, 64)), 0).astype(np.float32)
    g = rs.standard_normal((B, S, d)).astype(np.float32)
    return dict(cfg=cfg, P=P, act=act, inact=inact, s=s, pre=pre, sm=sm, pm=pm, rel=rel, g=g, kind=kind,
                mode=mode)


# ---- capture (v): optimizer trajectory of the bilevel loop (make_golden.gen_traj) ---------------------------------
# search_vqa.py:135-161 values except the learning rate (25x: visible parameter motion in two fp32 steps) and
# epoch_steps = 1, so the three warm-up rates of WarmupOptimizer (1/4, 2/4, 3/4) are all exercised
TRAJ_HYPER = dict(net_lr=0.002, net_betas=(0.9, 0.98), net_eps=1e-9, clip=1.0, epoch_steps=1,
                  alpha_lr=0.1, alpha_betas=(0.0, 0.999))
TRAJ_FULL_KEYS = ('proj.bias', 'proj_norm.a_2', 'linear_y_rel.weight', 'imgfeat_linear.bias',
                  'backnone.cells_enc.0.dag.0.0.candidate_ops.0.ln.a_2',
                  'backnone.cells_enc.0.dag.0.0.candidate_ops.1.ln.a_2',
                  'backnone.cells_dec.0.dag.3.0.candidate_ops.2.mhatt.linear_merge.weight')


def traj_setup(seed=9100, full64=False):
    """Weights + batch of the weight steps, batch of the arch step, and the four injected samples.  full64: the same at
    BASELINE configs[2]'s own dimensions and batch (HSIZE 256, B = 64, 100 regions, 14 tokens, 3129 answers; traj64.npz)."""
    if full64:
        spec = ('search', 'vqa', None, 256, 64, 14, 100, None)
        c = net_case_full(spec, seed + 700)
        c2 = net_case_full(spec, seed + 701)
        rs = np.random.RandomState(seed + 50700)
    else:
        c = net_case('vqa', None, seed, search=True)
        c2 = net_case('vqa', None, seed + 1, search=True)   # only its inputs/target are used (the eval_loader batch)
        rs = np.random.RandomState(seed + 50000)
    plans = [search_plan(rs, None), search_plan(rs, None), search_plan(rs, 'full'), search_plan(rs, None)]
    return c, c2, plans


# ---- answer targets (make_golden.gen_loader: proc_ans / get_score, load_data_vqa.py:299-333) ---------------------------
LOADER_ANSWERS = ['yes', 'no', 'red', 'blue', '2', 'frisbee', 'left', 'tennis', 'white']
LOADER_ANSWER_SETS = (
    ['yes'] * 10,
    ['red'] * 3 + ['blue'] * 2 + ['white'] + ['maroon'] * 4,              # 'maroon' is not in the answer vocabulary
    ['2'] * 1 + ['frisbee'] * 4 + ['left'] * 5,
    ['purple'] * 10,                                                      # nothing in the vocabulary: all-zero target
    ['no'] * 2 + ['yes'] * 3 + ['tennis'] * 1 + ['white'] * 4,
)


# ---- the fixed-architecture training loop (make_golden.gen_train_traj; train_vqa.py:291-311) -------------------------
TRAIN_HYPER = dict(lr=0.002, betas=(0.9, 0.98), eps=1e-9, clip=1.0, epoch_steps=1, decay_r=0.2)
TRAIN_FULL_KEYS = ('proj.bias', 'proj_norm.a_2', 'linear_y_rel.weight', 'imgfeat_linear.bias', 'lstm.bias_hh_l0',
                   'backnone.cells_enc.0.dag.0.0.ln.a_2', 'backnone.cells_dec.0.dag.0.0.mhatt.linear_merge.weight')


def train_traj_setup(seed=9300, full64=False):
    """Weights + two batches (steps alternate between them) for arch/mmnas_vqa.json.  full64: at BASELINE configs[1]'s own
    dimensions and batch (HSIZE 512, B = 64, 100 regions, 14 tokens, 3129 answers; train_traj64.npz)."""
    if full64:
        spec = ('full', 'vqa', 'mmnas_vqa', 512, 64, 14, 100, None)
        return net_case_full(spec, seed + 700), net_case_full(spec, seed + 701)
    c = net_case('vqa', 'mmnas_vqa', seed)
    c2 = net_case('vqa', 'mmnas_vqa', seed + 1)    # only its inputs / target are used
    return c, c2


# ---- loader functions (make_golden.gen_loader) ---------------------------------------------------------------------
LOADER_VOCAB = ['PAD', 'UNK', 'what', 'is', 'the', 'man', 'holding', 'color', 'of', 'cat', 'how', 'many', 'dogs',
                'are', 'there', 'in', 'this', 'picture', 'on', 'table', 'a', 'red', 'ball', 'left', 'right', 'side']
LOADER_QUESTIONS = ('What is the man holding?', "What color is the cat's ball?", 'How many dogs are there?',
                    'Is there a red ball on the left-side of the table in this picture, or on the right/left side of it?',
                    'Why?')


def loader_boxes(n, seed, w=640.0, h=480.0):
    """n boxes (x1, y1, x2, y2) inside a w x h image, float32; includes a duplicated box and equal centres."""
    rs = np.random.RandomState(seed)
    x1 = rs.uniform(0, w * 0.8, n); y1 = rs.uniform(0, h * 0.8, n)
    x2 = x1 + rs.uniform(1, w * 0.2, n); y2 = y1 + rs.uniform(1, h * 0.2, n)
    b = np.stack([x1, y1, x2, y2], 1).astype(np.float32)
    if n > 2:
        b[2] = b[0]                 # identical boxes: delta clamps at 1e-3
        b[1, 0::2] = b[0, 0::2]     # same centre x, different height
    return b


def loader_glove(V, seed):
    e = np.random.RandomState(seed).standard_normal((V, 300)).astype(np.float32)
    e[0] = 0                        # PAD row, as in the spaCy table
    return e


def losses_cases(full64=False):
    """(ITM positive batch, ITM negative batch, VGD batch) of make_golden.gen_losses: small (losses.npz) or at BASELINE configs[4] /
    configs[3]'s own dimensions and batch (ITM: HSIZE 512, B = 160, 50 tokens, 36 regions; VGD: B = 64, 15 tokens, 100 regions;
    losses64.npz)."""
    if full64:
        itm = ('full', 'itm', 'mmnas_itm', 512, 160, 50, 36, None)
        vgd = ('full', 'vgd', 'mmnas_vgd', 512, 64, 15, 100, None)
        return net_case_full(itm, 9901), net_case_full(itm, 9902), net_case_full(vgd, 9903)
    return net_case('itm', 'mmnas_itm', 9201), net_case('itm', 'mmnas_itm', 9202), net_case('vgd', 'mmnas_vgd', 9203)


def vgd_targets(c, seed):
    """The VGD step's supervision tensors (train_vgd.py:309-313): soft scores, score mask, box targets, box mask."""
    rs = np.random.RandomState(seed)
    B, Sy = c['inputs'][0].shape[:2]
    scores = rs.uniform(size=(B, Sy)).astype(np.float32)
    scores /= scores.sum(-1, keepdims=True)
    smask = (rs.uniform(size=(B, Sy)) < 0.7).astype(np.float32)
    smask[:, 0] = 1
    bbox = rs.standard_normal((B, Sy, 4)).astype(np.float32)
    bmask = (rs.uniform(size=(B, Sy, 1)) < 0.4).astype(np.float32) * np.ones((1, 1, 4), np.float32)
    bmask[:, 0] = 1
    return dict(scores=scores, scores_mask=smask, bbox=bbox, bbox_mask=bmask)

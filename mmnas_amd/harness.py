"""Step harness: the statement sequences of the reference's entry scripts around the hot path, as calls.

The reference scripts cannot be imported (datasets, spaCy, torchvision); what they DO per step is short and is
restated here against the same model API, so that the bench, the tests and an integrator's loop share one
implementation:

  * SearchLoop.weight_step / arch_step / bilevel_round -- search_vqa.py:279-337 (and its _vgd / _itm twins):
    sample -> forward -> loss -> backward -> (gradient exchange) -> clip + Adam, and every ALPHA_EVERY-th step the
    architecture step in mode 'full'.  Differences from the script are mechanical, not arithmetic:
      - no `0 * sum(p.sum())` terms (search_vqa.py:285-288): they exist to give DDP a gradient for every parameter.
        Their arithmetic consequence -- torch Adam steps EVERY net parameter at every step, unsampled candidates
        included (zero gradient, decaying moments, stale-momentum motion) -- is kept by FlatAdam(absent_grads='zero');
      - gradients live in one flat buffer (dp.SupernetReducer); clip_grad_norm_ + Adam are two kernels (optim.FlatAdam);
      - the arch step's alpha gradient + alpha_optim.step() are one kernel (ArchAdam, mode 'full').
    Pinned against the reference loop itself by tests/golden/traj.npz (tests/test_harness_gpu.py::test_bilevel_trajectory_vs_reference_loop).
  * itm_triplet_step -- train_itm.py:380-391: three forwards (positive, negative caption, negative image), BCE_Loss.
  * BCE_Loss -- mmnas/utils/itm_loss.py:4-24.  vgd_loss -- train_vgd.py:316-333.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dp, ops
from .model.mixed import MixedOp
from .optim import FlatAdam, WarmupOptimizer


class BCEWithLogitsSum(nn.Module):
    """torch.nn.BCEWithLogitsLoss(reduction='sum') (search_vqa.py:211) as one HIP kernel per direction on the GPU."""

    def forward(self, pred, target):
        if pred.is_cuda and pred.dtype == torch.float32:
            return ops.bce_with_logits_sum(pred, target)
        return F.binary_cross_entropy_with_logits(pred, target, reduction='sum')


def fused_loss(loss_fn):
    """The HIP form of a loss module the scripts use, when there is one."""
    if isinstance(loss_fn, nn.BCEWithLogitsLoss) and loss_fn.reduction == 'sum' and loss_fn.weight is None and loss_fn.pos_weight is None:
        return BCEWithLogitsSum()
    return loss_fn


class BCE_Loss(nn.Module):
    """mmnas/utils/itm_loss.py:4-24: BCE on the sigmoid scores, label 1 for the matching pair and 0 for the two
    negatives; the positive term enters twice (`loss_pos + loss_negc + loss_pos + loss_negi`)."""

    def __init__(self, __C=None):
        super().__init__()
        self.reduction = getattr(__C, 'REDUCTION', 'sum') if __C is not None else 'sum'

    def forward(self, scores_pos, scores_negc, scores_negi):
        lp = F.binary_cross_entropy(scores_pos, torch.ones_like(scores_pos), reduction=self.reduction)
        lc = F.binary_cross_entropy(scores_negc, torch.zeros_like(scores_negc), reduction=self.reduction)
        li = F.binary_cross_entropy(scores_negi, torch.zeros_like(scores_negi), reduction=self.reduction)
        return lp + lc + lp + li


def vgd_loss(pred_scores, pred_reg, scores, scores_mask, bbox, bbox_mask, lam=0.5, scores_loss='kld', loss_avg=True,
             batch_size=None):
    """train_vgd.py:316-333 (REDUCTION='sum'): KLDiv (or BCE-with-logits) on the masked scores + lam * SmoothL1 on the
    masked box targets, each divided by its mask count when LOSS_AVG."""
    if scores_loss == 'bce':
        ls = F.binary_cross_entropy_with_logits(pred_scores, scores, reduction='sum')
    else:
        ls = F.kl_div(pred_scores * scores_mask, scores * scores_mask, reduction='sum')
    lr = F.smooth_l1_loss(pred_reg * bbox_mask, bbox * bbox_mask, reduction='sum')
    if loss_avg:
        if batch_size is None:
            batch_size = pred_scores.shape[0]     # train_vgd.py divides by the loader's batch size
        ls = ls / (batch_size if scores_loss == 'bce' else scores_mask.sum())
        lr = lr / bbox_mask.sum()
    return ls + lam * lr


def itm_triplet_step(net, loss_fn, pos, neg, reducer=None, optim=None):
    """train_itm.py:380-395: `pos` / `neg` are the 5-tuples (frcn, bbox, rel_img, cap_ix, rel_cap) of the matching
    pair and of the mined negatives; three forwards share the weights, one backward."""
    if reducer is not None:
        reducer.begin_step()
    elif optim is not None:
        optim.zero_grad()
    negc = (pos[0], pos[1], pos[2], neg[3], neg[4])
    negi = (neg[0], neg[1], neg[2], pos[3], pos[4])
    loss = loss_fn(net(pos), net(negc), net(negi))
    loss.backward()
    if reducer is not None:
        reducer.finish()
    if optim is not None:
        optim.step()
    return loss


def hard_negative_indices(scores, neg_idx, hard_size):
    """The selection step of the ITM hard-negative mining pass (train_itm.py:349-353): `scores` are the net's matching
    scores of every anchor against its NEG_RANDSIZE random candidates (flattened), `neg_idx` [N, NEG_RANDSIZE] the
    candidates' dataset indices; returns [N, hard_size] -- per anchor the indices of its highest-scoring candidates."""
    scores = scores.view(-1, neg_idx.shape[1])
    top = torch.argsort(scores, dim=-1, descending=True)[:, :hard_size]
    rows = torch.arange(top.size(0), device=top.device).unsqueeze(1).expand_as(top)
    return neg_idx.to(scores.device)[rows, top]


class ArchAdam:
    """alpha_optim of search_vqa.py:194 (torch.optim.Adam over alpha_prob_parameters, lr 0.1, betas (0, 0.999)) fused
    with Net_Search.set_arch_param_grad() for ALPHA_BINARY_MODE 'full': one kernel over the [n_nodes, width] blocks."""

    def __init__(self, net, lr=0.1, betas=(0.0, 0.999), eps=1e-8):
        self.net, self.lr, self.betas, self.eps = net, lr, betas, eps
        prob, _ = net._flat_alphas()
        self.m = torch.zeros_like(prob)
        self.v = torch.zeros_like(prob)
        self.steps = 0

    def step(self):
        net = self.net
        prob, _ = net._flat_alphas()
        gg, pg = net._flat_grads
        for i, m in enumerate(net.redundant_modules):    # gate gradients autograd produced outside the block
            g = m.alpha_gate.grad
            if g is not None and g.data_ptr() != gg[i].data_ptr():
                gg[i, :m.n_choices].copy_(g)
        self.steps += 1
        ops.alpha_full_step(prob, gg, self.m, self.v, pg, self.lr, self.betas, self.eps, self.steps)
        for i, m in enumerate(net.redundant_modules):
            m.alpha_prob.grad = pg[i, :m.n_choices]
            m.alpha_version += 1                          # (the update wrote through the flat block: drop the sampling cache)

    # torch.optim.Adam's checkpoint format over alpha_prob_parameters() in module order (search_vqa.py:350: the reference
    # saves `alpha_optim.state_dict()` beside the network's)
    def state_dict(self):
        state = {}
        mods = self.net.redundant_modules
        if self.steps:
            for i, m in enumerate(mods):
                n = m.n_choices
                state[i] = {'step': torch.tensor(float(self.steps)), 'exp_avg': self.m[i, :n].clone(),
                            'exp_avg_sq': self.v[i, :n].clone()}
        group = {'lr': self.lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': 0, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'params': list(range(len(mods)))}
        return {'state': state, 'param_groups': [group]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        mods = self.net.redundant_modules
        g0 = sd['param_groups'][0]
        order = [i for g in sd['param_groups'] for i in g['params']]
        if len(order) != len(mods):
            raise ValueError('ArchAdam.load_state_dict: %d parameters in the file, %d alpha blocks here' % (len(order), len(mods)))
        self.lr, self.betas, self.eps = g0.get('lr', self.lr), tuple(g0.get('betas', self.betas)), g0.get('eps', self.eps)
        self.m.zero_()
        self.v.zero_()
        steps = set()
        for pos, key in enumerate(order):
            st = sd['state'].get(key)
            if st is None:
                continue
            n = mods[pos].n_choices
            self.m[pos, :n].copy_(st['exp_avg'].to(self.m.device, torch.float32))
            self.v[pos, :n].copy_(st['exp_avg_sq'].to(self.v.device, torch.float32))
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError('ArchAdam keeps one step count for all alpha blocks; the file holds %s' % sorted(steps))
        self.steps = steps.pop() if steps else 0


class TrainLoop:
    """The fixed-architecture training loop body of train_vqa.py:291-311 (train_vgd.py / train_itm.py use the same
    statements around their own losses) for one data-parallel rank: zero_grad, forward, loss, backward with the
    bucketed gradient exchange, clip_grad_norm_, warm-up Adam step -- the clip and the update as two launches over the
    flat buffers.  `decay(r)` is the epoch-boundary learning-rate decay (train_vqa.py:285-287)."""

    def __init__(self, net, loss_fn=None, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, clip=1.0, epoch_steps=1000, warmup=True,
                 group=None, bucket_mb=64.0, force_collectives=False):
        self.net = net
        self.loss_fn = fused_loss(loss_fn if loss_fn is not None else nn.BCEWithLogitsLoss(reduction='sum'))
        self.reducer = dp.GradReducer(list(net.parameters()), bucket_mb=bucket_mb, group=group,
                                      force_collectives=force_collectives)
        # every parameter of a fixed architecture receives a gradient, so "absent gradients" never occur; 'zero' keeps
        # the arithmetic of the reference's `0 * sum(p.sum())` line (train_vqa.py:299) for parameters an architecture
        # leaves unused
        self.net_optim = WarmupOptimizer(lr, FlatAdam(self.reducer.fg.params, betas=betas, eps=eps, grads=self.reducer.fg,
                                                      absent_grads='zero'),
                                         epoch_steps=epoch_steps, warmup=warmup, max_norm=clip if clip and clip > 0 else None)
        self.steps = 0

    def step(self, inputs, target, optimize=True):
        red = self.reducer
        red.begin_step()
        loss = self.loss_fn(self.net(inputs), target)
        loss.backward()
        red.finish()
        if optimize:
            self.net_optim.step()
        self.steps += 1
        return loss

    def decay(self, decay_r):
        self.net_optim.decay(decay_r)


class SearchLoop:
    """The bilevel NAS loop body of search_vqa.py:279-337 for one data-parallel rank."""

    def __init__(self, net, loss_fn=None, net_lr=4e-4, net_betas=(0.9, 0.98), net_eps=1e-9, clip=1.0, epoch_steps=1000,
                 warmup=True, alpha_lr=0.1, alpha_betas=(0.0, 0.999), alpha_every=5, arch_mode='full', group=None,
                 absent_grads='zero', n_buckets=3, force_collectives=False):
        if arch_mode not in ('full', 'two'):
            raise ValueError("ALPHA_BINARY_MODE is 'full' or 'two' (search_vqa.py:151), got %r" % (arch_mode,))
        self.net = net
        self.loss_fn = fused_loss(loss_fn if loss_fn is not None else nn.BCEWithLogitsLoss(reduction='sum'))
        dense = absent_grads == 'zero'
        self.reducer = dp.SupernetReducer(net, group=group, n_buckets=n_buckets, force_collectives=force_collectives,
                                          attach_all=dense)
        net.keep_candidate_grads = dense
        self.net_optim = WarmupOptimizer(net_lr, FlatAdam(self.reducer.fg.params, betas=net_betas, eps=net_eps,
                                                          grads=self.reducer.fg, absent_grads=absent_grads),
                                         epoch_steps=epoch_steps, warmup=warmup, max_norm=clip if clip and clip > 0 else None)
        # 'full' (the shipped setting): alpha gradient + Adam as one kernel over the [n_nodes, width] blocks.  'two': the
        # reference's own statements -- MixedOp.set_arch_param_grad over the sampled pair, torch Adam on the alpha
        # parameters, rescale_updated_arch_param (search_vqa.py:330-334, mixed.py:179-208)
        if arch_mode == 'full':
            self.alpha_optim = ArchAdam(net, alpha_lr, alpha_betas)
        else:
            net._flat_alphas()          # (the parameters' storage moves into the flat blocks before Adam sees them)
            self.alpha_optim = torch.optim.Adam(list(net.alpha_prob_parameters()), alpha_lr, betas=tuple(alpha_betas))
        self.alpha_every = alpha_every
        self.arch_mode = arch_mode
        self.steps = 0

    def _sample(self, plan):
        if plan is None:
            self.net.reset_binary_gates()
        else:            # injected (active, inactive) lists per node: tests / replay of a logged search
            self.net.set_sampled(plan)
            if not getattr(self.net, 'keep_candidate_grads', False):
                for m in self.net.redundant_modules:
                    m.clear_candidate_grads()

    def weight_step(self, inputs, target, optimize=True, plan=None):
        net, red = self.net, self.reducer
        MixedOp.MODE = None
        self._sample(plan)
        red.begin_weight_step()
        loss = self.loss_fn(net(inputs), target)
        loss.backward()
        red.finish_weight_step()
        if optimize:
            self.net_optim.step()
        self.steps += 1
        return loss

    def arch_step(self, inputs, target, optimize=True, plan=None):
        net, red = self.net, self.reducer
        MixedOp.MODE = self.arch_mode
        try:
            self._sample(plan)
            net.begin_arch_step()
            # the network weights take no update here (search_vqa.py:331 steps alpha_optim only); their gradients are
            # produced -- as in the reference -- into the flat buffer, which the next weight step zeroes
            red.fg.zero()
            red.fg.attach()
            loss = self.loss_fn(net(inputs), target)
            loss.backward()
            red.reduce_alpha_gate_grads()
            if self.arch_mode == 'two':
                for m in net.redundant_modules:      # (the script's net.zero_grad() before backward, search_vqa.py:328)
                    m.alpha_prob.grad = None
                net.set_arch_param_grad()
                if optimize:
                    self.alpha_optim.step()
                    net.rescale_updated_arch_param()
            elif optimize:
                self.alpha_optim.step()
        finally:
            MixedOp.MODE = None
        return loss

    def bilevel_round(self, train_batches, eval_batch):
        """ALPHA_EVERY weight steps on training batches, then one arch step on a held-out batch
        (search_vqa.py:149-150,303-305).  Returns the list of losses (device tensors)."""
        out = [self.weight_step(inp, tgt) for inp, tgt in train_batches]
        out.append(self.arch_step(*eval_batch))
        return out

"""Fused optimizer step for the bilevel loop: gradient clipping + Adam over flat
fp32 buffers, replacing `clip_grad_norm_` + `torch.optim.Adam` over ~230 small tensors
(search_vqa.py:296-300, train_vqa.py:308-311) and the reference's `WarmupOptimizer`
(mmnas/utils/optimizer.py).

Semantics kept from the reference stack:
  * torch.optim.Adam arithmetic.  What happens to a parameter WITHOUT a gradient is selectable:
      absent_grads='zero' (default) -- the reference loop's behaviour.  `MixedOp.binarize` clears the candidates'
        grads (mixed.py:160-163), but the loop then adds `0 * sum(p.sum() for p in net_parameters())` to the loss
        (search_vqa.py:285-288), so EVERY parameter -- unsampled candidates included -- reaches torch Adam with a
        (zero) gradient at every step: its moments decay, it keeps moving on stale momentum, and there is one
        global step count.  Here: one Adam launch over the whole flat buffer, absent gradients read as the zeros
        the buffer holds.  Pinned by tests/golden/traj.npz (the reference loop itself).
      absent_grads='skip' -- torch Adam's own rule for `grad is None` (the parameter is frozen, per-parameter step
        counts): what the loop would do WITHOUT the `0 * sum` lines.  Kept as an option, not the default.
  * clip_grad_norm_ over the parameters that have a gradient: total norm in one device scalar, the
    scale min(1, max_norm / (norm + 1e-6)) applied inside the Adam kernel (no host round trip);
  * WarmupOptimizer's schedule: lr = base * {1/4, 2/4, 3/4, 1} over the first three epochs, `decay()`.
"""
import torch

from . import _lib as L
from .dp import FlatGrads


def _note_write():
    from . import ops          # (ops imports nothing from here; late import keeps the module graph acyclic)
    ops.note_raw_parameter_write()


class FlatAdam:
    def __init__(self, params, lr=0.0, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0, grads=None, absent_grads='zero'):
        if absent_grads not in ('zero', 'skip'):
            raise ValueError("absent_grads must be 'zero' or 'skip'")
        self.absent_grads = absent_grads
        self.fg = grads if grads is not None else FlatGrads(list(params))
        self.params = self.fg.params
        dev = self.fg.flat.device
        if not self.fg.flat.is_cuda:
            raise L.MMNasHipError('FlatAdam runs on the MI355X only (no CPU fallback)')
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        # re-home the parameters into one flat buffer with the gradient buffer's layout
        self.flat_p = torch.zeros(self.fg.total, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.fg.offsets):
            view = self.flat_p[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.steps = [0] * len(self.params)   # per-parameter step counts ('skip' mode)
        self.global_step = 0                  # 'zero' mode
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self.param_groups = [{'lr': lr, 'params': self.params}]   # WarmupOptimizer writes param_groups[i]['lr']

    def zero_grad(self, set_to_none=True):
        """Zero the flat gradient buffer.  'zero' mode keeps every p.grad attached to its view (so the backward
        kernels keep accumulating straight into the buffer and a data-parallel reducer armed before this call still
        sees the gradients); 'skip' mode detaches them: there `grad is None` is the signal that freezes a parameter."""
        self.fg.zero()
        if self.absent_grads == 'zero':
            self.fg.attach()
        else:
            for p in self.params:
                p.grad = None

    def _live_runs(self):
        """Maximal runs of consecutive parameters that have a gradient and share a step count."""
        runs = []
        for i, p in enumerate(self.params):
            g = p.grad
            if g is None:
                continue
            view = self.fg.views[i]
            if g.data_ptr() != view.data_ptr():     # gradient produced outside the flat buffer: bring it in
                view.copy_(g)
                p.grad = view
            o, n = self.fg.offsets[i], p.numel()
            end = o + ((n + 63) // 64) * 64
            if runs and runs[-1][1] == o and runs[-1][2] == self.steps[i]:
                runs[-1][1] = end
                runs[-1][3].append(i)
            else:
                runs.append([o, end, self.steps[i], [i]])
        return runs

    def _step_dense(self, max_norm):
        """'zero' mode: Adam over the whole buffer in one launch, one global step."""
        lib = L.lib()
        st = L.stream()
        lr = self.param_groups[0]['lr']
        fg = self.fg
        stale = []
        for i, p in enumerate(self.params):
            if p.grad is None:
                if fg.dirty[i]:          # a view that held a gradient earlier and was then dropped: must read as zero
                    stale.append(i)
            else:
                fg.adopt(i)
        for i in stale:
            fg.views[i].zero_()
            fg.dirty[i] = False
        n = fg.total
        sumsq_ptr = None
        if max_norm is not None and max_norm > 0:
            self._sumsq.zero_()
            L.check(lib.mmnas_sumsq(L.fptr(fg.flat), n, L.fptr(self._sumsq), st))
            sumsq_ptr = L.fptr(self._sumsq)
        self.global_step += 1
        L.check(lib.mmnas_adam_step(L.fptr(self.flat_p), L.fptr(fg.flat), L.fptr(self.m), L.fptr(self.v), n, lr,
                                    self.betas[0], self.betas[1], self.eps, self.weight_decay, sumsq_ptr,
                                    float(max_norm or 0.0), self.global_step, st))
        _note_write()

    @torch.no_grad()
    def step(self, max_norm=None):
        if self.absent_grads == 'zero':
            return self._step_dense(max_norm)
        lib = L.lib()
        st = L.stream()
        lr = self.param_groups[0]['lr']
        runs = self._live_runs()
        if not runs:
            return
        sumsq_ptr = None
        if max_norm is not None and max_norm > 0:
            self._sumsq.zero_()
            for o, e, _, idx in runs:
                # padding between parameters is zero in the gradient buffer, so whole runs can be summed
                L.check(lib.mmnas_sumsq(L.fptr(self.fg.flat[o:e]), e - o, L.fptr(self._sumsq), st))
            sumsq_ptr = L.fptr(self._sumsq)
        for o, e, k, idx in runs:
            L.check(lib.mmnas_adam_step(L.fptr(self.flat_p[o:e]), L.fptr(self.fg.flat[o:e]), L.fptr(self.m[o:e]),
                                        L.fptr(self.v[o:e]), e - o, lr, self.betas[0], self.betas[1], self.eps,
                                        self.weight_decay, sumsq_ptr, float(max_norm or 0.0), k + 1, st))
            for i in idx:
                self.steps[i] += 1
        _note_write()

    def grad_norm(self):
        """Total gradient norm of the last clipped step (device -> host; diagnostics only)."""
        return float(self._sumsq.sqrt())

    # -- checkpoints: torch.optim.Adam's format, so that the reference's `'net_optim': net_optim.optimizer.state_dict()`
    #    (search_vqa.py:342-346, train_vqa.py:314-319) and its resume path interchange with a torch Adam's files ----------
    def state_dict(self):
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.fg.offsets)):
            k = self.global_step if self.absent_grads == 'zero' else self.steps[i]
            if k == 0:
                continue            # torch Adam creates a parameter's state at its first step
            n = p.numel()
            state[i] = {'step': torch.tensor(float(k)),
                        'exp_avg': self.m[o:o + n].view_as(p).clone(),
                        'exp_avg_sq': self.v[o:o + n].view_as(p).clone()}
        group = {'lr': self.param_groups[0]['lr'], 'betas': tuple(self.betas), 'eps': self.eps,
                 'weight_decay': self.weight_decay, 'amsgrad': False, 'maximize': False, 'foreach': None,
                 'capturable': False, 'differentiable': False, 'fused': None, 'params': list(range(len(self.params)))}
        return {'state': state, 'param_groups': [group]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        groups = sd['param_groups']
        order = [i for g in groups for i in g['params']]
        if len(order) != len(self.params):
            raise ValueError('FlatAdam.load_state_dict: %d parameters in the file, %d here' % (len(order), len(self.params)))
        g0 = groups[0]
        self.param_groups[0]['lr'] = g0.get('lr', self.param_groups[0]['lr'])
        self.betas, self.eps = tuple(g0.get('betas', self.betas)), g0.get('eps', self.eps)
        self.weight_decay = g0.get('weight_decay', self.weight_decay)
        self.m.zero_()
        self.v.zero_()
        self.steps = [0] * len(self.params)
        steps = []
        for pos, key in enumerate(order):
            st = sd['state'].get(key)
            if st is None:
                continue
            p, o = self.params[pos], self.fg.offsets[pos]
            n = p.numel()
            self.m[o:o + n].copy_(st['exp_avg'].reshape(-1).to(self.m.device, torch.float32))
            self.v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1).to(self.v.device, torch.float32))
            k = int(float(st['step']))
            self.steps[pos] = k
            steps.append(k)
        if self.absent_grads == 'zero':
            if steps and min(steps) != max(steps):
                raise ValueError("FlatAdam(absent_grads='zero') keeps ONE step count; the file holds %d..%d (written by a loop "
                                 "without the reference's `0 * sum` lines?): load it into absent_grads='skip'" % (min(steps), max(steps)))
            self.global_step = steps[0] if steps else 0



class WarmupOptimizer:
    """mmnas/utils/optimizer.py restated: lr warm-up over three epochs, decay(), set_start_step()."""

    def __init__(self, lr_base, optimizer, epoch_steps, warmup, max_norm=None):
        self.optimizer = optimizer
        self._step = 0
        self.lr_base = lr_base
        self._rate = 0
        self.epoch_steps = epoch_steps
        self.warmup = warmup
        self.max_norm = max_norm

    def rate(self, step=None):
        step = self._step if step is None else step
        if self.warmup:
            for k in (1, 2, 3):
                if step <= int(self.epoch_steps * k):
                    return self.lr_base * k / 4.0
        return self.lr_base

    def step(self):
        self._step += 1
        self._rate = self.rate()
        for g in self.optimizer.param_groups:
            g['lr'] = self._rate
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.step(max_norm=self.max_norm)
        else:
            self.optimizer.step()

    def zero_grad(self):
        self.optimizer.zero_grad()

    def decay(self, decay_r):
        self.lr_base *= decay_r

    def set_start_step(self, step):
        self._step = step

"""MixedOp: one supernet node holding every candidate operator plus its architecture logits
(mmnas/model/mixed.py:36-208).  Same public surface -- class attribute ``MODE``, ``candidate_ops``,
``alpha_prob``, ``alpha_gate``, ``active_index``, ``inactive_index``, ``n_choices``, ``binarize``,
``set_arch_param_grad``, ``rescale_updated_arch_param``, ``set_chosen_op_active``,
``probs_over_ops``, ``active_op``, ``chosen_index`` -- but built for a device-resident loop:

* sampling draws from a dedicated CPU generator (identical on every data-parallel rank, immune to
  dropout RNG use; SURVEY 2.2 rank-consistency note) and `Net_Search.reset_binary_gates` batches all
  30 nodes into one device->host copy instead of 30 `.item()` syncs (mixed.py:138,151);
* the alpha-gradient (mixed.py:171-198) is the closed form g*p - p*sum(g*p) on device tensors
  instead of an O(n^2) Python loop of scalar reads;
* candidates that only contribute a detached output (mixed.py:65-68) run under no_grad.
The dead 'full_v2' branch (mixed.py:70-101, guarded by an assert) is not carried over.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils.ops_adapter import OpsAdapter

OPS_ADAPTER = OpsAdapter()

_sampler = torch.Generator(device='cpu')
_sampler.manual_seed(888)  # the reference's search seed (search_vqa.py:62)


def seed_arch_sampler(seed):
    """Seed the architecture sampler; call with the same value on every rank."""
    _sampler.manual_seed(int(seed))


def sample_indices(probs_cpu, mode):
    """probs_cpu: [n] CPU tensor.  Returns (active, inactive) index lists (mixed.py:136-158)."""
    n = probs_cpu.numel()
    if mode == 'two':
        pair = torch.multinomial(probs_cpu, 2, replacement=False, generator=_sampler)
        sl = torch.softmax(torch.log(probs_cpu[pair]), dim=0)  # == softmax(alpha[pair])
        c = int(torch.multinomial(sl, 1, generator=_sampler)[0])
        return [int(pair[c])], [int(pair[1 - c])]
    a = int(torch.multinomial(probs_cpu, 1, generator=_sampler)[0])
    return [a], [i for i in range(n) if i != a]


def sample_rows(probs_cpu):
    """probs_cpu: [n, width] CPU tensor, one distribution per row -> list of n sampled column indices (MODE None)."""
    return torch.multinomial(probs_cpu, 1, generator=_sampler).view(-1).tolist()


class MixedOp(nn.Module):
    MODE = None  # None | 'full' | 'two'

    def __init__(self, __C, name):
        super().__init__()
        self.Used_OPS = OPS_ADAPTER.Used_OPS[name] if name in OPS_ADAPTER.Used_OPS else [name]
        self.n_choices = len(self.Used_OPS)
        self.candidate_ops = nn.ModuleList(
            [OPS_ADAPTER.OPS[n](__C, norm=__C.OPS_NORM, residual=__C.OPS_RESIDUAL) for n in self.Used_OPS])
        self.alpha_prob = nn.Parameter(torch.zeros(self.n_choices))
        self.alpha_gate = nn.Parameter(torch.zeros(self.n_choices))
        self.active_index = None
        self.inactive_index = None
        self._two_snapshot = None
        self._cand_params = None
        self.alpha_version = 0

    def forward(self, s, pre=None, s_mask=None, pre_mask=None, rel_embed=None):
        if MixedOp.MODE in ('full', 'two'):
            outs = [None] * self.n_choices
            a = self.active_index[0]
            outs[a] = self.candidate_ops[a](s, pre, s_mask, pre_mask, rel_embed)
            with torch.no_grad():   # these only feed the gate gradient (`.detach()` in mixed.py:67)
                for i in self.inactive_index:
                    outs[i] = self.candidate_ops[i](s, pre, s_mask, pre_mask, rel_embed)
            if outs[a].is_cuda and len(self.active_index) == 1 and outs[a].numel() % 4 == 0:
                from .. import ops   # one kernel: sum_j gate[j] * o_j; its backward gives every gate <dout, o_j>
                return ops.mixed_sum(self.alpha_gate, outs, a)
            out = 0
            for i in self.active_index:
                out = out + self.alpha_gate[i] * outs[i]
            for i in self.inactive_index:
                out = out + self.alpha_gate[i] * outs[i]
            return out
        return self.active_op(s, pre, s_mask, pre_mask, rel_embed)

    @property
    def probs_over_ops(self):
        return F.softmax(self.alpha_prob, dim=0)

    @property
    def active_op(self):
        return self.candidate_ops[self.active_index[0]]

    @property
    def chosen_index(self):
        probs = self.probs_over_ops.data.cpu().numpy()
        index = int(np.argmax(probs))
        return index, probs[index]

    def set_chosen_op_active(self):
        chosen, _ = self.chosen_index
        self.set_active([chosen], [i for i in range(self.n_choices) if i != chosen], write_gate=False)

    def set_active(self, active, inactive, write_gate=True):
        """Install a sampled (or injected) choice: index lists and, optionally, the binary gate."""
        self.active_index, self.inactive_index = list(active), list(inactive)
        self._two_snapshot = None
        if write_gate:
            g = torch.zeros(self.n_choices)
            g[self.active_index[0]] = 1.0
            self.alpha_gate.data.copy_(g)

    def binarize(self):
        """Sample this node's operator from softmax(alpha_prob) and reset its gate (mixed.py:131-163)."""
        probs = self.probs_over_ops.data.float().cpu()
        act, inact = sample_indices(probs, MixedOp.MODE)
        self.set_active(act, inact)
        self.clear_candidate_grads()

    def candidate_parameters(self):
        """All candidates' parameters (cached: walking the module tree 30 times per step cost 2 ms of host time;
        unused_modules_off() only swaps entries of candidate_ops for None, the parameters stay the same objects)."""
        if self._cand_params is None:
            self._cand_params = [p for op in self.candidate_ops if op is not None for p in op.parameters()]
            assert all(op is not None for op in self.candidate_ops), 'candidate_parameters() before unused_modules_back()'
        return self._cand_params

    def clear_candidate_grads(self):
        # "avoid over-regularization" (mixed.py:160-163): unsampled candidates must not see Adam momentum
        for p in self.candidate_parameters():
            p.grad = None

    def set_arch_param_grad(self):
        """dL/dalpha from dL/dgate (mixed.py:171-198)."""
        g = self.alpha_gate.grad.data
        if self.alpha_prob.grad is None:
            self.alpha_prob.grad = torch.zeros_like(self.alpha_prob.data)
        if MixedOp.MODE == 'two':
            idx = torch.as_tensor(self.active_index + self.inactive_index, device=g.device)
            a = self.alpha_prob.data[idx]
            p = torch.softmax(a, dim=0)
            gp = g[idx] * p
            self.alpha_prob.grad.data[idx] += gp - p * gp.sum()
            self._two_snapshot = (idx, a.clone())
        else:
            p = torch.softmax(self.alpha_prob.data, dim=0)
            gp = g * p
            self.alpha_prob.grad.data += gp - p * gp.sum()

    def rescale_updated_arch_param(self):
        """'two' mode: keep the sampled pair's total probability mass (mixed.py:200-208)."""
        idx, old = self._two_snapshot
        new = self.alpha_prob.data[idx]
        offset = torch.logsumexp(new, 0) - torch.logsumexp(old, 0)
        self.alpha_prob.data[idx] -= offset
        self.alpha_version += 1   # (.data writes do not bump the tensor version the sampling cache keys on)

"""The reference loops' `0 * sum(p.sum() for p in ...)` lines from the library's side.

search_vqa.py:285-288 / :322-324 and train_vqa.py:299 add

    loss += 0 * sum(p.sum() for p in net.module.net_parameters())          (and the two alpha sets)

to give EVERY parameter a (zero) gradient: stock DDP then finds no unused parameter, and torch Adam steps the unsampled
candidates too (their moments decay).  Taken literally that is, per step of the d = 256 supernet, ~900 `sum` launches,
~900 scalar adds behind Python's `sum()`, ~2100 autograd nodes and -- in backward -- ~900 expanded zero gradients of
which ~640 are cloned into `.grad`: 27 of the 34 ms an unchanged script needs per step (profiles/r04_host_dropin.txt),
around a 4.7 ms GPU step.

The scripts stay byte for byte; the PARAMETERS are the library's.  Every parameter of a Net_* is re-classed to
`SumParameter` (an nn.Parameter whose only change is the argument-less `.sum()`), which returns a `LazySum` -- a lazy
expression "scale * (sum of the parameters' sums)" that supports what the scripts do with it (`+` between themselves and
with Python's `sum()` start value 0, `*` by a number) and turns into a real tensor the moment anything else is asked
of it (`float()`, `.item()`, arithmetic with a tensor when the scale is not zero, ...).  `loss + LazySum` with scale 0
is the case the scripts write: it attaches ONE autograd node (`_ZeroGrads`) that passes the loss through and, in backward,
hands every parameter of the expression a zero gradient as a fresh view of one zero-filled flat buffer (one fill
launch, views made in one C++ call).  Autograd accumulates those through the ordinary AccumulateGrad nodes, so stock DDP
sees every parameter ready exactly as with the literal lines, torch Adam steps every parameter, and a sampled
parameter's real gradient is added to its zero.

Differences from the literal arithmetic, both outside what a healthy run can observe: a non-finite PARAMETER no longer
turns the loss into NaN through `0 * inf` (the loss itself still does), and the zero gradients are exact zeros even
when the incoming gradient of the loss is non-finite.  `MMNAS_ZERO_TERMS=0` keeps plain parameters (the literal lines
then run as written).
"""
import os

import torch
from torch import nn


def enabled():
    return os.environ.get('MMNAS_ZERO_TERMS', '1') != '0'


class _ZeroGrads(torch.autograd.Function):
    """loss -> loss (same value); backward: the incoming gradient for the loss, an exact zero for every parameter."""

    @staticmethod
    def forward(ctx, loss, *params):
        ctx.like = params
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        params = ctx.like
        by_key = {}
        for i, p in enumerate(params):          # (one flat buffer per device / dtype: in practice one)
            by_key.setdefault((p.device, p.dtype), []).append(i)
        out = [None] * len(params)
        for (dev, dt), idx in by_key.items():
            group = [params[i] for i in idx]
            flat = torch.zeros(sum(p.numel() for p in group), device=dev, dtype=dt)
            # fresh view objects (use count 1): AccumulateGrad adopts them as `.grad` without a copy
            for i, v in zip(idx, torch._utils._unflatten_dense_tensors(flat, group)):
                out[i] = v
        return (g, *out)


class LazySum:
    """scale * sum_i sum(p_i), unevaluated.  `parts` is a binary tree of parameter lists (O(1) per `+`)."""
    __slots__ = ('parts', 'scale')
    __array_priority__ = 1000     # (numpy scalars on the left defer to __r*__)

    def __init__(self, parts, scale=1.0):
        self.parts, self.scale = parts, scale

    # -- the expression ------------------------------------------------------------------------------------------------
    def params(self):
        out, stack = [], [self.parts]
        while stack:
            n = stack.pop()
            if isinstance(n, tuple):
                stack.append(n[1])
                stack.append(n[0])
            else:
                out.append(n)
        return out

    def materialize(self):
        ps = self.params()
        total = torch.Tensor.sum(ps[0])
        for p in ps[1:]:
            total = total + torch.Tensor.sum(p)
        return total if self.scale == 1.0 else total * self.scale

    # -- what the scripts do with it -----------------------------------------------------------------------------------
    def __add__(self, other):
        if isinstance(other, LazySum):
            if other.scale == self.scale:
                return LazySum((self.parts, other.parts), self.scale)
            return self.materialize() + other.materialize()
        if isinstance(other, (int, float)) and not isinstance(other, bool) and other == 0:
            return self                                  # Python's sum() starts from 0
        if isinstance(other, torch.Tensor):
            return self.__radd__(other)
        return self.materialize() + other

    def __radd__(self, other):
        if isinstance(other, (int, float)) and not isinstance(other, bool) and other == 0:
            return self
        if isinstance(other, torch.Tensor) and self.scale == 0:
            ps = [p for p in self.params() if p.requires_grad]
            if not ps or not torch.is_grad_enabled():
                return other.clone()                     # (0 * finite == 0: nothing to add, nothing to differentiate)
            return _ZeroGrads.apply(other, *ps)
        return other + self.materialize()

    def __mul__(self, k):
        if isinstance(k, (int, float)) and not isinstance(k, bool):
            return LazySum(self.parts, self.scale * k)
        return self.materialize() * k

    __rmul__ = __mul__

    def __neg__(self):
        return LazySum(self.parts, -self.scale)

    def __sub__(self, other):
        return self.materialize() - (other.materialize() if isinstance(other, LazySum) else other)

    def __rsub__(self, other):
        return other - self.materialize()

    def __truediv__(self, k):
        if isinstance(k, (int, float)) and not isinstance(k, bool):
            return LazySum(self.parts, self.scale / k)
        return self.materialize() / k

    # -- everything else: the real tensor ------------------------------------------------------------------------------
    # (comparisons, powers, abs, truth value, int(): Python looks dunders up on the TYPE, so __getattr__ below never sees
    #  them -- `p.sum() > 0`, `p.sum() ** 2`, `abs(p.sum())`, `bool(p.sum())` raised TypeError)
    def __lt__(self, o): return self.materialize() < (o.materialize() if isinstance(o, LazySum) else o)
    def __le__(self, o): return self.materialize() <= (o.materialize() if isinstance(o, LazySum) else o)
    def __gt__(self, o): return self.materialize() > (o.materialize() if isinstance(o, LazySum) else o)
    def __ge__(self, o): return self.materialize() >= (o.materialize() if isinstance(o, LazySum) else o)
    def __eq__(self, o): return self.materialize() == (o.materialize() if isinstance(o, LazySum) else o)
    def __ne__(self, o): return self.materialize() != (o.materialize() if isinstance(o, LazySum) else o)
    __hash__ = object.__hash__
    def __pow__(self, k): return self.materialize() ** (k.materialize() if isinstance(k, LazySum) else k)
    def __rpow__(self, k): return k ** self.materialize()
    def __abs__(self): return abs(self.materialize())
    def __pos__(self): return self
    def __rtruediv__(self, o): return o / self.materialize()
    def __bool__(self): return bool(self.materialize())
    def __int__(self): return int(self.materialize().detach())

    def __float__(self):
        return float(self.materialize().detach())

    def item(self):
        return self.materialize().item()

    def __getattr__(self, name):          # .backward(), .detach(), .shape, ... : whatever a 0-d tensor offers
        if name.startswith('__'):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __repr__(self):
        return 'LazySum(%d parameters, scale=%g)' % (len(self.params()), self.scale)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        # `loss += lazy` / `loss + lazy` / torch.add(loss, lazy) arrive here (Tensor.__iadd__ does not return NotImplemented
        # for a foreign operand, it consults __torch_function__): the scripts' case goes to __radd__, i.e. to the one node
        kwargs = kwargs or {}
        if getattr(func, '__name__', '') in ('__iadd__', 'add_', '__add__', 'add', '__radd__') and len(args) == 2 and not kwargs:
            a, b = args
            if isinstance(a, torch.Tensor) and isinstance(b, LazySum):
                return b.__radd__(a)
            if isinstance(b, torch.Tensor) and isinstance(a, LazySum):
                return a.__radd__(b)
        # any other torch function that met a LazySum among its arguments -- also inside lists / tuples / dicts
        # (torch.stack([p.sum() for p in ps]): a top-level-only conversion re-entered this hook forever):
        # evaluate and carry on with tensors
        from torch.utils._pytree import tree_map
        conv = lambda a: a.materialize() if isinstance(a, LazySum) else a
        return func(*tree_map(conv, tuple(args)), **tree_map(conv, dict(kwargs)))


class SumParameter(nn.Parameter):
    """nn.Parameter whose argument-less `.sum()` is lazy (see the module docstring).  Nothing else changes: it is a
    Parameter to isinstance(), to optimizers, to DDP, to state_dict() and to deepcopy."""

    def sum(self, *args, **kwargs):
        if args or kwargs:
            return super().sum(*args, **kwargs)
        return LazySum(self)


def _register_with_torch_optim():
    """torch.optim picks its multi-tensor (`foreach`) implementations only when every parameter's EXACT type is in
    torch.optim.optimizer._foreach_supported_types ([Tensor, Parameter]); any other subclass silently gets the one-tensor-
    at-a-time loop -- for the supernet's ~900 parameters 6 launches each, 24 ms of host time per Adam step (measured: the
    first build of this module made the unchanged script SLOWER for exactly that reason).  SumParameter is a Parameter in
    everything but `.sum()`, so it joins the list."""
    try:
        from torch.optim import optimizer as _opt
        if SumParameter not in _opt._foreach_supported_types:
            _opt._foreach_supported_types.append(SumParameter)
    except (ImportError, AttributeError):        # (another torch layout: the optimizers still work, on their slow path)
        pass


_register_with_torch_optim()


def adopt(module):
    """Re-class every nn.Parameter of `module` (exact type only: other subclasses keep theirs) to SumParameter."""
    if not enabled():
        return module
    for p in module.parameters():
        if type(p) is nn.Parameter:
            p.__class__ = SumParameter
    return module

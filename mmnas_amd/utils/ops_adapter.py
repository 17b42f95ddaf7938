"""Operator registry and search spaces: the plugin surface arch/*.json names resolve through
(mmnas/utils/ops_adapter.py:5-74).  ``OpsAdapter().OPS[name](__C, norm, residual)`` returns an
nn.Module with the 5-argument cell signature; ``Used_OPS`` lists the candidates of a MixedOp.
"""
from ..model import modules as M

_HEAD_DIMS = (256, 128, 64, 32, 16)


def _att(cls, base, hsize_k=None):
    return lambda __C, norm, residual: cls(__C, norm, residual, base=base, hsize_k=hsize_k)


def _build_registry():
    ops = {
        'none': lambda __C, norm, residual: M.Zero(),
        'skip_connect': lambda __C, norm, residual: M.Identity(),
        'relu': lambda __C, norm, residual: M.ReLU(),
        'gelu': lambda __C, norm, residual: M.GELU(),
        'leakyrelu': lambda __C, norm, residual: M.LeakyReLU(),
    }
    for b in _HEAD_DIMS:
        ops['self_att_%d' % b] = _att(M.SelfAtt, b)
        ops['rel_self_att_%d' % b] = _att(M.RelSelfAtt, b)
        ops['guided_att_%d' % b] = _att(M.GuidedAtt, b)
    ops['self_att_64_2'] = _att(M.SelfAtt, 64, 2)
    ops['guided_att_64_2'] = _att(M.GuidedAtt, 64, 2)
    for b in (128, 64, 32):
        ops['uniimg_att_%d' % b] = _att(M.UniimgAtt, b)
    for k in (3, 5, 7, 11):
        ops['sep_conv_%d' % k] = (lambda kk: lambda __C, norm, residual: M.SepConv(__C, norm, residual, k=kk))(k)
        ops['std_conv_%d' % k] = (lambda kk: lambda __C, norm, residual: M.StdConv(__C, norm, residual, k=kk))(k)
    ops['feed_forward'] = lambda __C, norm, residual: M.FeedForward(__C, norm, residual)
    for mk in (2, 8, 16, 32):
        ops['feed_forward_%d' % mk] = (lambda m: lambda __C, norm, residual: M.FeedForward(__C, norm, residual, mid_k=m))(mk)
    ops['gated_linear_1'] = lambda __C, norm, residual: M.GLU(__C, norm, residual, layers=1)
    ops['gated_linear_2'] = lambda __C, norm, residual: M.GLU(__C, norm, residual, layers=2)
    ops['feed_forward_deep'] = lambda __C, norm, residual: M.FeedForward_deep(__C, norm, residual)
    return ops


class OpsAdapter:
    def __init__(self):
        self.Used_OPS = {
            'enc_safe': ['self_att_64', 'feed_forward'],
            'dec_safe': ['self_att_64', 'rel_self_att_64', 'guided_att_64', 'feed_forward'],
        }
        self.Used_OPS['enc'] = self.Used_OPS['enc_safe'] + ['none']
        self.Used_OPS['dec'] = self.Used_OPS['dec_safe'] + ['none']
        self.OPS = _build_registry()

"""Run the CPU oracle on a cases.py case and return outputs + gradients (numpy)."""
import numpy as np
import torch

from oracle import mmnas_oracle as O

T = torch.from_numpy


def run_oracle_op(case, drops=None, dtype=torch.float32):
    cfg = case['cfg']
    P = {k: T(v).to(dtype).requires_grad_(True) for k, v in case['P'].items()}
    x = T(case['x']).to(dtype).requires_grad_(True)
    y = T(case['y']).to(dtype).requires_grad_(True)
    rel = T(case['rel']).to(dtype).requires_grad_(True)
    if drops is not None:
        drops = {k: (T(np.ascontiguousarray(v)).to(dtype) if v is not None else None) for k, v in drops.items()}
    out = O.op_forward(case['name'], P, cfg, x, y, T(case['x_mask']), T(case['y_mask']), rel,
                       norm=cfg.OPS_NORM, residual=cfg.OPS_RESIDUAL, drops=drops)
    res = {'out': out.detach().float().numpy()}
    if out.requires_grad:
        (out * T(case['gout']).to(dtype)).sum().backward()
    res['dx'] = x.grad.float().numpy() if x.grad is not None else np.zeros_like(case['x'])
    if y.grad is not None:
        res['dy'] = y.grad.float().numpy()
    if rel.grad is not None:
        res['drel'] = rel.grad.float().numpy()
    for k, p in P.items():
        res['g:' + k] = p.grad.float().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    return res


def net_loss(task, pred, target):
    """The stand-in losses used for the net goldens (make_golden._net_loss)."""
    t = T(target)
    if task == 'vqa':
        return O.bce_with_logits_sum(pred, t)
    if task == 'itm':
        return torch.nn.functional.binary_cross_entropy(pred, t, reduction='sum')
    scores, reg = pred
    return (scores * t).sum() + 0.5 * (reg ** 2).sum()


def run_oracle_net(c, search=None):
    P = {k: T(v).clone().requires_grad_(True) for k, v in c['P'].items()}
    inputs = tuple(T(a) for a in c['inputs'])
    pred = O.net_forward(c['task'], P, c['cfg'], inputs, genotype=c['genotype'], search=search)
    loss = net_loss(c['task'], pred, c['target'])
    loss.backward()
    grads = {k: (p.grad.numpy() if p.grad is not None else None) for k, p in P.items()}
    return pred, float(loss.detach()), grads

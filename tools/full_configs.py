"""Full-size sanity of the other configs (BASELINE C4 mmnas_vgd B=64 S_x=15; C5 mmnas_itm B=160 S_y=36 S_x=50; C1 mcan
B=4 S_y=36): forward + backward run, outputs and gradients finite, timing."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.golden import cases
T = torch.from_numpy
for task, arch, kw in (('vgd', 'mmnas_vgd', dict(B=64, Sx=15, Sy=100)), ('itm', 'mmnas_itm', dict(B=160, Sx=50, Sy=36)),
                       ('vqa', 'mcan', dict(B=4, Sx=14, Sy=36)), ('vqa', 'mmnas_vqa', dict(B=64, Sx=14, Sy=100))):
    c = cases.net_case(task, arch, 7, HSIZE=512, token_size=2000, ans_size=3129, **kw)
    Net = importlib.import_module('mmnas.model.full_%s' % task).Net_Full
    init = {'token_size': c['token_size'], 'ans_size': c['ans_size'],
            'pretrained_emb': np.zeros((c['token_size'], c['cfg'].WORD_EMBED_SIZE), np.float32)}
    net = Net(c['cfg'], init)
    net.load_state_dict({k: T(v) for k, v in c['P'].items()})
    net = net.cuda().train()
    inp = tuple(T(a).cuda() for a in c['inputs'])
    def step():
        out = net(inp)
        loss = sum(o.float().pow(2).mean() for o in out) if isinstance(out, tuple) else out.float().pow(2).mean()
        net.zero_grad(); loss.backward()
        return out, loss
    for _ in range(3): out, loss = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out, loss = step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 100
    outs = out if isinstance(out, tuple) else (out,)
    ok = all(torch.isfinite(o).all() for o in outs) and all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    print('%s %-10s %s: finite=%s  %.2f ms/step  loss %.4g' % (task, arch, kw, bool(ok), ms, float(loss)))

"""Where the host time of the UNCHANGED search loop goes (search_vqa.py:279-301 on the per-operator path, N = 1, no DDP
wrapper): the script's own statements against what this library adds per step.  Wall clock per step with the queue
drained between steps (the loop is host-bound: ~6 ms of GPU work under 20+ ms of host issue).

    python tools/dropin_host_split.py > profiles/r04_host_dropin.txt

Variants (each the median of 30 steps after 10 warm-up steps):
  full        every statement of search_vqa.py:279-301
  no_zero_sum without the three `loss += 0 * sum(p.sum() for p in ...)` lines (:285-288)
  fwd_bwd     sample, unused_modules_off, forward, loss, backward, unused_modules_back -- no zero_grad, clip, Adam
  forward     sample, unused_modules_off, forward, loss, unused_modules_back (no backward)
  harness     the same arithmetic through mmnas_amd.harness.SearchLoop.weight_step (flat gradients, fused clip + Adam)
"""
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    from mmnas.model.hygr_vqa import Net_Search
    from mmnas.model.mixed import MixedOp
    from mmnas.utils.optimizer import WarmupOptimizer
    from mmnas_amd import ops
    from mmnas_amd.harness import SearchLoop
    from mmnas_amd.model import mixed
    dev = torch.device('cuda', 0)
    torch.manual_seed(888)
    ops.manual_seed(888)
    mixed.seed_arch_sampler(888)
    cfg = bench.make_cfg('search')
    emb = torch.randn(bench.VOCAB, 300, generator=torch.Generator().manual_seed(1)).numpy()
    init = {'token_size': bench.VOCAB, 'ans_size': bench.ANS, 'pretrained_emb': emb}
    cpu_in, cpu_tg = bench.synth_batch(cfg, 64, bench.SX, bench.SY, bench.VOCAB, bench.ANS, 888)
    inp, tgt = tuple(t.to(dev) for t in cpu_in), cpu_tg.to(dev)
    net = Net_Search(cfg, init).to(dev).train()
    opt = WarmupOptimizer(4e-4, torch.optim.Adam(net.net_parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9), epoch_steps=1000, warmup=True)
    loss_fn = torch.nn.BCEWithLogitsLoss(reduction='sum')

    noop_logits = torch.zeros(64, bench.ANS, device=dev, requires_grad=True)

    def step(zero_sum=True, backward=True, optimize=True, noop_forward=False):
        MixedOp.MODE = None
        net.reset_binary_gates()
        net.unused_modules_off()
        # noop_forward (--floor): the script's statements around a forward that does NOTHING -- what is left is torch's own
        # share of the unchanged loop (699 parameters: the zero-term node, zero_grad, clip_grad_norm_, Adam.step) plus the
        # sampling / module switching of this library's MixedOp mirror
        pred = noop_logits * 1.0 if noop_forward else net(inp)
        loss = loss_fn(pred, tgt)
        if zero_sum:
            loss += 0 * sum(p.sum() for p in net.alpha_prob_parameters())
            loss += 0 * sum(p.sum() for p in net.alpha_gate_parameters())
            loss += 0 * sum(p.sum() for p in net.net_parameters())
        if optimize:
            net.zero_grad()
        if backward:
            loss.backward()
        if optimize:
            torch.nn.utils.clip_grad_norm_(net.net_parameters(), 1.0)
            opt.step()
        net.unused_modules_back()
        return loss

    def step_statements(acc):
        """the full step with a host timer behind every statement (no synchronisation in between: issue time only)"""
        t = [time.perf_counter()]
        tick = lambda: t.append(time.perf_counter())
        MixedOp.MODE = None
        net.reset_binary_gates()
        net.unused_modules_off(); tick()
        pred = net(inp); tick()
        loss = loss_fn(pred, tgt); tick()
        e0 = 0 * sum(p.sum() for p in net.alpha_prob_parameters())
        e1 = 0 * sum(p.sum() for p in net.alpha_gate_parameters())
        e2 = 0 * sum(p.sum() for p in net.net_parameters()); tick()
        loss += e0
        loss += e1
        loss += e2; tick()
        net.zero_grad(); tick()
        loss.backward(); tick()
        torch.nn.utils.clip_grad_norm_(net.net_parameters(), 1.0); tick()
        opt.step(); tick()
        net.unused_modules_back(); tick()
        for i, k in enumerate(('sample + unused_modules_off', 'forward', 'loss', 'build 3 x `0 * sum(p.sum())`', '3 x `loss +=`', 'zero_grad',
                               'backward', 'clip_grad_norm_', 'opt.step', 'unused_modules_back')):
            acc.setdefault(k, []).append(1e3 * (t[i + 1] - t[i]))

    def measure(fn, n=30, warm=10):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            ts.append((1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)))
        return statistics.median(t[0] for t in ts), statistics.median(t[1] for t in ts)

    print('# host issue ms / wall ms per step (queue drained before every step), B = 64, HSIZE 256, %d parameter tensors (%d of them alpha)'
          % (len(list(net.parameters())), 2 * len(net.redundant_modules)))
    res = {}
    if '--floor' in sys.argv:
        # the floor under `search_vqa_dropin` -- every statement of search_vqa.py:279-301 with the forward
        # replaced by a no-op (so backward has only the zero-term node and the loss to run)
        res['full'] = measure(lambda: step())
        res['floor'] = measure(lambda: step(noop_forward=True))
        res['floor_no_zero_sum'] = measure(lambda: step(noop_forward=True, zero_sum=False))
        print('%-18s host %6.2f ms   wall %6.2f ms   (every statement, real forward)' % ('full', *res['full']))
        print('%-18s host %6.2f ms   wall %6.2f ms   (every statement, no-op forward: torch\'s share + sampling)' % ('floor', *res['floor']))
        print('%-18s host %6.2f ms   wall %6.2f ms   (the same without the three `0 * sum(p.sum())` lines)' % ('floor_no_zero_sum', *res['floor_no_zero_sum']))
        print('# search_vqa_dropin - floor = %.2f ms: the per-operator path\'s own host time (forward + backward nodes of ~30 operators)'
              % (res['full'][1] - res['floor'][1]))
        return
    for name, kw in (('full', {}), ('no_zero_sum', dict(zero_sum=False)), ('fwd_bwd', dict(zero_sum=False, optimize=False)),
                     ('forward', dict(zero_sum=False, optimize=False, backward=False))):
        if name in ('fwd_bwd',):
            net.zero_grad()
        res[name] = measure(lambda: step(**kw))
        print('%-12s host %6.2f ms   wall %6.2f ms' % (name, *res[name]))
    if '--cprofile' in sys.argv:      # where the per-operator path's own host time goes (forward + backward, no optimizer)
        import cProfile
        import pstats
        net.zero_grad()
        for _ in range(10):
            step(zero_sum=False, optimize=False)
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            step(zero_sum=False, optimize=False)
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr, stream=sys.stdout)
        st.sort_stats('tottime').print_stats(45)
        return
    if '--statements' in sys.argv:
        print()
        print('# the full step statement by statement (host issue ms, median of 30 steps, queue drained before every step)')
        for zs in (True, False):
            acc = {}
            for i in range(40):
                torch.cuda.synchronize()
                if zs:
                    step_statements(acc)
                else:      # the same statements without the three lines: what clip / Adam / zero_grad cost on the sampled parameters only
                    t0 = time.perf_counter()
                    MixedOp.MODE = None
                    net.reset_binary_gates()
                    net.unused_modules_off()
                    loss = loss_fn(net(inp), tgt)
                    t1 = time.perf_counter(); net.zero_grad()
                    t2 = time.perf_counter(); loss.backward()
                    t3 = time.perf_counter(); torch.nn.utils.clip_grad_norm_(net.net_parameters(), 1.0)
                    t4 = time.perf_counter(); opt.step()
                    t5 = time.perf_counter(); net.unused_modules_back()
                    for k, v in (('forward + loss', t1 - t0), ('zero_grad', t2 - t1), ('backward', t3 - t2), ('clip_grad_norm_', t4 - t3), ('opt.step', t5 - t4)):
                        acc.setdefault(k, []).append(1e3 * v)
            print('#   %s' % ('with the three lines' if zs else 'without them'))
            for k, v in acc.items():
                print('#     %-34s %6.2f ms' % (k, statistics.median(v[10:])))
    net2 = Net_Search(cfg, init).to(dev).train()
    loop = SearchLoop(net2, loss_fn, net_lr=4e-4, clip=1.0, epoch_steps=1000, warmup=True)
    res['harness'] = measure(lambda: loop.weight_step(inp, tgt))
    print('%-12s host %6.2f ms   wall %6.2f ms' % ('harness', *res['harness']))
    f, nz, fb, fw = (res[k][1] for k in ('full', 'no_zero_sum', 'fwd_bwd', 'forward'))
    print()
    print('# split of the unchanged loop (wall %.2f ms):' % f)
    print('#   (i) the script\'s own statements')
    print('#       three `0 * sum(p.sum())` lines, forward and backward side        %6.2f ms' % (f - nz))
    print('#       net.zero_grad() + clip_grad_norm_ + WarmupOptimizer/Adam.step()   %6.2f ms' % (nz - fb))
    print('#   (ii) this library: sample + unused_modules_off/back + forward          %6.2f ms' % fw)
    print('#        backward of the operators (autograd.Function nodes)              %6.2f ms' % (fb - fw))
    print('#   the same arithmetic through SearchLoop.weight_step                    %6.2f ms' % res['harness'][1])


if __name__ == '__main__':
    main()

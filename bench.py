"""Benchmark of the candidate-operator hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload train_vqa|search_vqa]

A "step" is one forward + loss + backward of the network over one synthetic batch of 64 samples per
GPU:
    train_vqa  (default; BASELINE.json configs[1]): Net_Full(arch/mmnas_vqa.json), HSIZE 512,
               S_y = 100 regions x 2048, S_x = 14 tokens, dropout 0.1, BCE(sum)
               (train_vqa.py:295-299)
    search_vqa (configs[2], weight step): Net_Search supernet, HSIZE 256: sample -> fwd -> bwd
               (search_vqa.py:279-292)
For N > 1 (launched by torch.distributed.run, one rank per GPU) every rank runs its own batch
(weak scaling) and the parameter gradients are averaged with RCCL all-reduce inside the step.
Rank 0 prints ONE JSON line.  `roofline` is the fp32-MFMA GEMM kernel class (the dominant kernel:
~65 % of device time), measured with HIP events on the launch stream (mmnas_prof_*: start/stop events
carried by each kernel's dispatch) over a repeat of the timed steps right after the timed region --
the events cost ~9 % of a step, so they stay out of the throughput measurement; `cpu_baseline` is the CPU oracle (a port of the reference step) timed on this
box's host cores on the same workload (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0


def pmc_traffic(workload):
    """HBM-side bytes per GEMM launch from the committed PMC passes (profiles/r01_traffic_<workload>.json, made by
    tools/pmc_traffic.py from two `rocprofv3 --pmc` runs of this benchmark: FETCH_SIZE and WRITE_SIZE cannot share a
    pass; traffic = 2*FETCH + WRITE, the gfx950 correction of MI355X_MICROARCH.md).  None when the file is absent."""
    path = os.path.join(REPO, 'profiles', 'r01_traffic_%s.json' % workload)
    if not os.path.exists(path):
        return None, None
    d = json.load(open(path))
    n = tot = fetch = write = 0.0
    for k, v in d.items():
        if 'gemm_kernel' in k or 'gemm_pair_kernel' in k:
            n += v['launches']
            tot += v['launches'] * v['traffic_bytes_per_launch']
            fetch += v['launches'] * v['fetch_kb_per_launch'] * 1024.0
            write += v['launches'] * v['write_kb_per_launch'] * 1024.0
    if n == 0:
        return None, None
    return tot / n, {'source': os.path.relpath(path, REPO), 'fetch_size_bytes_per_launch': fetch / n,
                     'write_size_bytes_per_launch': write / n, 'formula': '2*FETCH_SIZE + WRITE_SIZE (gfx950)'}


def make_cfg(workload):
    from types import SimpleNamespace
    c = dict(DROPOUT_R=0.1, REL_SIZE=64, OPS_NORM=True, OPS_RESIDUAL=True, LAYERS=1,
             NODES={'enc': 12, 'dec': 18}, ATTFLAT_GLIMPSES=1, ATTFLAT_MLP_SIZE=512, FRCNFEAT_SIZE=2048,
             BBOX_FEATURE=False, BBOXFEAT_EMB_SIZE=1024, WORD_EMBED_SIZE=300, ALPHA_INIT_TYPE='normal',
             SCORES_LOSS='kld', GENOTYPE=None)
    if workload == 'train_vqa':      # train_vqa.py:136-154
        c.update(HSIZE=512, ATTFLAT_OUT_SIZE=1024)
        with open(os.path.join(REPO, 'arch', 'mmnas_vqa.json')) as f:
            g = json.load(f)
        c['GENOTYPE'] = g[sorted(g)[-1]]
    else:                            # search_vqa.py:87-114
        c.update(HSIZE=256, ATTFLAT_OUT_SIZE=512)
    return SimpleNamespace(**c)


def synth_batch(cfg, B, Sx, Sy, V, ans, seed):
    """Synthetic batch with the loaders' tensor contract: zero rows = padding."""
    g = torch.Generator().manual_seed(seed)
    frcn = torch.relu(torch.randn(B, Sy, cfg.FRCNFEAT_SIZE, generator=g))
    y_rel = torch.randn(B, Sy, Sy, 4, generator=g)
    ques = torch.randint(1, V, (B, Sx), generator=g)
    x_rel = torch.randn(B, Sx, Sx, 3, generator=g)
    ny = torch.randint(10, Sy + 1, (B,), generator=g)
    nx = torch.randint(3, Sx + 1, (B,), generator=g)
    for b in range(B):
        frcn[b, ny[b]:] = 0
        y_rel[b, ny[b]:] = 0
        y_rel[b, :, ny[b]:] = 0
        ques[b, nx[b]:] = 0
        x_rel[b, nx[b]:] = 0
        x_rel[b, :, nx[b]:] = 0
    bbox = torch.zeros(B, Sy, 5)
    target = torch.rand(B, ans, generator=g) * (torch.rand(B, ans, generator=g) < 0.003)
    return (frcn, bbox, y_rel, ques, x_rel), target


def op_flops_fwd(name, B, Sx, Sy, d, kind):
    """Algorithmic forward flops of one cell operator (SURVEY 8d formulas)."""
    S = Sx if kind == 'enc' else Sy
    if 'guided' in name:
        return 4 * B * S * d * d + 4 * B * Sx * d * d + 4 * B * S * Sx * d
    if 'rel_self_att' in name:
        return 8 * B * S * d * d + 4 * B * S * S * d + 2 * B * S * S * 64 * (d // 64)
    if 'self_att' in name:
        return 8 * B * S * d * d + 4 * B * S * S * d
    if name == 'feed_forward':
        return 16 * B * S * d * d
    return 0


def step_flops(cfg, names_enc, names_dec, B, Sx, Sy, ans):
    d = cfg.HSIZE
    f = sum(op_flops_fwd(n, B, Sx, Sy, d, 'enc') for n in names_enc)
    f += sum(op_flops_fwd(n, B, Sx, Sy, d, 'dec') for n in names_dec)
    stem = 2 * B * Sy * cfg.FRCNFEAT_SIZE * d + 2 * B * Sx * 4 * d * (cfg.WORD_EMBED_SIZE + d)
    stem += 2 * B * Sy * Sy * 4 * 64 + 2 * B * (Sx + Sy) * d * cfg.ATTFLAT_MLP_SIZE
    stem += 2 * 2 * B * d * cfg.ATTFLAT_OUT_SIZE + 2 * B * cfg.ATTFLAT_OUT_SIZE * ans
    return 3 * (f + stem)  # backward = 2 x forward


def cpu_baseline(cfg, net, workload, inputs, target, plan, budget_s=25.0):
    """Time the CPU oracle (port of the reference step) on the host cores: same weights, same batch."""
    from oracle import mmnas_oracle as O
    # torch's intra-op pool degrades badly when it has far more threads than these small GEMMs can
    # use (measured: 256 threads -> 306 s/step vs ~7 s/step with 8): cap it and say so in `cores`
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    P = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in net.state_dict().items()}
    search = None
    if workload == 'search_vqa':
        search = {'mode': None, 'enc': plan[:12], 'dec': plan[12:]}
    p = float(cfg.DROPOUT_R)
    B = target.shape[0]

    def one(nb):
        for v in P.values():
            v.grad = None
        pred = O.net_forward('vqa', P, cfg, tuple(t[:nb] for t in inputs), genotype=cfg.GENOTYPE, search=search,
                             drops_for=(lambda key: p))
        loss = O.bce_with_logits_sum(pred, target[:nb])
        loss.backward()
        return float(loss.detach())

    nb = min(8, B)
    one(nb)                                 # warm-up on a slice (allocator, thread pool)
    t0 = time.perf_counter()
    one(nb)
    t_slice = time.perf_counter() - t0
    est_full = t_slice * B / nb
    if est_full <= budget_s / 2:            # full batches fit the budget
        n, t_sum = 0, 0.0
        while n < 10 and t_sum + est_full <= budget_s:
            t0 = time.perf_counter()
            one(B)
            t_sum += time.perf_counter() - t0
            n += 1
        return {'value': n / t_sum, 'unit': 'steps/s', 'cores': threads, 'kind': 'port',
                'sample': '%d full steps (B=%d, same weights and batch, dropout 0.1 via torch RNG) of the CPU oracle; %.2f s/step'
                          % (n, B, t_sum / n)}
    nb2 = max(nb, min(B, int(B * (budget_s / 2) / est_full) // 4 * 4 or nb))
    t0 = time.perf_counter()
    one(nb2)
    t2 = time.perf_counter() - t0
    return {'value': (nb2 / B) / t2, 'unit': 'steps/s', 'cores': threads, 'kind': 'port',
            'sample': 'one step on the first %d of the %d samples of the batch (same weights, dropout 0.1 via torch RNG), '
                      'scaled by %d/%d; %.2f s for the slice' % (nb2, B, nb2, B, t2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='train_vqa', choices=['train_vqa', 'search_vqa'])
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--gemm-split', type=int, default=0, choices=[0, 3, 6],
                    help='EXPERIMENT, not the headline: run the GEMMs as 3 / 6 bf16-MFMA products of split operands '
                         '(MMNAS_GEMM_SPLIT); the JSON line then says so in dtype and config')
    ap.add_argument('--no-prof', action='store_true', help='do not bracket kernels with HIP events')
    ap.add_argument('--with-optim', action='store_true',
                    help='also run gradient clipping + the fused Adam step inside the timed step (not part of the fwd+bwd metric)')
    args = ap.parse_args()
    if args.gemm_split:
        os.environ['MMNAS_GEMM_SPLIT'] = str(args.gemm_split)   # read when the library first schedules a GEMM
    else:
        os.environ.pop('MMNAS_GEMM_SPLIT', None)                 # the headline line is always the fp32-MFMA path

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch.distributed as dist
    # (tests only: MMNAS_BENCH_DEVICE / MMNAS_BENCH_BACKEND=gloo run several ranks on ONE GPU -- RCCL refuses that --
    #  so the N > 1 code path can be exercised end to end on a single-GPU box; see tests/test_bench_gpu.py)
    dev_index = int(os.environ.get('MMNAS_BENCH_DEVICE', local_rank))
    backend = os.environ.get('MMNAS_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from mmnas_amd import _lib as L, dp, ops
    from mmnas_amd.model import mixed
    from mmnas.model.mixed import MixedOp
    lib = L.lib()

    B, Sx, Sy, V, ANS = args.batch, 14, 100, 20000, 3129
    cfg = make_cfg(args.workload)
    torch.manual_seed(888)
    ops.manual_seed(888 + rank)
    mixed.seed_arch_sampler(888)
    emb = torch.randn(V, cfg.WORD_EMBED_SIZE, generator=torch.Generator().manual_seed(1)).numpy()
    init = {'token_size': V, 'ans_size': ANS, 'pretrained_emb': emb}
    if args.workload == 'train_vqa':
        from mmnas.model.full_vqa import Net_Full
        net = Net_Full(cfg, init)
    else:
        from mmnas.model.hygr_vqa import Net_Search
        net = Net_Search(cfg, init)
    net = net.to(dev).train()
    dp.broadcast_parameters(net)
    inputs_cpu, target_cpu = synth_batch(cfg, B, Sx, Sy, V, ANS, 888 + 1000 * rank)
    inputs = tuple(t.to(dev) for t in inputs_cpu)
    target = target_cpu.to(dev)
    loss_fn = torch.nn.BCEWithLogitsLoss(reduction='sum')

    if args.workload == 'train_vqa':
        reducer = dp.GradReducer(list(net.parameters()))
        names_enc = [n[0] for n in cfg.GENOTYPE['enc']]
        names_dec = [n[0] for n in cfg.GENOTYPE['dec']]
    else:
        reducer = dp.SupernetReducer(net)
        MixedOp.MODE = None

    optim = None
    if args.with_optim:   # train_vqa.py:174-183 / search_vqa.py:135-146: Adam(0.9, 0.98, eps 1e-9), clip 1.0
        from mmnas_amd.optim import FlatAdam, WarmupOptimizer
        optim = WarmupOptimizer(1.2e-4 if args.workload == 'train_vqa' else 4e-4,
                                FlatAdam(reducer.fg.params, betas=(0.9, 0.98), eps=1e-9, grads=reducer.fg),
                                epoch_steps=1000, warmup=True, max_norm=1.0)

    flops_acc = [0.0]

    def step():
        if args.workload == 'train_vqa':
            reducer.begin_step()
            loss = loss_fn(net(inputs), target)
            loss.backward()
            reducer.finish()
            if optim is not None:
                optim.step()
            flops_acc[0] += step_flops(cfg, names_enc, names_dec, B, Sx, Sy, ANS)
        else:
            net.reset_binary_gates()
            reducer.begin_weight_step()
            loss = loss_fn(net(inputs), target)
            loss.backward()
            reducer.finish_weight_step()
            if optim is not None:
                optim.step()
            ne = [m.Used_OPS[m.active_index[0]] for m in net.redundant_modules[:12]]
            nd = [m.Used_OPS[m.active_index[0]] for m in net.redundant_modules[12:]]
            flops_acc[0] += step_flops(cfg, ne, nd, B, Sx, Sy, ANS)
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    flops_acc[0] = 0.0
    prof_on = not args.no_prof
    if os.environ.get('MMNAS_BENCH_TORCHPROF'):   # diagnostic: which ATen ops run per step (counts), with python stacks
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU], with_stack=True) as tp:
            for _ in range(3):
                step()
        barrier()
        for ev in sorted(tp.key_averages(group_by_stack_n=12), key=lambda e: -e.count):
            if ev.key.split('::')[-1] in ('fill_', 'add_', 'add', 'copy_', 'cat', 'mul', 'zero_', 'sum', 'div', 'sub', 'where', 'gt'):
                where = [s for s in ev.stack if 'repo' in s or 'mmnas' in s]
                print('%-14s x%-3d %s' % (ev.key, ev.count, ' <- '.join(w.split('repo/')[-1] for w in where[:4])), file=sys.stderr)
    prof_host = None
    if os.environ.get('MMNAS_BENCH_CPROFILE'):   # diagnostic: python-level profile of the issuing thread (forward + loss)
        import cProfile
        prof_host = cProfile.Profile()
        prof_host.enable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_enqueue = time.perf_counter() - t0   # host time to issue the steps (diagnostic: close to `elapsed` = host-bound)
    barrier()
    elapsed = time.perf_counter() - t0
    if prof_host is not None:
        import pstats
        prof_host.disable()
        pstats.Stats(prof_host, stream=sys.stderr).sort_stats('tottime').print_stats(40)
    timed_flops = flops_acc[0]
    # Roofline pass: the same step repeated right after the timed region with every library launch carrying a
    # start/stop HIP event (on the launch stream).  It is kept out of the timed region because the events
    # themselves cost ~9 % of the step (a completion signal per dispatch); launches, shapes and data are identical.
    stats = None
    prof_steps = min(args.steps, 10)
    if prof_on:
        L.check(lib.mmnas_prof_enable(1))
        tp = time.perf_counter()
        for _ in range(prof_steps):
            step()
        barrier()
        prof_elapsed = time.perf_counter() - tp
        arr = (L.ProfStat * len(L.K_NAMES))()
        L.check(lib.mmnas_prof_collect(arr))
        L.check(lib.mmnas_prof_enable(0))
        stats = {n: dict(ms=arr[i].ms, flops=arr[i].flops, bytes=arr[i].bytes, launches=arr[i].launches)
                 for i, n in enumerate(L.K_NAMES)}
    flops_acc[0] = timed_flops
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    final_loss = float(loss.detach())

    if rank == 0:
        out = {
            'metric': 'supernet fwd+bwd steps/sec (VQA arch, bs=64)',
            'value': world * args.steps / elapsed,
            'unit': 'steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1000.0 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if not args.gemm_split else 'f32 results from bf16x%d split-operand MFMA products (experiment)' % args.gemm_split,
            'data': 'synthetic',
            'config': {'workload': {'train_vqa': 'arch/mmnas_vqa.json Net_Full fwd+loss+bwd, HSIZE 512, B=64/GPU, 100x2048 regions + 14 tokens, dropout 0.1 (BASELINE configs[1])',
                                    'search_vqa': 'Net_Search supernet weight step (sample+fwd+loss+bwd), HSIZE 256, B=64/GPU (BASELINE configs[2])'}[args.workload],
                       'global_batch': B * world, 'parallelism': 'dp%d' % world,
                       'grad_allreduce': ('rccl' if backend == 'nccl' else backend) if world > 1 else 'none',
                       'optimizer_in_step': bool(args.with_optim), 'gemm_split': args.gemm_split},
            'samples_per_s': world * args.steps * B / elapsed,
            'algorithmic_tflops_per_gpu': flops_acc[0] / elapsed / 1e12,
            'final_loss': final_loss,
            'host_issue_ms_per_step': 1000.0 * t_enqueue / args.steps,
        }
        if stats:
            gm = stats['gemm']
            ach = gm['flops'] / (gm['ms'] * 1e-3) / 1e12 if gm['ms'] > 0 else 0.0
            traffic, traffic_detail = pmc_traffic(args.workload)
            kname = 'gemm_kernel / gemm_pair_kernel <BM,BN> (fp32 MFMA 32x32x2; NT/NN/TN, grouped; dgrad+wgrad pairs in one launch)'
            if args.gemm_split:   # algorithmic (fp32-equivalent) flops still priced against the fp32-MFMA peak, for comparison only
                kname = 'gemm_kernel<BM,BN,NS=%d> (%d bf16 MFMA 32x32x16 products of split operands per fp32 product; ' \
                        'achieved = algorithmic flops, peak = the fp32 MFMA peak)' % (args.gemm_split // 3 + 1, args.gemm_split)
            out['roofline'] = {'kernel': kname, 'bound': 'mfma',
                               'achieved': ach, 'peak': PEAK_MFMA_F32_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': ach / PEAK_MFMA_F32_TFLOPS, 'traffic': traffic, 'traffic_pmc': traffic_detail,
                               'algorithmic_bytes_per_launch': gm['bytes'] / max(gm['launches'], 1),
                               'avg_launch_us': 1e3 * gm['ms'] / max(gm['launches'], 1),
                               'launches_per_step': gm['launches'] / prof_steps,
                               'share_of_step_time': gm['ms'] * 1e-3 / prof_elapsed}
            tot_ms = sum(s['ms'] for s in stats.values())
            out['kernel_classes'] = {
                n: {'ms_per_step': s['ms'] / prof_steps, 'launches_per_step': s['launches'] / prof_steps,
                    'tflops': (s['flops'] / (s['ms'] * 1e-3) / 1e12) if s['ms'] > 0 else 0.0,
                    'algorithmic_gbs': (s['bytes'] / (s['ms'] * 1e-3) / 1e9) if s['ms'] > 0 else 0.0}
                for n, s in stats.items()}
            ro = stats['rowops']   # the HBM-bound class (LayerNorm forward / backward, column sums)
            if ro['ms'] > 0:
                gbs = ro['bytes'] / (ro['ms'] * 1e-3) / 1e9
                out['hbm_kernels'] = {'kernel': 'ln_fwd/ln_bwd/colsum (rowops.hip)', 'bound': 'hbm', 'achieved': gbs,
                                      'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS}
            out['roofline_pass'] = {'steps': prof_steps, 'ms_per_step': 1000.0 * prof_elapsed / prof_steps,
                                    'kernel_ms_per_step': tot_ms / prof_steps,
                                    'note': 'same step repeated after the timed region with per-launch HIP events'}
        if world == 1 and not args.no_cpu_baseline:
            plan = None
            if args.workload == 'search_vqa':
                plan = [(m.active_index, m.inactive_index) for m in net.redundant_modules]
            out['cpu_baseline'] = cpu_baseline(cfg, net, args.workload, inputs_cpu, target_cpu, plan)
            out['gpu_vs_cpu'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

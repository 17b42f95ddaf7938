"""Benchmark of the candidate-operator hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload all|search_vqa|arch_vqa|bilevel_vqa|train_vqa]

Headline (BASELINE.json `metric`: supernet fwd+bwd steps/sec, VQA, bs=64 per GPU): the supernet WEIGHT step of
search_vqa.py:279-292 -- sample an architecture, forward, BCE(sum), backward, gradient exchange -- on Net_Search with
HSIZE 256, 100x2048 region features + 14 tokens, dropout 0.1, one synthetic batch of 64 per GPU resident in HBM.
With the default `--workload all` the same process then times, each for K steps of its own, and reports under "sub":
    arch_step  the architecture step, MODE 'full' (search_vqa.py:317-331): forward of all 96 candidates, backward
               through the 30 sampled ones, alpha-gate gradient exchange
    bilevel    one round of the bilevel loop (search_vqa.py:149-150): 5 weight steps WITH clip + Adam and 1 arch
               step WITH the alpha update; a "step" of this record is one of its 6 steps
    train_vqa  Net_Full(arch/mmnas_vqa.json) forward + loss + backward, HSIZE 512 (train_vqa.py:295-299;
               BASELINE configs[1])
and, at N = 1 only (short records: throughput, host issue time, library launches per step):
    search_vqa_stream  the headline step fed a FRESH batch every step: 8 distinct pinned CPU batches (52 MB of region
               features + raw boxes) through data.DevicePrefetcher, the [B,100,100,4] relation tensor computed on the
               device from the boxes (search_vqa.py:271,282 ships a CPU batch through DDP's scatter every step)
    search_vqa_dropin  the reference's OWN statement sequence (search_vqa.py:279-301: unused_modules_off, forward,
               BCEWithLogitsLoss, the three `0 * sum(p.sum())` lines, net.zero_grad(), backward, clip_grad_norm_,
               WarmupOptimizer over torch.optim.Adam, unused_modules_back) on the per-operator path -- what an unchanged
               script gets, optimizer included
    search_vqa_dp1 / train_vqa_dp1  the search / training steps with the data-parallel exchange machinery ON in a
               one-rank RCCL group (pack -> all-reduce -> scatter, communication stream, bucket events inside the
               backbone call): the non-network cost of the exchange
    search_vqa_unpad / train_vqa_unpad  the same steps with the RAGGED decoder stream (ops.set_unpad): the decoder operators
               on the valid region rows only; logits and gradients equal the padded computation's
Every timed block is repeated `--repeats` times (default 5, same `--steps` each); `value` / `ms_per_step` are the
MEDIAN block, `value_min` / `value_max` the slowest / fastest block.
Every record carries its own `roofline` (the fp32-MFMA GEMM kernel class, the dominant kernel, measured with HIP
events on the launch stream over a repeat of the timed steps right after the timed region -- the events cost ~9 % of a
step, so they stay out of the throughput measurement) and, at N = 1, its own `cpu_baseline` (the CPU oracle, a port of
the reference step, timed on this box's host cores on the same weights and batch).

N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (one rank per GPU,
RCCL).  Every rank runs its own batch (weak scaling), gradients are averaged inside the step, time is the max over
ranks.  `--gpus N` without a launcher (WORLD_SIZE unset) spawns the N ranks itself before anything touches the GPU;
a WORLD_SIZE that disagrees with --gpus is an error.  Rank 0 prints ONE compact JSON line (< 8000
characters: the contract's fields + roofline + cpu_baseline + one short object per sub record) as the last line of stdout;
the full records (kernel classes, per-block times, PMC detail, notes) go to gpurun_out/bench_full.json (--full-out).
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

PEAK_MFMA_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_MFMA_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense (v_mfma_f32_32x32x16_bf16 at 32 cycles)
PEAK_HBM_GBS = 8000.0
B_DEFAULT, SX, SY, VOCAB, ANS = 64, 14, int(os.environ.get('MMNAS_BENCH_SY', '100')), 20000, 3129   # (MMNAS_BENCH_SY: tuning experiments only)

WORKLOADS = {
    'search_vqa': 'Net_Search supernet WEIGHT step: sample + fwd + BCE(sum) + bwd (+ gradient all-reduce), HSIZE 256, B=64/GPU, '
                  '100x2048 regions + 14 tokens, dropout 0.1 (search_vqa.py:279-292; BASELINE configs[2])',
    'arch_vqa': "Net_Search ARCH step, MODE 'full': fwd of all 96 candidates, bwd through the 30 sampled, alpha-gate exchange "
                '(search_vqa.py:317-331)',
    'bilevel_vqa': 'bilevel NAS round: 5 weight steps with clip_grad_norm + Adam, then 1 arch step with the alpha Adam update '
                   '(search_vqa.py:149-150,279-337); one "step" = one of the 6',
    'train_vqa': 'arch/mmnas_vqa.json Net_Full fwd + BCE(sum) + bwd, HSIZE 512, B=64/GPU, 100x2048 regions + 14 tokens, dropout 0.1 '
                 '(train_vqa.py:295-299; BASELINE configs[1])',
    'search_vqa_stream': 'the search_vqa step with a fresh batch per step: 8 pinned CPU batches through DevicePrefetcher (52 MB features '
                         '+ boxes H2D on a copy stream), relation tensor computed on the device (search_vqa.py:271,282)',
    'search_vqa_dropin': "the reference's own statements (search_vqa.py:279-301) on the per-operator path: unused_modules_off, fwd, "
                         'BCEWithLogitsLoss, 3x `0 * sum(p.sum())`, net.zero_grad(), bwd, clip_grad_norm_, WarmupOptimizer + '
                         'torch.optim.Adam, unused_modules_back; HSIZE 256, B=64, resident batch',
    'search_vqa_dp1': 'the search_vqa step with the gradient exchange running in a one-rank RCCL group (3 buckets: pack -> all-reduce '
                      '-> scatter on the communication stream, bucket events inside the backbone call)',
    'train_vqa_dp1': 'the train_vqa step with the bucketed in-place all-reduce running in a one-rank RCCL group',
    'search_vqa_unpad': 'the search_vqa step with the RAGGED decoder stream (ops.set_unpad): same batch, same logits and gradients; the '
                        'decoder operators run on the valid region rows only (the synthetic batch pads n_b ~ U{10..100} regions to 100, '
                        'SURVEY 8d; the reference computes on the padding rows and masks them)',
    'train_vqa_unpad': 'the train_vqa step with the ragged decoder stream',
    'train_vgd': 'arch/mmnas_vgd.json Net_Full fwd + KLDiv/SmoothL1 loss + bwd, HSIZE 512, B=64/GPU, 100x2048 regions + 15 tokens, '
                 'dropout 0.1 (train_vgd.py:309-334; BASELINE configs[3])',
    'train_itm': 'arch/mmnas_itm.json Net_Full hard-negative triplet step: 3 fwd + BCE_Loss + bwd, HSIZE 512, B=160/GPU, 36x2048 regions '
                 '+ 50 tokens, dropout 0.1, fp32 (train_itm.py:380-391; BASELINE configs[4] at the reference precision)',
}
EXTRA = ('train_vgd', 'train_itm')   # not part of --workload all (the driver line): run them by name
N1_SUBS = ('search_vqa_stream', 'search_vqa_dropin', 'search_vqa_dp1', 'train_vqa_dp1', 'search_vqa_unpad', 'train_vqa_unpad')   # part of `all` at N = 1 only
METRICS = {
    'search_vqa': 'supernet fwd+bwd steps/sec (VQA arch, bs=64)',
    'arch_vqa': 'supernet arch-step (all candidates fwd, sampled bwd) steps/sec (VQA, bs=64)',
    'bilevel_vqa': 'bilevel NAS steps/sec (5 weight + 1 arch per round, optimizers included; VQA, bs=64)',
    'train_vqa': 'fixed-architecture fwd+bwd steps/sec (arch/mmnas_vqa.json, bs=64)',
    'search_vqa_stream': 'supernet fwd+bwd steps/sec, fresh host batch per step (VQA arch, bs=64)',
    'search_vqa_dropin': 'reference-loop weight steps/sec, optimizer included (search_vqa.py:279-301 unchanged; VQA arch, bs=64)',
    'search_vqa_dp1': 'supernet fwd+bwd steps/sec with the exchange machinery in a one-rank RCCL group (VQA arch, bs=64)',
    'train_vqa_dp1': 'fixed-architecture fwd+bwd steps/sec with the exchange machinery in a one-rank RCCL group (bs=64)',
    'search_vqa_unpad': 'supernet fwd+bwd steps/sec, decoder stream on the valid region rows only (VQA arch, bs=64)',
    'train_vqa_unpad': 'fixed-architecture fwd+bwd steps/sec, decoder stream on the valid region rows only (bs=64)',
    'train_vgd': 'fixed-architecture fwd+bwd steps/sec (arch/mmnas_vgd.json, bs=64)',
    'train_itm': 'triplet (3 fwd + 1 bwd) steps/sec (arch/mmnas_itm.json, bs=160)',
}


# ---------------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without torch.distributed.run
# ---------------------------------------------------------------------------------------------------------------------
def _self_spawn(n, argv, script=None):
    """Start one child per GPU (fresh interpreters: nothing here has touched the GPU) and relay rank 0's line."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # a rank that dies leaves the others parked in a collective: poll all of them, and once one has exited non-zero
    # (or the whole run exceeds its limit) end the exact children started here
    import threading
    import time as _time
    got = {}
    reader = threading.Thread(target=lambda: got.setdefault('out', procs[0].stdout.read()), daemon=True)
    reader.start()
    limit = float(os.environ.get('MMNAS_BENCH_SPAWN_TIMEOUT', '3600'))
    t0, rc = _time.time(), 0
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            rc = max(abs(c) for c in codes)
            break
        bad = [c for c in codes if c not in (None, 0)]
        if bad or _time.time() - t0 > limit:
            rc = abs(bad[0]) if bad else 124
            sys.stderr.write('bench.py: %s; stopping the other ranks\n'
                             % ('a rank exited with %d' % rc if bad else 'no result after %.0f s' % limit))
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            break
        _time.sleep(0.2)
    reader.join(timeout=5)
    sys.stdout.write(got.get('out', b'').decode())
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# workload construction
# ---------------------------------------------------------------------------------------------------------------------
def make_cfg(kind):
    from types import SimpleNamespace
    c = dict(DROPOUT_R=0.1, REL_SIZE=64, OPS_NORM=True, OPS_RESIDUAL=True, LAYERS=1,
             NODES={'enc': 12, 'dec': 18}, ATTFLAT_GLIMPSES=1, ATTFLAT_MLP_SIZE=512, FRCNFEAT_SIZE=2048,
             BBOX_FEATURE=False, BBOXFEAT_EMB_SIZE=1024, WORD_EMBED_SIZE=300, ALPHA_INIT_TYPE='normal',
             SCORES_LOSS='kld', GENOTYPE=None)
    if kind.startswith('train'):      # train_vqa.py:136-154 (train_vgd.py / train_itm.py: the same values)
        c.update(HSIZE=512, ATTFLAT_OUT_SIZE=1024)
        arch = {'train': 'mmnas_vqa', 'train_vgd': 'mmnas_vgd', 'train_itm': 'mmnas_itm'}[kind]
        with open(os.path.join(REPO, 'arch', arch + '.json')) as f:
            g = json.load(f)
        c['GENOTYPE'] = g[sorted(g)[-1]]
    else:                    # search_vqa.py:87-114
        c.update(HSIZE=256, ATTFLAT_OUT_SIZE=512)
    return SimpleNamespace(**c)


def synth_batch(cfg, B, Sx, Sy, V, ans, seed):
    """Synthetic batch with the loaders' tensor contract: zero rows = padding."""
    import torch
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


def stem_flops_fwd(cfg, B, Sx, Sy, ans):
    d = cfg.HSIZE
    stem = 2 * B * Sy * cfg.FRCNFEAT_SIZE * d + 2 * B * Sx * 4 * d * (cfg.WORD_EMBED_SIZE + d)
    stem += 2 * B * Sy * Sy * 4 * 64 + 2 * B * (Sx + Sy) * d * cfg.ATTFLAT_MLP_SIZE
    stem += 2 * 2 * B * d * cfg.ATTFLAT_OUT_SIZE + 2 * B * cfg.ATTFLAT_OUT_SIZE * ans
    return stem


def step_flops(cfg, names_enc, names_dec, B, Sx, Sy, ans, all_enc=None, all_dec=None, lens=None):
    """fwd + bwd (= 2 x fwd) of the differentiated operators and the stem/head; `all_*`: operators that are evaluated
    forward only (the detached candidates of the arch step).  lens (the ragged records): the region count of every sample
    -- the work is then counted on the VALID rows, sample by sample (n_b rows, n_b x n_b attention), which is what the
    ragged stream computes; without it a `_unpad` record would be credited with the padding rows' flops it never executes."""
    if lens is not None:       # (every term of the count is per sample)
        return sum(step_flops(cfg, names_enc, names_dec, 1, Sx, int(n), ans, all_enc, all_dec) for n in lens)
    d = cfg.HSIZE
    f = sum(op_flops_fwd(n, B, Sx, Sy, d, 'enc') for n in names_enc)
    f += sum(op_flops_fwd(n, B, Sx, Sy, d, 'dec') for n in names_dec)
    extra = sum(op_flops_fwd(n, B, Sx, Sy, d, 'enc') for n in (all_enc or []))
    extra += sum(op_flops_fwd(n, B, Sx, Sy, d, 'dec') for n in (all_dec or []))
    return 3 * (f + stem_flops_fwd(cfg, B, Sx, Sy, ans)) + extra


_FIXED_COST = {}


def gemm_fixed_cost_us(L, ops, torch, dev):
    """Measured live: what ONE launch of the default matrix-product kernel costs before its first K-tile and after its last
    -- the intercept of launch time over K at one 64 x 64 tile per CU (N = 64, M = 64 * 256; K = 64 / 256 / 512; 200
    back-to-back launches each, HIP events on the launch stream).  `roofline.fixed_cost_share` = launches per step x this
    / the class's time per step: the part of the d = 256 GEMM class no MFMA rate can shrink."""
    if 'us' in _FIXED_COST:
        return _FIXED_COST
    import ctypes as C
    N, M = 64, 64 * 256
    pts = []
    for K in (64, 256, 512):
        a, b, c = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
        d = ops.gemm_desc(L.GEMM_NT, [dict(M=M, A=[a], B=[b], C=c)], N, K, K, K, N)
        for _ in range(20):
            L.check(L.lib().mmnas_gemm(C.byref(d), L.stream()))
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                L.lib().mmnas_gemm(C.byref(d), L.stream())
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / 200
            best = t if best is None else min(best, t)
        pts.append((K // 32, best))
    n = len(pts)
    mx, my = sum(x for x, _ in pts) / n, sum(y for _, y in pts) / n
    slope = sum((x - mx) * (y - my) for x, y in pts) / sum((x - mx) ** 2 for x, _ in pts)
    _FIXED_COST.update({'us': my - slope * mx, 'us_per_k_tile': slope, 'points_us': {str(32 * x): round(y, 3) for x, y in pts},
                        'method': 'intercept of back-to-back launch time over K, one 64x64 tile per CU (N=64, M=16384), NT'})
    return _FIXED_COST


def pmc_traffic(workload):
    """HBM-side bytes per GEMM launch from the committed PMC passes (profiles/rNN_traffic_<workload>.json, made by
    tools/pmc_traffic.py from two `rocprofv3 --pmc` runs of this benchmark: FETCH_SIZE and WRITE_SIZE cannot share a
    pass; traffic = 2*FETCH + WRITE, the gfx950 correction of MI355X_MICROARCH.md).  None when no file exists."""
    path = None
    for r in ('r05', 'r04', 'r03', 'r02', 'r01'):
        p = os.path.join(REPO, 'profiles', '%s_traffic_%s.json' % (r, workload))
        if os.path.exists(p):
            path = p
            break
    if path is None:
        return None, None
    d = json.load(open(path))
    meta = d.pop('_meta', None)   # written by tools/stamp_profiles.py when the file is committed: which tree it measured
    n = tot = fetch = write = 0.0
    for k, v in d.items():
        if 'gemm_kernel' in k or 'gemm_pair_kernel' in k or 'gemm_ln_kernel' in k:
            n += v['launches']
            tot += v['launches'] * v['traffic_bytes_per_launch']
            fetch += v['launches'] * v['fetch_kb_per_launch'] * 1024.0
            write += v['launches'] * v['write_kb_per_launch'] * 1024.0
    if n == 0:
        return None, None
    return tot / n, {'source': os.path.relpath(path, REPO), 'source_commit': (meta or {}).get('commit'),
                     'source_library_md5': (meta or {}).get('library_md5'),
                     'note': 'replayed from the committed PMC passes, not observed in this run',
                     'fetch_size_bytes_per_launch': fetch / n,
                     'write_size_bytes_per_launch': write / n, 'formula': '2*FETCH_SIZE + WRITE_SIZE (gfx950)'}


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle: allowed here and only here)
# ---------------------------------------------------------------------------------------------------------------------
def cpu_step_seconds(cfg, net, inputs, target, genotype, search, budget_s):
    """Seconds per step of the CPU oracle (port of the reference step: same weights, same batch, dropout through
    torch's RNG) on the host cores; bounded to ~budget_s of CPU work.  Returns (seconds_per_full_step, sample text)."""
    import torch
    from oracle import mmnas_oracle as O
    P = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in net.state_dict().items()}
    p = float(cfg.DROPOUT_R)
    B = target.shape[0]

    def one(nb):
        for v in P.values():
            v.grad = None
        pred = O.net_forward('vqa', P, cfg, tuple(t[:nb] for t in inputs), genotype=genotype, search=search,
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
        return t_sum / n, '%d full steps (B=%d, same weights and batch, dropout 0.1 via torch RNG) of the CPU oracle; %.2f s/step' % (n, B, t_sum / n)
    nb2 = max(nb, min(B, int(B * (budget_s / 2) / est_full) // 4 * 4 or nb))
    t0 = time.perf_counter()
    one(nb2)
    t2 = time.perf_counter() - t0
    return t2 * B / nb2, ('one step on the first %d of the %d samples of the batch (same weights, dropout 0.1 via torch RNG), '
                          'scaled by %d/%d; %.2f s for the slice' % (nb2, B, B, nb2, t2))


def cpu_threads():
    import torch
    # torch's intra-op pool degrades badly when it has far more threads than these small GEMMs can use
    # (measured: 256 threads -> 306 s/step vs ~2.6 s/step with 32): cap it and say so in `cores`
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    return threads


LINE_LIMIT = 8000   # characters; the driver parses the last stdout line and keeps ~16 KB of it


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def compact_line(out, full_ref):
    """The one stdout line: the contract's fields, `roofline`, `cpu_baseline` and one short object per sub record."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'repeats', 'value_min', 'value_max', 'final_loss', 'samples_per_s',
            'algorithmic_tflops_per_gpu', 'step_frac_of_mfma_peak', 'host_issue_ms_per_step_empty_queue', 'gpu_vs_cpu',
            'rank_ms_per_step')
    line = {k: _r(out[k]) for k in keep if k in out}

    def roof(r):
        ks = ('kernel', 'bound', 'limited_by', 'achieved', 'peak', 'unit', 'frac', 'frac_bf16_div6', 'peak_bf16_div6', 'fixed_cost_share',
              'fixed_cost_us_per_launch', 'traffic', 'traffic_source', 'algorithmic_bytes_per_launch',
              'avg_launch_us', 'launches_per_step', 'share_of_step_time', 'peak_bf16_div3', 'frac_bf16_div3')
        return {k: _r(r[k]) for k in ks if k in r}
    if 'roofline' in out:
        line['roofline'] = roof(out['roofline'])
    if 'hbm_kernels' in out:
        line['hbm_kernels'] = {k: _r(v) for k, v in out['hbm_kernels'].items()}
    if 'kernel_classes' in out:   # ms per step of each kernel class: where the step goes
        line['kernel_ms_per_step'] = {n: _r(c['ms_per_step'], 3) for n, c in out['kernel_classes'].items() if c['ms_per_step'] > 0}
    if 'cpu_baseline' in out:
        line['cpu_baseline'] = {k: _r(v) for k, v in out['cpu_baseline'].items()}
    subs = {}
    for name, r in out.get('sub', {}).items():
        if r.get('value') is None:
            subs[name] = {'value': None, 'error': str(r.get('error'))[:120]}
            continue
        s = {'value': _r(r['value'], 3), 'ms_per_step': _r(r['ms_per_step'])}
        for k in ('ms_per_step_vs_plain', 'gpu_vs_cpu', 'library_launches_per_step', 'host_issue_ms_per_step_empty_queue'):
            if k in r:
                s[k] = _r(r[k], 3)
        if 'roofline' in r:
            s['roofline_frac'] = _r(r['roofline']['frac'])
            s['gemm_tflops'] = _r(r['roofline']['achieved'], 2)
        if 'cpu_baseline' in r:
            s['cpu_steps_per_s'] = _r(r['cpu_baseline']['value'])
        subs[name] = s
    if subs:
        line['sub'] = subs
    line['full_record'] = full_ref
    return line


def fit_line(obj):
    """The compact line as text, never longer than LINE_LIMIT: optional detail is dropped, in a fixed order, until it fits
    (the full record is in the side file either way) -- a long record must not cost the run its line."""
    dumps = lambda o: json.dumps(o, separators=(',', ':'))
    line = dumps(obj)
    dropped = []
    for victim in ('kernel_ms_per_step', 'hbm_kernels', 'sub_detail', 'sub', 'cpu_baseline_sample'):
        if len(line) < LINE_LIMIT:
            break
        if victim == 'sub_detail':
            for s_ in obj.get('sub', {}).values():
                for k in [k for k in s_ if k not in ('value', 'ms_per_step', 'error')]:
                    del s_[k]
        elif victim == 'cpu_baseline_sample':
            obj.get('cpu_baseline', {}).pop('sample', None)
        else:
            obj.pop(victim, None)
        dropped.append(victim)
        obj['dropped_for_length'] = dropped
        line = dumps(obj)
    return line


# ---------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', default='all', choices=['all'] + list(WORKLOADS),
                    help='all = search_vqa (headline) + arch_vqa + bilevel_vqa + train_vqa; train_vgd / train_itm run by name only')
    ap.add_argument('--batch', type=int, default=B_DEFAULT)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-budget', type=float, default=20.0, help='seconds of CPU work per cpu_baseline record')
    ap.add_argument('--gemm-split', type=int, default=6, choices=[0, 1, 3, 6],
                    help='matrix products (MMNAS_GEMM_SPLIT): 6 = the library default, 6 bf16-MFMA products of exactly split '
                         'fp32 operands (fp32-grade error); 0 = the fp32 MFMA; 3 = EXPERIMENT (2^-16-class error, not a '
                         'headline); 1 = ONE bf16-MFMA product of bf16-rounded operands, fp32 accumulate -- the reduced-precision '
                         'flavour BASELINE configs[4] names ("fp16 MFMA"; --workload train_itm only, never a headline); the JSON '
                         'line says which in dtype and config')
    ap.add_argument('--no-prof', action='store_true', help='skip the roofline pass (per-launch HIP events)')
    ap.add_argument('--full-out', default=None, help='where the full record goes (default gpurun_out/bench_full.json)')
    ap.add_argument('--no-fixed-cost', action='store_true', help='skip the per-launch fixed-cost microbenchmark behind roofline.fixed_cost_share '
                    '(~1900 extra GEMM launches: keep them out of a kernel trace of the steps)')
    ap.add_argument('--repeats', type=int, default=5, help='timed blocks of --steps steps each; value = the median block')
    args = ap.parse_args()

    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and args.gpus > 1:
        sys.exit(_self_spawn(args.gpus, sys.argv[1:]))
    world = int(env_world or 1)
    if args.gemm_split == 1 and args.workload != 'train_itm':
        sys.stderr.write('bench.py: --gemm-split 1 (single-pass bf16 products) is the BASELINE configs[4] flavour: --workload train_itm only\n')
        sys.exit(2)
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a mislabelled number\n' % (args.gpus, world))
        sys.exit(2)
    os.environ['MMNAS_GEMM_SPLIT'] = str(args.gemm_split)       # read when the library first schedules a GEMM
    # Kernel arguments written straight into device memory: 5.1 ms instead of 6.1-7.3 ms per supernet step on this pool
    # (README).  The image exports it; a box that does not must not silently lose 20-40 %: set it before HIP initialises
    # and say in `config` what the run had.
    # Round 6: the LIBRARY does that when it is imported (mmnas_amd/_lib.py::_launch_configuration, before the first HIP call
    # of this process, below); bench.py only records what `ops.runtime_config()` reports.
    # The contract is ONE JSON line on stdout.  Libraries of the process write there too (RCCL prints its version banner to
    # stdout under NCCL_DEBUG=VERSION, which this pool exports -- through C stdio, i.e. behind the JSON line when stdout is a
    # pipe): drop that setting (the version goes into the JSON line instead), and keep file descriptor 1 pointed at stderr
    # while the workloads run; it is restored for the one line.
    if os.environ.get('NCCL_DEBUG', '').upper() == 'VERSION':
        del os.environ['NCCL_DEBUG']
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist
    from mmnas_amd import ops as _ops_early   # BEFORE the first HIP call: the library establishes its launch configuration at import
    runtime_cfg = _ops_early.runtime_config()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
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
    from mmnas_amd.harness import SearchLoop
    from mmnas_amd.model import mixed
    from mmnas.model.mixed import MixedOp
    lib = L.lib()
    B = args.batch

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: what the line reports about the exchange is OBSERVED, not configured -- the rank count from a real all-reduce of
    # rank ids on the backend the steps use (sum of 0..N-1 and of ones), the sampled architectures compared across ranks
    # before the timed blocks, the persistent LSTM's hand-off timeout flag read after every block, every rank's own time
    multi = {'rccl_ranks_observed': 1, 'rank_id_sum_ok': True, 'same_architecture': None, 'lstm_timed_out': 0, 'lstm_fallback': False}
    if world > 1:
        t = torch.tensor([float(rank), 1.0], device=dev, dtype=torch.float64)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        multi['rccl_ranks_observed'] = int(round(float(t[1])))
        multi['rank_id_sum_ok'] = bool(abs(float(t[0]) - world * (world - 1) / 2.0) < 1e-9)
        if multi['rccl_ranks_observed'] != world or not multi['rank_id_sum_ok']:
            sys.stderr.write('bench.py: the all-reduce saw %d ranks (rank-id sum %s), expected %d\n'
                             % (multi['rccl_ranks_observed'], float(t[0]), world))
            sys.exit(3)

    torch.manual_seed(888)
    ops.manual_seed(888 + rank)
    mixed.seed_arch_sampler(888)
    emb = torch.randn(VOCAB, 300, generator=torch.Generator().manual_seed(1)).numpy()
    init = {'token_size': VOCAB, 'ans_size': ANS, 'pretrained_emb': emb}
    from mmnas_amd.harness import fused_loss
    loss_fn = fused_loss(torch.nn.BCEWithLogitsLoss(reduction='sum'))
    wanted = [w for w in WORKLOADS if w not in EXTRA and (w not in N1_SUBS or world == 1)] if args.workload == 'all' else [args.workload]
    if world > 1 and any(w in N1_SUBS for w in wanted):
        sys.stderr.write('bench.py: %s is an N = 1 record\n' % wanted[0])
        sys.exit(2)

    state = {}

    def search_state():
        if 'search' not in state:
            from mmnas.model.hygr_vqa import Net_Search
            cfg = make_cfg('search')
            net = Net_Search(cfg, init).to(dev).train()
            dp.broadcast_parameters(net)
            # search_vqa.py:135-161: Adam(0.9, 0.98, eps 1e-9) lr 4e-4 with warm-up, clip 1.0; alpha Adam lr 0.1 (0, 0.999)
            loop = SearchLoop(net, loss_fn, net_lr=4e-4, clip=1.0, epoch_steps=1000, warmup=True)
            cpu_in, cpu_tg = synth_batch(cfg, B, SX, SY, VOCAB, ANS, 888 + 1000 * rank)
            cpu_in2, cpu_tg2 = synth_batch(cfg, B, SX, SY, VOCAB, ANS, 777 + 1000 * rank)   # the eval_loader batch of the arch step
            state['search'] = dict(cfg=cfg, net=net, loop=loop, cpu=(cpu_in, cpu_tg),
                                   gpu=(tuple(t.to(dev) for t in cpu_in), cpu_tg.to(dev)),
                                   gpu2=(tuple(t.to(dev) for t in cpu_in2), cpu_tg2.to(dev)))
        return state['search']

    def train_state():
        if 'train' not in state:
            from mmnas.model.full_vqa import Net_Full
            cfg = make_cfg('train')
            net = Net_Full(cfg, init).to(dev).train()
            dp.broadcast_parameters(net)
            cpu_in, cpu_tg = synth_batch(cfg, B, SX, SY, VOCAB, ANS, 888 + 1000 * rank)
            state['train'] = dict(cfg=cfg, net=net, reducer=dp.GradReducer(list(net.parameters())), cpu=(cpu_in, cpu_tg),
                                  gpu=(tuple(t.to(dev) for t in cpu_in), cpu_tg.to(dev)))
        return state['train']

    def rccl_one_rank():
        """A one-rank process group on RCCL inside this process (N = 1 only): lets the reducers run their collectives
        (force_collectives) so the non-network cost of the exchange is on the clock.  Made on first use, AFTER the plain
        records have been measured."""
        if world == 1 and not dist.is_initialized():
            import socket
            sk = socket.socket()
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
            sk.close()
            os.environ['MASTER_ADDR'] = '127.0.0.1'
            os.environ['MASTER_PORT'] = str(port)
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
            state['own_group'] = True

    def dp1_search_state():
        if 'search_dp1' not in state:
            from mmnas.model.hygr_vqa import Net_Search
            rccl_one_rank()
            S0 = search_state()
            net = Net_Search(S0['cfg'], init).to(dev).train()
            net.load_state_dict(S0['net'].state_dict())
            loop = SearchLoop(net, loss_fn, net_lr=4e-4, clip=1.0, epoch_steps=1000, warmup=True, force_collectives=True)
            assert loop.reducer.comm and loop.reducer.comm_stream is not None
            state['search_dp1'] = dict(cfg=S0['cfg'], net=net, loop=loop, gpu=S0['gpu'], gpu2=S0['gpu2'])
        return state['search_dp1']

    def dp1_train_state():
        if 'train_dp1' not in state:
            from mmnas.model.full_vqa import Net_Full
            rccl_one_rank()
            S0 = train_state()
            net = Net_Full(S0['cfg'], init).to(dev).train()
            net.load_state_dict(S0['net'].state_dict())
            red = dp.GradReducer(list(net.parameters()), force_collectives=True)
            assert red.comm and red.comm_stream is not None
            state['train_dp1'] = dict(cfg=S0['cfg'], net=net, reducer=red, gpu=S0['gpu'])
        return state['train_dp1']

    def dropin_state():
        """search_vqa.py:174-199 without the DDP wrapper: the net, torch Adam behind the warm-up schedule, the reference's
        own loss module.  No reducer, no flat gradient buffer: every operator is its own autograd node."""
        if 'dropin' not in state:
            from mmnas.model.hygr_vqa import Net_Search
            from mmnas.utils.optimizer import WarmupOptimizer
            S0 = search_state()
            net = Net_Search(S0['cfg'], init).to(dev).train()
            net.load_state_dict(S0['net'].state_dict())
            opt = WarmupOptimizer(4e-4, torch.optim.Adam(net.net_parameters(), lr=0, betas=(0.9, 0.98), eps=1e-9),
                                  epoch_steps=1000, warmup=True)
            state['dropin'] = dict(cfg=S0['cfg'], net=net, optim=opt, gpu=S0['gpu'], loss=torch.nn.BCEWithLogitsLoss(reduction='sum'))
        return state['dropin']

    def stream_state():
        """8 distinct batches in pinned host memory with the loaders' tensor contract, except that the image relation
        tensor is NOT shipped: the raw boxes are (6 KB instead of 10 MB) and data.relations_on_device builds it."""
        if 'stream' not in state:
            S0 = search_state()
            cfg = S0['cfg']
            batches = []
            for i in range(8):
                (frcn, bbox, _y_rel, ques, x_rel), tg = synth_batch(cfg, B, SX, SY, VOCAB, ANS, 4000 + 17 * i + 1000 * rank)
                g = torch.Generator().manual_seed(50 + i)
                xy = torch.rand(B, SY, 2, generator=g) * 400
                wh = torch.rand(B, SY, 2, generator=g) * 200 + 8
                boxes = torch.cat((xy, xy + wh), dim=-1)
                nobj = (frcn.abs().sum(-1) != 0).sum(-1).to(torch.int32)
                batches.append(tuple(t.pin_memory() for t in (frcn, bbox, boxes, nobj, ques, x_rel, tg)))
            state['stream'] = dict(batches=batches)
        return state['stream']

    def other_state(wl):
        """train_vgd / train_itm: net, reducer and synthetic batches of that task's shapes."""
        if wl not in state:
            import importlib
            from mmnas_amd.harness import BCE_Loss
            task = wl.split('_')[1]
            cfg = make_cfg(wl)
            Net = importlib.import_module('mmnas.model.full_%s' % task).Net_Full
            net = Net(cfg, init).to(dev).train()
            dp.broadcast_parameters(net)
            Bt, Sx, Sy = (160, 50, 36) if task == 'itm' else (B, 15, SY)
            mk = lambda seed: tuple(t.to(dev) for t in synth_batch(cfg, Bt, Sx, Sy, VOCAB, 4, seed)[0])
            g = torch.Generator().manual_seed(5 + rank)
            st = dict(cfg=cfg, net=net, reducer=dp.GradReducer(list(net.parameters())), B=Bt, Sx=Sx, Sy=Sy,
                      pos=mk(888 + 1000 * rank), neg=mk(999 + 1000 * rank), loss=BCE_Loss())
            if task == 'vgd':
                sc = torch.rand(Bt, Sy, generator=g)
                st['tg'] = dict(scores=(sc / sc.sum(-1, keepdim=True)).to(dev), scores_mask=(torch.rand(Bt, Sy, generator=g) < 0.7).float().to(dev),
                                bbox=torch.randn(Bt, Sy, 4, generator=g).to(dev),
                                bbox_mask=((torch.rand(Bt, Sy, 1, generator=g) < 0.4).float() * torch.ones(1, 1, 4)).to(dev))
                st['tg']['scores_mask'][:, 0] = 1
                st['tg']['bbox_mask'][:, 0] = 1
            state[wl] = st
        return state[wl]

    def used_names(net):
        ms = net.redundant_modules
        return ([m.Used_OPS[m.active_index[0]] for m in ms[:12]], [m.Used_OPS[m.active_index[0]] for m in ms[12:]],
                [m.Used_OPS[i] for m in ms[:12] for i in m.inactive_index], [m.Used_OPS[i] for m in ms[12:] for i in m.inactive_index])

    valid_lens = [None]     # set while a `_unpad` record's step runs: step_flops then counts the valid region rows only

    def make_step(wl):
        """-> (step() -> loss, flops accumulator [1], steps-per-call)"""
        if wl.endswith('_unpad'):      # the plain step under ops.set_unpad(True): same net, same batch, same loop object
            base_wl = wl[:-len('_unpad')]
            S0 = train_state() if base_wl.startswith('train') else search_state()
            frcn = S0['cpu'][0][0]
            lens_of_record = [int(v) for v in (frcn.abs().sum(-1) != 0).sum(-1)]     # flops of these records: valid rows only
            inner, fl, per = make_step(base_wl)

            def step():
                prev = ops.set_unpad(True)
                valid_lens[0] = lens_of_record
                try:
                    return inner()
                finally:
                    ops.set_unpad(prev)
                    valid_lens[0] = None
            return step, fl, per
        fl = [0.0]
        if wl in EXTRA:
            from mmnas_amd.harness import itm_triplet_step, vgd_loss
            S = other_state(wl)
            cfg, net, red = S['cfg'], S['net'], S['reducer']
            ne = [n[0] for n in cfg.GENOTYPE['enc']]
            nd = [n[0] for n in cfg.GENOTYPE['dec']]
            passes = 3 if wl == 'train_itm' else 1
            per = passes * step_flops(cfg, ne, nd, S['B'], S['Sx'], S['Sy'], 4)

            def step():
                if wl == 'train_itm':
                    loss = itm_triplet_step(net, S['loss'], S['pos'], S['neg'], reducer=red)
                else:
                    red.begin_step()
                    ps, pr = net(S['pos'])
                    t = S['tg']
                    loss = vgd_loss(ps, pr, t['scores'], t['scores_mask'], t['bbox'], t['bbox_mask'])
                    loss.backward()
                    red.finish()
                fl[0] += per
                return loss
            return step, fl, 1
        if wl == 'train_vqa':
            S = train_state()
            cfg, net, red = S['cfg'], S['net'], S['reducer']
            ne = [n[0] for n in cfg.GENOTYPE['enc']]
            nd = [n[0] for n in cfg.GENOTYPE['dec']]
            per = step_flops(cfg, ne, nd, B, SX, SY, ANS)
            per_valid = {}

            def step():
                red.begin_step()
                loss = loss_fn(net(S['gpu'][0]), S['gpu'][1])
                loss.backward()
                red.finish()
                if valid_lens[0] is None:
                    fl[0] += per
                else:
                    if 'v' not in per_valid:
                        per_valid['v'] = step_flops(cfg, ne, nd, B, SX, SY, ANS, lens=valid_lens[0])
                    fl[0] += per_valid['v']
                return loss
            return step, fl, 1
        if wl == 'train_vqa_dp1':
            S = dp1_train_state()
            cfg, net, red = S['cfg'], S['net'], S['reducer']
            per = step_flops(cfg, [n[0] for n in cfg.GENOTYPE['enc']], [n[0] for n in cfg.GENOTYPE['dec']], B, SX, SY, ANS)

            def step():
                red.begin_step()
                loss = loss_fn(net(S['gpu'][0]), S['gpu'][1])
                loss.backward()
                red.finish()
                fl[0] += per
                return loss
            return step, fl, 1
        if wl == 'search_vqa_dropin':
            S = dropin_state()
            cfg, net, opt, ref_loss = S['cfg'], S['net'], S['optim'], S['loss']
            inp, tgt = S['gpu']

            def step():           # search_vqa.py:279-301, statement by statement (sampling: the net's own reset_binary_gates)
                MixedOp.MODE = None
                net.reset_binary_gates()
                net.unused_modules_off()
                pred = net(inp)
                loss = ref_loss(pred, tgt)
                loss += 0 * sum(p.sum() for p in net.alpha_prob_parameters())
                loss += 0 * sum(p.sum() for p in net.alpha_gate_parameters())
                loss += 0 * sum(p.sum() for p in net.net_parameters())
                net.zero_grad()
                loss.backward()
                torch.nn.utils.clip_grad_norm_(net.net_parameters(), 1.0)
                opt.step()
                ne, nd, _, _ = used_names(net)
                net.unused_modules_back()
                fl[0] += step_flops(cfg, ne, nd, B, SX, SY, ANS)
                return loss
            return step, fl, 1
        if wl == 'search_vqa_stream':
            from mmnas_amd.data import DevicePrefetcher, relations_on_device
            S = search_state()
            cfg, net, loop = S['cfg'], S['net'], S['loop']
            batches = stream_state()['batches']

            def endless():
                i = 0
                while True:
                    yield batches[i % len(batches)]
                    i += 1
            # batch i + 1 uploads on the copy stream while step i computes; the region counts ride on the features (checked
            # against the zero-row padding on the host copy): the ragged stream then needs no device-to-host copy per batch
            it = iter(DevicePrefetcher(endless(), dev, lengths=(0, 3)))

            def step():
                frcn, bbox, boxes, nobj, ques, x_rel, tgt = next(it)
                y_rel = relations_on_device(boxes, nobj)
                loss = loop.weight_step((frcn, bbox, y_rel, ques, x_rel), tgt, optimize=False)
                ne, nd, _, _ = used_names(net)
                fl[0] += step_flops(cfg, ne, nd, B, SX, SY, ANS)
                return loss
            return step, fl, 1
        S = dp1_search_state() if wl == 'search_vqa_dp1' else search_state()
        cfg, net, loop = S['cfg'], S['net'], S['loop']

        def weight(optimize):
            loss = loop.weight_step(S['gpu'][0], S['gpu'][1], optimize=optimize)
            ne, nd, _, _ = used_names(net)
            fl[0] += step_flops(cfg, ne, nd, B, SX, SY, ANS, lens=valid_lens[0])
            return loss

        def arch(optimize):
            loss = loop.arch_step(S['gpu2'][0], S['gpu2'][1], optimize=optimize)
            ne, nd, ie, idec = used_names(net)
            fl[0] += step_flops(cfg, ne, nd, B, SX, SY, ANS, ie, idec)
            return loss

        if wl in ('search_vqa', 'search_vqa_dp1'):
            return (lambda: weight(False)), fl, 1
        if wl == 'arch_vqa':
            return (lambda: arch(False)), fl, 1

        def round_():
            for _ in range(loop.alpha_every):
                weight(True)
            return arch(True)
        return round_, fl, loop.alpha_every + 1

    def lstm_timed_out_any():
        """1 when the persistent LSTM's hand-off timed out on ANY rank since the last call (MAX over the ranks: all of them
        take the same branch afterwards); a negative return of the query is an error of the library, not a time-out."""
        t = int(lib.mmnas_lstm_seq_timed_out(L.stream()))
        if t < 0:
            sys.stderr.write('bench.py: mmnas_lstm_seq_timed_out failed: %s\n' % lib.mmnas_last_error().decode())
            sys.exit(4)
        t = 1 if t else 0
        if world > 1:
            f = torch.tensor([float(t)], device=dev)
            dist.all_reduce(f, op=dist.ReduceOp.MAX)
            t = int(f.item() > 0)
        return t

    def measure(wl, steps, warmup):
        step, fl, per_call = make_step(wl)
        calls = max(1, steps // per_call) if per_call > 1 else steps
        wcalls = max(1, warmup // per_call) if per_call > 1 else warmup
        for _ in range(wcalls):
            step()
        barrier()
        if lstm_timed_out_any() and os.environ.get('MMNAS_LSTM', '1') != '0':
            # the persistent LSTM's hand-off gave up during warm-up ON SOME RANK (its grid was not co-resident -- e.g. beside a
            # collective's kernels on a box this was never run on): EVERY rank falls back to nn.LSTM (MIOpen) and re-warms
            # together -- the re-warm-up steps contain collectives, so the decision must not be one rank's own -- instead of
            # timing NaNs, and the line says so
            os.environ['MMNAS_LSTM'] = '0'
            multi['lstm_fallback'] = True
            for _ in range(wcalls):
                step()
            barrier()
        if world > 1 and wl in ('search_vqa', 'arch_vqa', 'bilevel_vqa'):
            # every rank must have sampled the same operators (seeded CPU sampler): one check before the timed blocks
            same = dp.check_same_architecture(search_state()['net'])
            multi['same_architecture'] = same if multi['same_architecture'] is None else (multi['same_architecture'] and same)
            if not same:
                sys.stderr.write('bench.py: the ranks sampled DIFFERENT architectures: refusing to time a mislabelled exchange\n')
                sys.exit(3)
        if os.environ.get('MMNAS_BENCH_HOST_PROFILE'):   # tuning aid: where the HOST spends a step (cProfile, not timed)
            import cProfile, pstats
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(calls):
                step()
            pr.disable()
            barrier()
            with open(os.environ['MMNAS_BENCH_HOST_PROFILE'] + '.' + wl, 'w') as f:
                pstats.Stats(pr, stream=f).sort_stats('tottime').print_stats(45)
                pstats.Stats(pr, stream=f).sort_stats('cumulative').print_stats(60)
        # `repeats` timed blocks of exactly `calls` calls each, every block bracketed by barrier + synchronize on both
        # sides; the reported block is the MEDIAN (a 0.1 s block on a fresh box is at the mercy of clock ramps)
        nsteps = calls * per_call
        blocks = []
        rank_el = []
        rep = 0
        while rep < max(1, args.repeats):
            fl[0] = 0.0
            barrier()
            t0 = time.perf_counter()
            for _ in range(calls):
                loss = step()
            t_enq = time.perf_counter() - t0   # host time to issue the steps (close to `elapsed` = host-bound)
            barrier()
            el = time.perf_counter() - t0
            # the persistent LSTM's hand-off gives up (and poisons the pass with NaN) when a workgroup of its grid was not
            # resident -- e.g. a collective's kernel held the slot: read the flag after EVERY block.  A block that saw one (on
            # any rank) is not a measurement: all ranks fall back to nn.LSTM together, re-warm, and the blocks start over
            if lstm_timed_out_any():
                if os.environ.get('MMNAS_LSTM', '1') != '0':
                    os.environ['MMNAS_LSTM'] = '0'
                    multi['lstm_fallback'] = True
                    for _ in range(wcalls):
                        step()
                    barrier()
                    lstm_timed_out_any()
                    blocks, rank_el, rep = [], [], 0
                    continue
                multi['lstm_timed_out'] += 1          # (cannot happen: the fallback has no hand-off; reported if it does)
            if world > 1:
                mine = torch.zeros(world, device=dev, dtype=torch.float64)
                mine[rank] = el
                dist.all_reduce(mine)                      # every rank's own time (the line's time is their maximum)
                rank_el.append([float(v) for v in mine])
                el = max(rank_el[-1])
            blocks.append((el, t_enq, fl[0]))
            rep += 1
        order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
        elapsed, t_enqueue, timed_flops = blocks[order[len(order) // 2]]
        el_min, el_max = blocks[order[0]][0], blocks[order[-1]][0]
        # host time of ONE call issued into an EMPTY queue (synchronise first): in the timed blocks the host runs ahead
        # until the launch queue is full and then waits on it, so there `host_issue` approaches the GPU time of a
        # GPU-bound step; this is what the host itself needs
        host_alone = []
        for _ in range(3):
            barrier()
            th = time.perf_counter()
            step()
            host_alone.append(time.perf_counter() - th)
        barrier()
        host_alone_ms = 1000.0 * sorted(host_alone)[1] / per_call
        # Roofline pass: the same steps repeated right after the timed region with every library launch carrying a
        # start/stop HIP event (on the launch stream).  Kept out of the timed region because the events themselves
        # cost ~9 % of the step (a completion signal per dispatch); launches, shapes and data are identical.
        stats = None
        prof_calls = min(calls, max(1, (4 if (wl in N1_SUBS and not wl.endswith('_unpad')) else 10) // per_call))
        prof_elapsed = 0.0
        if not args.no_prof:
            L.check(lib.mmnas_prof_enable(1))
            tp = time.perf_counter()
            for _ in range(prof_calls):
                step()
            barrier()
            prof_elapsed = time.perf_counter() - tp
            arr = (L.ProfStat * len(L.K_NAMES))()
            L.check(lib.mmnas_prof_collect(arr))
            L.check(lib.mmnas_prof_enable(0))
            stats = {n: dict(ms=arr[i].ms, flops=arr[i].flops, bytes=arr[i].bytes, launches=arr[i].launches)
                     for i, n in enumerate(L.K_NAMES)}
        rec = {
            'metric': METRICS[wl], 'value': world * nsteps / elapsed, 'unit': 'steps/s',
            'steps': nsteps, 'warmup': wcalls * per_call, 'ms_per_step': 1000.0 * elapsed / nsteps,
            'repeats': len(blocks), 'value_min': world * nsteps / el_max, 'value_max': world * nsteps / el_min,
            'value_note': 'median of `repeats` timed blocks of `steps` steps each; value_min / value_max = slowest / fastest block',
            'blocks_ms_per_step': [round(1000.0 * b[0] / nsteps, 4) for b in blocks],   # in the order they ran
            'workload': WORKLOADS[wl],
            'samples_per_s': world * nsteps * (state[wl]['B'] if wl in EXTRA else B) / elapsed,
            'algorithmic_tflops_per_gpu': timed_flops / elapsed / 1e12,
            'step_frac_of_mfma_peak': timed_flops / elapsed / 1e12 / PEAK_MFMA_F32_TFLOPS,
            'final_loss': float(loss.detach()),
            'host_issue_ms_per_step': 1000.0 * t_enqueue / nsteps,
            'host_issue_ms_per_step_empty_queue': host_alone_ms,
        }
        if world > 1:
            mid = order[len(order) // 2]
            rec['rank_ms_per_step'] = [round(1000.0 * v / nsteps, 4) for v in rank_el[mid]]   # of the median block, by rank
            rec['rank_ms_per_step_blocks'] = [[round(1000.0 * v / nsteps, 4) for v in row] for row in rank_el]
        if stats and wl in N1_SUBS and not wl.endswith('_unpad'):
            psteps = prof_calls * per_call
            rec['library_launches_per_step'] = sum(s_['launches'] for s_ in stats.values()) / psteps
            rec['library_kernel_ms_per_step'] = sum(s_['ms'] for s_ in stats.values()) / psteps
            rec['launch_note'] = 'launches of libmmnas_hip.so kernels per step (per-launch HIP events over %d extra steps); torch kernels ' \
                                 '(fills, gathers, the reference loop\'s p.sum() / clip / Adam launches) are not counted' % psteps
        elif stats:
            psteps = prof_calls * per_call
            gm = stats['gemm']
            ach = gm['flops'] / (gm['ms'] * 1e-3) / 1e12 if gm['ms'] > 0 else 0.0
            traffic, traffic_detail = pmc_traffic(wl)
            kname = 'gemm_kernel / gemm_pair_kernel <BM,BN> (fp32 MFMA 32x32x2; NT/NN/TN, grouped; dgrad+wgrad pairs in one launch)'
            if args.gemm_split:
                # achieved = ALGORITHMIC (fp32) flops of the products; peak = the fp32 MFMA peak, i.e. the rate the
                # reference's arithmetic type has on the matrix pipe.  The kernel executes gemm_split bf16 MFMA
                # products per fp32 product: the `executed` object prices that work against the bf16 dense peak.
                kname = 'gemm_kernel / gemm_pair_kernel <BM,BN,NS=%d> (each fp32 product as %d v_mfma_f32_32x32x16_bf16 ' \
                        'products of exactly split operands, fp32 accumulate; NT/NN/TN, grouped; dgrad+wgrad pairs in one ' \
                        'launch)' % (args.gemm_split // 3 + 1, args.gemm_split)
                if args.gemm_split == 1:
                    kname = 'gemm_kernel / gemm_pair_kernel <BM,BN,NS=1> (one v_mfma_f32_32x32x16_bf16 product of bf16-rounded operands, fp32 accumulate)'
            # (single-pass bf16 products are bf16 arithmetic: priced against the bf16 dense peak, nothing else)
            peak = PEAK_MFMA_BF16_TFLOPS if args.gemm_split == 1 else PEAK_MFMA_F32_TFLOPS
            rec['roofline'] = {'kernel': kname, 'bound': 'mfma', 'achieved': ach, 'peak': peak,
                               'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': traffic,
                               'traffic_pmc': traffic_detail,
                               'traffic_source': (traffic_detail or {}).get('source') and
                               'PMC 2*FETCH+WRITE per launch, replayed from %s' % traffic_detail['source'],
                               'algorithmic_bytes_per_launch': gm['bytes'] / max(gm['launches'], 1),
                               'avg_launch_us': 1e3 * gm['ms'] / max(gm['launches'], 1),
                               'launches_per_step': gm['launches'] / psteps,
                               # share of the TIMED step (the un-instrumented blocks `value` comes from); the roofline pass
                               # itself runs 15-25 % longer per step (a completion signal per dispatch), so the share of
                               # THAT pass is kept under its own name
                               'share_of_step_time': (gm['ms'] / psteps) / (1000.0 * elapsed / nsteps),
                               'share_of_instrumented_pass': gm['ms'] * 1e-3 / prof_elapsed}
            if wl in ('search_vqa', 'arch_vqa', 'bilevel_vqa', 'train_vqa') and world == 1 and not args.no_fixed_cost:
                # What limits the class: at d = 256 a launch is ~4.5 us of fixed cost around 8 K-tiles, and halving the MFMA
                # work moves the step by 2 % (profiles/r05_bench_search_vqa_gemm_split3.json) -- the class is priced against
                # the MFMA peak (`bound`, the contract's word for which roofline `peak` is) but LIMITED by launch latency /
                # fixed cost; `fixed_cost_share` says how much of the class's time that is, measured in this run.
                fc = gemm_fixed_cost_us(L, ops, torch, dev)
                share = (gm['launches'] / psteps) * fc['us'] * 1e-3 / max(gm['ms'] / psteps, 1e-9)
                rec['roofline']['fixed_cost_us_per_launch'] = fc['us']
                rec['roofline']['fixed_cost_share'] = share
                rec['roofline']['fixed_cost_detail'] = fc
                rec['roofline']['limited_by'] = ('latency / fixed cost' if share >= 0.15 else 'mfma issue (K loop)')
            if args.gemm_split > 1:
                rec['roofline']['peak_note'] = ('`peak` / `frac`: fp32 MFMA dense peak (the reference arithmetic type; a frac > 1 would be '
                                                'possible here because the products run on the bf16 pipe); `peak_bf16_div%d` / `frac_bf16_div%d`: '
                                                'the bf16 dense peak divided by the %d bf16 products per fp32 product') % ((args.gemm_split,) * 3)
                rec['roofline']['peak_bf16_div%d' % args.gemm_split] = PEAK_MFMA_BF16_TFLOPS / args.gemm_split
                rec['roofline']['frac_bf16_div%d' % args.gemm_split] = ach * args.gemm_split / PEAK_MFMA_BF16_TFLOPS
                rec['roofline']['executed'] = {'mfma_tflops': ach * args.gemm_split, 'peak': PEAK_MFMA_BF16_TFLOPS,
                                               'frac': ach * args.gemm_split / PEAK_MFMA_BF16_TFLOPS,
                                               'note': '%d bf16 MFMA products per algorithmic fp32 product' % args.gemm_split}
            tot_ms = sum(s['ms'] for s in stats.values())
            rec['kernel_classes'] = {
                n: {'ms_per_step': s['ms'] / psteps, 'launches_per_step': s['launches'] / psteps,
                    'tflops': (s['flops'] / (s['ms'] * 1e-3) / 1e12) if s['ms'] > 0 else 0.0,
                    'algorithmic_gbs': (s['bytes'] / (s['ms'] * 1e-3) / 1e9) if s['ms'] > 0 else 0.0}
                for n, s in stats.items()}
            ro = stats['rowops']   # the HBM-bound class (LayerNorm forward / backward, column sums, gated sums)
            if ro['ms'] > 0:
                gbs = ro['bytes'] / (ro['ms'] * 1e-3) / 1e9
                rec['hbm_kernels'] = {'kernel': 'ln_fwd/ln_bwd/colsum/mixed_sum (rowops.hip, mixed.hip)', 'bound': 'hbm',
                                      'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS}
            rec['roofline_pass'] = {'steps': psteps, 'ms_per_step': 1000.0 * prof_elapsed / psteps,
                                    'kernel_ms_per_step': tot_ms / psteps,
                                    'note': 'same steps repeated after the timed region with per-launch HIP events'}
        return rec

    recs = {}
    for wl in wanted:
        steps, warm = args.steps, args.warmup
        if wl == 'bilevel_vqa':
            steps, warm = max(6, args.steps // 6 * 6), 6
        if wl in N1_SUBS and args.workload == 'all':
            # secondary records must never cost the driver its headline line: a failure (e.g. RCCL refusing a one-rank
            # group on some box) is reported in place of the record
            try:
                recs[wl] = measure(wl, steps, warm)
            except Exception as e:   # noqa: BLE001
                recs[wl] = {'metric': METRICS[wl], 'value': None, 'error': '%s: %s' % (type(e).__name__, str(e)[:300])}
        else:
            recs[wl] = measure(wl, steps, warm)
    for wl, ref in (('search_vqa_stream', 'search_vqa'), ('search_vqa_dp1', 'search_vqa'), ('train_vqa_dp1', 'train_vqa'),
                    ('search_vqa_unpad', 'search_vqa'), ('train_vqa_unpad', 'train_vqa')):
        if recs.get(wl, {}).get('value') and recs.get(ref, {}).get('value'):
            recs[wl]['ms_per_step_vs_plain'] = recs[wl]['ms_per_step'] / recs[ref]['ms_per_step']
            recs[wl]['plain_record'] = ref
    for wl in ('search_vqa_dp1', 'train_vqa_dp1'):
        if recs.get(wl, {}).get('value'):
            recs[wl]['config'] = {'grad_allreduce': 'rccl', 'rccl_ranks': 1, 'force_collectives': True}
    if recs.get('search_vqa_dropin', {}).get('value') and recs.get('bilevel_vqa', {}).get('value'):
        recs['search_vqa_dropin']['harness_bilevel_ms_per_step'] = recs['bilevel_vqa']['ms_per_step']

    # ---- CPU baselines (rank 0, N = 1 only) ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = cpu_threads()
        t_w = t_a = None
        if any(w in recs for w in ('search_vqa', 'arch_vqa', 'bilevel_vqa')):
            S = search_state()
            net = S['net']
            mixed.seed_arch_sampler(888)
            MixedOp.MODE = None
            net.reset_binary_gates()
            plan = [(m.active_index, m.inactive_index) for m in net.redundant_modules]
            if 'search_vqa' in recs or 'bilevel_vqa' in recs:
                t_w, txt_w = cpu_step_seconds(S['cfg'], net, S['cpu'][0], S['cpu'][1], None,
                                              {'mode': None, 'enc': plan[:12], 'dec': plan[12:]}, args.cpu_budget)
            if 'arch_vqa' in recs or 'bilevel_vqa' in recs:
                t_a, txt_a = cpu_step_seconds(S['cfg'], net, S['cpu'][0], S['cpu'][1], None,
                                              {'mode': 'full', 'enc': plan[:12], 'dec': plan[12:]}, args.cpu_budget)
            if 'search_vqa' in recs:
                recs['search_vqa']['cpu_baseline'] = {'value': 1.0 / t_w, 'unit': 'steps/s', 'cores': threads, 'kind': 'port', 'sample': txt_w,
                                                      'host_cpus': os.cpu_count()}   # (`cores` = threads used: capped at 32, see cpu_threads)
            if 'arch_vqa' in recs:
                recs['arch_vqa']['cpu_baseline'] = {'value': 1.0 / t_a, 'unit': 'steps/s', 'cores': threads, 'kind': 'port', 'sample': txt_a}
            if 'bilevel_vqa' in recs:
                recs['bilevel_vqa']['cpu_baseline'] = {
                    'value': 6.0 / (5 * t_w + t_a), 'unit': 'steps/s', 'cores': threads, 'kind': 'port',
                    'sample': 'composed from the two samples above: 6 / (5 x weight-step + 1 x arch-step seconds); the optimizer '
                              'updates (Adam over 37 M parameters, < 0.1 s on these cores) are not included'}
        if 'train_vqa' in recs:
            S = train_state()
            t_t, txt_t = cpu_step_seconds(S['cfg'], S['net'], S['cpu'][0], S['cpu'][1], S['cfg'].GENOTYPE, None, args.cpu_budget)
            recs['train_vqa']['cpu_baseline'] = {'value': 1.0 / t_t, 'unit': 'steps/s', 'cores': threads, 'kind': 'port', 'sample': txt_t}
        for r in recs.values():
            if 'cpu_baseline' in r:
                r['gpu_vs_cpu'] = r['value'] / r['cpu_baseline']['value']

    if rank == 0:
        head_wl = wanted[0]
        head = recs[head_wl]
        out = {
            'metric': head['metric'], 'value': head['value'], 'unit': 'steps/s',
            'n_gpus': world, 'steps': head['steps'], 'warmup': head['warmup'], 'ms_per_step': head['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {0: 'f32', 6: 'f32 (bf16x6 split-operand MFMA, fp32 accumulate)',
                      3: 'f32 results from bf16x3 split-operand MFMA products (EXPERIMENT: 2^-16-class products)',
                      1: 'bf16 matrix products (one v_mfma_f32_32x32x16_bf16 pass on bf16-rounded operands, fp32 accumulate; storage, '
                         'attention cores and everything else f32): REDUCED PRECISION, BASELINE configs[4] flavour'}[args.gemm_split],
            'data': 'synthetic',
            'config': {'workload': head['workload'], 'global_batch': (state[head_wl]['B'] if head_wl in EXTRA else B) * world, 'parallelism': 'dp%d' % world,
                       'grad_allreduce': ('rccl' if backend == 'nccl' else backend) if world > 1 else 'none',
                       'rccl_ranks': multi['rccl_ranks_observed'],     # counted by an all-reduce of ones, not read from the env
                       'rank_id_sum_ok': multi['rank_id_sum_ok'], 'same_architecture': multi['same_architecture'],
                       'lstm_timed_out': multi['lstm_timed_out'], 'lstm_fallback': multi['lstm_fallback'],
                       'dp_rows': os.environ.get('MMNAS_DP_ROWS', '1'),      # embedding gradient exchanged as rows (dp.RowExchange)
                       'dp_buckets': getattr(getattr(state.get('search', {}).get('loop'), 'reducer', None), 'n_buckets', None),
                       'rccl_env': {k: v for k, v in os.environ.items() if k.startswith(('NCCL_', 'RCCL_'))},
                       'rccl_version': '.'.join(str(x) for x in torch.cuda.nccl.version()) if hasattr(torch.cuda, 'nccl') else None,
                       'optimizer_in_step': False, 'gemm_split': args.gemm_split,
                       'hip_force_dev_kernarg': runtime_cfg['hip_force_dev_kernarg'],
                       # 'set_by_library' takes effect only if nothing initialised HIP before the library was imported (under
                       # rocprofv3 the tool's preloaded library does: the variable must then come from the environment)
                       'hip_force_dev_kernarg_source': runtime_cfg['hip_force_dev_kernarg_source']},
        }
        for k, v in head.items():
            if k not in out and k not in ('workload',):
                out[k] = v
        sub = {{'arch_vqa': 'arch_step', 'bilevel_vqa': 'bilevel'}.get(w, w): r for w, r in recs.items() if w != head_wl}
        if sub:
            out['sub'] = sub
        # The driver keeps ~16 KB of stdout: the LINE is the compact record (headline + config + roofline + cpu_baseline +
        # one short object per sub record); everything measured goes to the side file the line names.
        full_path = args.full_out or os.path.join(REPO, 'gpurun_out', 'bench_full.json')
        try:
            os.makedirs(os.path.dirname(full_path), exist_ok=True)
            with open(full_path, 'w') as f:
                json.dump(out, f, indent=1)
            full_ref = os.path.relpath(full_path, REPO)
        except OSError as e:
            full_ref = 'not written: %s' % e
        line = fit_line(compact_line(out, full_ref))
    if world > 1 or state.get('own_group'):
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)      # the untouched stdout back in place: the line is its last (and only) line
    os.close(real_stdout)
    if rank == 0:
        print(line, flush=True)


if __name__ == '__main__':
    main()

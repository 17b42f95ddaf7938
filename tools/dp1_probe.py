import os, sys, time, statistics
sys.path.insert(0, '/root/repo')
import torch, torch.distributed as dist
import bench
from mmnas.model.hygr_vqa import Net_Search
from mmnas_amd import ops, dp
from mmnas_amd.harness import SearchLoop
from mmnas_amd.model import mixed
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = '29533'
os.environ.pop('NCCL_DEBUG', None)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
torch.manual_seed(888); ops.manual_seed(888); mixed.seed_arch_sampler(888)
cfg = bench.make_cfg('search')
emb = torch.randn(bench.VOCAB, 300, generator=torch.Generator().manual_seed(1)).numpy()
init = {'token_size': bench.VOCAB, 'ans_size': bench.ANS, 'pretrained_emb': emb}
cpu_in, cpu_tg = bench.synth_batch(cfg, 64, bench.SX, bench.SY, bench.VOCAB, bench.ANS, 888)
inp, tgt = tuple(t.to(dev) for t in cpu_in), cpu_tg.to(dev)
res = {}
for force, nb, rows, inl in ((False, 3, '1', '1'), (True, 3, '1', '0'), (True, 3, '1', '1'), (True, 3, '0', '1'), (True, 1, '1', '1'), (False, 3, '1', '1'), (True, 3, '1', '0'), (True, 3, '1', '1')):
    os.environ['MMNAS_DP_ROWS'] = rows
    os.environ['MMNAS_DP_INLINE'] = inl
    net = Net_Search(cfg, init).to(dev).train()
    loop = SearchLoop(net, None, net_lr=4e-4, clip=1.0, epoch_steps=1000, warmup=True, force_collectives=force, n_buckets=nb)
    red = loop.reducer
    T = {'finish': [], 'begin': [], 'launch': []}
    of, ob, ol = red.finish_weight_step, red.begin_weight_step, red._launch
    def wrap(name, f):
        def g(*a, **k):
            t0 = time.perf_counter(); r = f(*a, **k); T[name].append(time.perf_counter() - t0); return r
        return g
    red.finish_weight_step = wrap('finish', of); red.begin_weight_step = wrap('begin', ob); red._launch = wrap('launch', ol)
    for _ in range(10): loop.weight_step(inp, tgt, optimize=False)
    torch.cuda.synchronize()
    for k in T: T[k].clear()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): loop.weight_step(inp, tgt, optimize=False)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    print('force_collectives', force, 'buckets', nb, 'rows', rows, 'inline', inl, 'ms/step', statistics.median(ts), {k: (len(v) / 100.0, 1e6 * sum(v) / 100.0) for k, v in T.items()}, flush=True)
dist.destroy_process_group()

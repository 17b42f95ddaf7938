import sys, time, cProfile, pstats, io, torch
sys.path.insert(0, '/root/repo')
import bench
from mmnas_amd import _lib as L, dp, ops
from mmnas_amd.model import mixed
from mmnas.model.mixed import MixedOp
dev = torch.device('cuda', 0)
cfg = bench.make_cfg('search_vqa')
torch.manual_seed(888); ops.manual_seed(888); mixed.seed_arch_sampler(888)
B, Sx, Sy, V, ANS = 64, 14, 100, 20000, 3129
emb = torch.randn(V, cfg.WORD_EMBED_SIZE, generator=torch.Generator().manual_seed(1)).numpy()
from mmnas.model.hygr_vqa import Net_Search
net = Net_Search(cfg, {'token_size': V, 'ans_size': ANS, 'pretrained_emb': emb}).to(dev).train()
inputs_cpu, target_cpu = bench.synth_batch(cfg, B, Sx, Sy, V, ANS, 888)
inputs = tuple(t.to(dev) for t in inputs_cpu); target = target_cpu.to(dev)
loss_fn = torch.nn.BCEWithLogitsLoss(reduction='sum')
reducer = dp.SupernetReducer(net)
MixedOp.MODE = None
def step():
    net.reset_binary_gates()
    reducer.begin_weight_step()
    loss = loss_fn(net(inputs), target)
    loss.backward()
    reducer.finish_weight_step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print('ms/step %.3f' % ((time.perf_counter() - t0) / 20 * 1e3))
# host time per step without waiting for the GPU
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host-side ms/step (launch only) %.3f' % ((t1 - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35); print(s.getvalue()[:6000])

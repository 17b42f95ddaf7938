"""Tuning aid: mmnas_sumsq + mmnas_adam_step over the supernet's 37 M parameters (HIP events around 20 calls)."""
import sys, torch
sys.path.insert(0, '.')
from mmnas_amd import _lib as L
lib = L.lib()
n = 37_000_000
p, g, m, v = (torch.randn(n, device='cuda') for _ in range(4)); v.abs_()
ss = torch.zeros(1, device='cuda')
def run():
    ss.zero_()
    L.check(lib.mmnas_sumsq(L.fptr(g), n, L.fptr(ss), L.stream()))
    L.check(lib.mmnas_adam_step(L.fptr(p), L.fptr(g), L.fptr(m), L.fptr(v), n, 1e-3, 0.9, 0.98, 1e-9, 0.0, L.fptr(ss), 1.0, 3, L.stream()))
for off in (0, 1):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print('sumsq + adam over %d floats: %.1f us' % (n, e0.elapsed_time(e1) * 1e3 / 20))

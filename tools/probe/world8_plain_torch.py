"""Eight processes on ONE device running plain torch work (no kernel of this library): does the runtime abort them too?
    python tools/probe/world8_plain_torch.py [nproc] [seconds]"""
import os
import sys
import time

import torch
import torch.multiprocessing as mp


def work(rank, seconds, use_lib):
    torch.cuda.set_device(0)
    x = torch.randn(2048, 2048, device='cuda')
    ln = torch.nn.LayerNorm(2048).cuda()
    if use_lib:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
        from mmnas_amd import ops
        a, b = torch.randn(6400, 256, device='cuda'), torch.randn(256, 256, device='cuda')
    mha = None
    if use_lib == 'mha':      # the attention backward kernels (440-452 registers: accumulator registers in use, 152 KB of LDS)
        import ctypes as C
        from mmnas_amd import _lib as L
        B, H, S, dh = 16, 4, 100, 64
        di = H * dh
        Q, dO, K, V = (torch.randn(B, S, di, device='cuda') for _ in range(4))
        O, stats = torch.empty(B, S, di, device='cuda'), torch.empty(B, H, S, 2, device='cuda')
        dQ, dK, dV, delta = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V), torch.empty(B, H, S, device='cuda')
        d = L.MhaDesc()
        d.B, d.H, d.Sq, d.Sk, d.dh = B, H, S, S, dh
        d.ldq = d.ldk = d.ldv = d.ldo = di
        d.Q, d.K, d.V, d.O, d.lse = L.fptr(Q), L.fptr(K), L.fptr(V), L.fptr(O), L.fptr(stats)
        d.dO, d.dQ, d.dK, d.dV, d.delta = L.fptr(dO), L.fptr(dQ), L.fptr(dK), L.fptr(dV), L.fptr(delta)
        mha = (L, C, d, (Q, dO, K, V, O, stats, dQ, dK, dV, delta))
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            y = ln(x @ x).relu()
            if use_lib:
                z = ops.linear(a, b)
            if mha:
                L, C, d, _keep = mha
                L.check(L.lib().mmnas_mha_core_fwd(C.byref(d), L.stream()))
                L.check(L.lib().mmnas_mha_core_bwd(C.byref(d), L.stream()))
        torch.cuda.synchronize()
        n += 1
    print('rank %d ok (%d rounds)' % (rank, n), flush=True)


if __name__ == '__main__':
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    use_lib = sys.argv[3] if len(sys.argv) > 3 else ''
    mp.spawn(work, args=(seconds, use_lib), nprocs=nproc, join=True)

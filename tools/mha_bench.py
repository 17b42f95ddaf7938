"""Tuning aid: times mmnas_mha_core_fwd / _bwd on the workloads' attention shapes (HIP events around 50 launches).

    python tools/mha_bench.py            # MMNAS_MHA_BWD_FUSED8 / MMNAS_MHA_BWD_FUSED select the backward kernel
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import _lib as L  # noqa: E402


def main():
    dev = 'cuda'
    lib = L.lib()
    for (B, H, Sq, Sk, bias) in [(64, 4, 100, 100, False), (64, 4, 100, 100, True), (64, 8, 100, 100, False), (64, 8, 100, 100, True),
                                 (64, 4, 100, 14, False), (64, 8, 100, 14, False), (64, 4, 14, 14, False)]:
        dh = 64
        di = H * dh
        Q, dO = torch.randn(B, Sq, di, device=dev), torch.randn(B, Sq, di, device=dev)
        K, V = torch.randn(B, Sk, di, device=dev), torch.randn(B, Sk, di, device=dev)
        mask = torch.zeros(B, Sk, dtype=torch.uint8, device=dev)
        bT = torch.randn(B, H, Sk, Sq, device=dev) if bias else None
        O, stats = torch.empty(B, Sq, di, device=dev), torch.empty(B, H, Sq, 2, device=dev)
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
        dbT = torch.empty(B, H, Sk, Sq, device=dev) if bias else None
        delta = torch.empty(B, H, Sq, device=dev)
        d = L.MhaDesc()
        d.B, d.H, d.Sq, d.Sk, d.dh = B, H, Sq, Sk, dh
        d.ldq = d.ldk = d.ldv = d.ldo = di
        d.Q, d.K, d.V, d.mask, d.biasT, d.O, d.lse = L.fptr(Q), L.fptr(K), L.fptr(V), L.ptr(mask), L.fptr(bT), L.fptr(O), L.fptr(stats)
        d.drop_p, d.drop_site, d.drop_seed = 0.1, 0, 1234
        d.dO, d.dQ, d.dK, d.dV, d.dbiasT, d.delta = L.fptr(dO), L.fptr(dQ), L.fptr(dK), L.fptr(dV), L.fptr(dbT), L.fptr(delta)
        out = []
        for name, fn in (('fwd', lib.mmnas_mha_core_fwd), ('bwd', lib.mmnas_mha_core_bwd)):
            for _ in range(5):
                L.check(fn(C.byref(d), L.stream()))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn(C.byref(d), L.stream())
            e1.record()
            torch.cuda.synchronize()
            out.append('%s %6.1f us' % (name, e0.elapsed_time(e1) * 1e3 / 50))
        print('B=%d H=%d Sq=%3d Sk=%3d bias=%d | %s' % (B, H, Sq, Sk, bias, ' | '.join(out)))


if __name__ == '__main__':
    main()

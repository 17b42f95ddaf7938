"""Tuning aid: times mmnas_rel_fused_fwd / _bwd on the workloads' relation-bias shapes (HIP events around 30 launches).

    python tools/rel_bench.py            # MMNAS_REL_BWD_VALU=0 selects the all-MFMA backward kernel
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import _lib as L  # noqa: E402


def main():
    dev = 'cuda'
    lib = L.lib()
    for (B, S, C, H) in [(64, 100, 4, 4), (64, 100, 4, 8), (64, 14, 3, 4), (160, 36, 4, 8)]:
        raw = torch.randn(B, S, S, C, device=dev)
        Wy, by = torch.randn(64, C, device=dev) * 0.5, torch.randn(64, device=dev) * 0.1
        Wr, br = torch.randn(H, 64, device=dev) * 0.2, torch.rand(H, device=dev)
        bias = torch.empty(B, H, S, S, device=dev)
        dbias = torch.randn(B, H, S, S, device=dev)
        dWy, dby, dWr, dbr = torch.zeros_like(Wy), torch.zeros_like(by), torch.zeros_like(Wr), torch.zeros_like(br)
        ws = torch.empty(lib.mmnas_rel_fused_bwd_ws_floats(B, S, S), device=dev)
        fwd = lambda: lib.mmnas_rel_fused_fwd(L.fptr(raw), L.fptr(Wy), L.fptr(by), L.fptr(Wr), L.fptr(br), L.fptr(bias), B, S, S, C, 64, H, L.stream())
        bwd = lambda: lib.mmnas_rel_fused_bwd(L.fptr(raw), L.fptr(Wy), L.fptr(by), L.fptr(Wr), L.fptr(br), L.fptr(dbias), L.fptr(dWy), L.fptr(dby),
                                              L.fptr(dWr), L.fptr(dbr), L.fptr(ws), B, S, S, C, 64, H, L.stream())
        out = []
        for name, fn in (('fwd', fwd), ('bwd', bwd)):
            for _ in range(5):
                L.check(fn())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                fn()
            e1.record()
            torch.cuda.synchronize()
            out.append('%s %6.1f us' % (name, e0.elapsed_time(e1) * 1e3 / 30))
        print('B=%d S=%3d C=%d H=%d | %s  (bwd includes the 5 us partial-row reduction launch)' % (B, S, C, H, ' | '.join(out)))
        # all relation operators of a stream in one launch per direction (relmulti.hip)
        import ctypes as C_
        for n_ops in (1, 6, 8, 18):
            Wrs = [torch.randn(H, 64, device=dev) * 0.2 for _ in range(n_ops)]
            brs = [torch.rand(H, device=dev) for _ in range(n_ops)]
            biases = [torch.empty(B, H, S, S, device=dev) for _ in range(n_ops)]
            dbs = [torch.randn(B, H, S, S, device=dev) for _ in range(n_ops)]
            dWrs, dbrs = [torch.zeros(H, 64, device=dev) for _ in range(n_ops)], [torch.zeros(H, device=dev) for _ in range(n_ops)]
            wsm = torch.empty(lib.mmnas_rel_multi_bwd_ws_floats(B, S), device=dev)
            m = L.RelMulti()
            m.B, m.S, m.C, m.R, m.H, m.n_ops = B, S, C, 64, H, n_ops
            m.raw, m.Wy, m.by, m.dWy, m.dby, m.ws = L.fptr(raw), L.fptr(Wy), L.fptr(by), L.fptr(dWy), L.fptr(dby), L.fptr(wsm)
            for i in range(n_ops):
                m.Wr[i], m.br[i], m.biasT[i], m.dbiasT[i], m.dWr[i], m.dbr[i] = (L.fptr(t) for t in (Wrs[i], brs[i], biases[i], dbs[i], dWrs[i], dbrs[i]))
            out = []
            for name, fn in (('fwd', lambda: lib.mmnas_rel_multi_fwd(C_.byref(m), L.stream())), ('bwd', lambda: lib.mmnas_rel_multi_bwd(C_.byref(m), L.stream()))):
                for _ in range(5):
                    L.check(fn())
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                out.append('%s %6.1f us' % (name, e0.elapsed_time(e1) * 1e3 / 30))
            print('    rel_multi, %2d operators in one call | %s' % (n_ops, ' | '.join(out)))


if __name__ == '__main__':
    main()

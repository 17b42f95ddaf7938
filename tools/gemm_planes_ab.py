"""A/B of the forward (NT) products: weight operand staged through registers with the in-kernel split (the default,
two register stages; and MMNAS_GEMM_PF=1, the loop shape the LDS-DMA kernel has) against weight operand as pre-split
bf16 planes streamed global -> LDS by LDS-DMA (mmnas_gemm_desc.b_planes).  Same process, interleaved launches, the
results compared bit for bit first.

    python tools/gemm_planes_ab.py [--reps 200] > profiles/r04_gemm_planes_ab.txt
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import _lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=200)
    ap.add_argument('--ksweep', action='store_true', help='N = 256, M = 4096 w (w workgroups per CU), K = 64..2048: cost per 32-deep K-tile '
                                                          '(slope) and fixed part (intercept) of A, A1, B')
    args = ap.parse_args()
    lib = L.lib()
    dev = 'cuda'
    torch.manual_seed(0)
    shapes = []
    for d in (256, 512):
        for M in (6400, 896):
            shapes += [([M], d, d), ([M] * 3, d, d), ([M], 4 * d, d), ([M], d, 4 * d)]
        shapes += [([6400], d, 2048)]
    if args.ksweep:
        return ksweep(args, lib)
    print('# NT products C = A W^T, A [M,K] fp32, W [N,K]; us per launch (median of 5 blocks of %d back-to-back launches), TF/s algorithmic' % args.reps)
    print('# A = default (in-kernel split, PF=2)   A1 = in-kernel split, PF=1   B = W as pre-split bf16 planes by LDS-DMA (PF=1 loop)')
    for Ms, N, K in shapes:
        As = [torch.randn(M, K, device=dev) for M in Ms]
        Ws = [torch.randn(N, K, device=dev) * 0.05 for _ in Ms]
        Ps = [ops.split_planes(W) for W in Ws]
        bias = torch.randn(N, device=dev)
        Cs = [[torch.empty(M, N, device=dev) for M in Ms] for _ in range(3)]

        def desc(which):
            planes = which == 2
            return ops.gemm_desc(L.GEMM_NT, [dict(M=M, A=[a], B=[(p if planes else w)], C=c, bias=bias)
                                             for M, a, w, p, c in zip(Ms, As, Ws, Ps, Cs[which])], N, K, K, K, N, b_planes=planes)
        descs = [desc(0), desc(1), desc(2)]
        st = L.stream()

        def launch(which):
            L.check(lib.mmnas_gemm(C.byref(descs[which]), st))

        def set_pf(pf):
            os.environ['MMNAS_GEMM_PF'] = str(pf)
            L.check(lib.mmnas_gemm_reload_tuning())
        set_pf(2); launch(0)
        set_pf(1); launch(1)
        launch(2)
        torch.cuda.synchronize()
        for c0, c1, c2 in zip(*Cs):
            assert torch.equal(c0, c2) and torch.equal(c1, c2), 'planes path differs from the in-kernel split'
        times = [[], [], []]
        for _ in range(5):
            for which, pf in ((0, 2), (1, 1), (2, 1)):
                set_pf(pf)
                for _ in range(10):
                    launch(which)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    launch(which)
                e1.record()
                torch.cuda.synchronize()
                times[which].append(1e3 * e0.elapsed_time(e1) / args.reps)
        set_pf(2)
        fl = 2.0 * sum(Ms) * N * K
        med = [sorted(t)[2] for t in times]
        print('NT M=%-20s N=%-5d K=%-5d | A %7.1f us %6.1f TF | A1 %7.1f us %6.1f TF | B %7.1f us %6.1f TF | B/A %.3f  B/A1 %.3f'
              % (Ms, N, K, med[0], fl / med[0] / 1e6, med[1], fl / med[1] / 1e6, med[2], fl / med[2] / 1e6, med[2] / med[0], med[2] / med[1]))
        sys.stdout.flush()


def ksweep(args, lib):
    import numpy as np
    dev = 'cuda'
    N = 256
    print('# K sweep, N = 256, M = 4096 w: every CU holds exactly w workgroups of 64^2; slope = us per 32-deep K-tile round (cycles at 2.4 GHz), '
          'intercept = fixed part of a launch.  Back-to-back launches, median of 5 blocks of %d' % args.reps)
    for w in (1, 2, 3):
        M = 4096 * w
        res = {0: [], 1: [], 2: []}
        Ks = (64, 128, 256, 512, 1024, 2048)
        for K in Ks:
            a = torch.randn(M, K, device=dev)
            W = torch.randn(N, K, device=dev) * 0.05
            P = ops.split_planes(W)
            c = torch.empty(M, N, device=dev)
            ds = [ops.gemm_desc(L.GEMM_NT, [dict(M=M, A=[a], B=[(P if pl else W)], C=c)], N, K, K, K, N, b_planes=pl) for pl in (False, False, True)]
            st = L.stream()
            for which, pf in ((0, 2), (1, 1), (2, 1)):
                os.environ['MMNAS_GEMM_PF'] = str(pf)
                L.check(lib.mmnas_gemm_reload_tuning())
                ts = []
                for _ in range(5):
                    for _ in range(10):
                        L.check(lib.mmnas_gemm(C.byref(ds[which]), st))
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.reps):
                        L.check(lib.mmnas_gemm(C.byref(ds[which]), st))
                    e1.record()
                    torch.cuda.synchronize()
                    ts.append(1e3 * e0.elapsed_time(e1) / args.reps)
                res[which].append(sorted(ts)[2])
        for which, name in ((0, 'A  in-kernel split, PF=2'), (1, 'A1 in-kernel split, PF=1'), (2, 'B  weight planes by LDS-DMA')):
            sl, ic = np.polyfit(np.array(Ks) / 32.0, np.array(res[which]), 1)
            print('%d workgroups/CU (M=%5d) %-28s: %s | per K-tile %.3f us = %4.0f cycles, intercept %.1f us'
                  % (w, M, name, '  '.join('K=%d %.1f' % (k, t) for k, t in zip(Ks, res[which])), sl, sl * 2400, ic))
        sys.stdout.flush()
    os.environ['MMNAS_GEMM_PF'] = '2'
    L.check(lib.mmnas_gemm_reload_tuning())


if __name__ == '__main__':
    main()

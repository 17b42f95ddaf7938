"""A/B of the lean whole-tile GEMM kernels (gemm_body<..., LEAN>: short set-up, transposed accumulators, 16-byte epilogue
rows) against the general kernel on the supernet's and the training step's product shapes, with the epilogues the operators
use.  Results of the two are asserted equal bit for bit before timing (MMNAS_GEMM_LEAN=0 selects the general kernel).

    python tools/gemm_lean_ab.py            # us per launch, median of 5 blocks of 200 back-to-back launches
"""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import ops  # noqa: E402
import mmnas_amd._lib as L  # noqa: E402

DEV = 'cuda'


def set_lean(v):
    os.environ['MMNAS_GEMM_LEAN'] = str(v)
    L.check(L.lib().mmnas_gemm_reload_tuning())


def make(layout, Ms, N, K, epi, nseg=1):
    g = torch.Generator(device=DEV).manual_seed(1)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)   # noqa: E731
    groups = []
    for M in Ms:
        A = [r(K, M) if layout == 'TN' else r(M, K) for _ in range(nseg)]
        B = [r(N, K) if layout == 'NT' else r(K, N) for _ in range(nseg)]
        grp = dict(M=M, A=A, B=B, C=torch.zeros(M, N, device=DEV))
        if 'b' in epi:
            grp['bias'] = r(N)
        if 'r' in epi:
            grp['residual'] = r(M, N)
        if 'g' in epi:
            grp['gate'] = r(M, N)
        if 'c' in epi:
            grp['colsum'] = torch.zeros(N, device=DEV)
        groups.append(grp)
    kw = dict(nseg=nseg)
    if 'R' in epi:
        kw['relu'] = True
    if 'd' in epi:
        kw['drop'] = (0.1, 1234, 7)
    if 'g' in epi:
        kw.update(gate_scale=1.0 / 0.9, ldgate=N)
    if 'r' in epi:
        kw['ldres'] = N
    if 'a' in epi:
        kw['accumulate'] = True
    lay = {'NT': L.GEMM_NT, 'NN': L.GEMM_NN, 'TN': L.GEMM_TN}[layout]
    if layout == 'TN':
        d = ops.gemm_desc(lay, groups, N, K, Ms[0], N, N, **kw)
    else:
        d = ops.gemm_desc(lay, groups, N, K, K, K if layout == 'NT' else N, N, **kw)
    return d, groups


def time_desc(d, blocks=5, n=200, d2=None):
    lib, st = L.lib(), L.stream()
    if d2 is not None:   # a gradient pair: data gradient d, weight gradient d2, one launch
        call = lambda: lib.mmnas_gemm_pair(C.byref(d), C.byref(d2), st)   # noqa: E731
    else:
        call = lambda: lib.mmnas_gemm(C.byref(d), st)   # noqa: E731
    for _ in range(10):
        L.check(call())
    torch.cuda.synchronize()
    out = []
    for _ in range(blocks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            call()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / n)
    return statistics.median(out)


def main():
    cases = [
        ('NT', [6400], 256, 256, ''), ('NT', [6400] * 3, 256, 256, ''), ('NT', [6400], 256, 256, 'dr'),
        ('NT', [6400], 1024, 256, 'bRd'), ('NT', [6400], 256, 1024, 'bdr'),
        ('NT', [896], 256, 256, ''), ('NT', [896] * 3, 256, 256, ''), ('NT', [896], 1024, 256, 'bRd'),
        ('NN', [6400], 256, 256, 'r'), ('NN', [6400], 1024, 256, 'g'), ('NN', [6400], 256, 256, '', 3), ('NN', [6400], 256, 1024, 'r'),
        ('NN', [896], 256, 256, 'r'), ('NN', [6400], 256, 256, 'a'), ('NN', [6400], 1024, 256, 'gc'),
        ('NT', [6400], 512, 512, ''), ('NT', [6400] * 3, 512, 512, ''), ('NT', [6400], 512, 512, 'dr'), ('NT', [6400], 2048, 512, 'bRd'),
        ('NN', [6400], 512, 512, 'r'),
        ('NT', [6397], 256, 256, 'bdr'), ('NT', [100, 6400, 37], 256, 256, 'r'),
        ('TN', [256], 256, 6400, 'a'), ('TN', [256] * 3, 256, 6400, 'a'), ('TN', [1024], 256, 6400, 'a'), ('TN', [256], 1024, 6400, 'a'),
        ('TN', [256], 256, 896, 'a'), ('TN', [512], 512, 6400, 'a'), ('TN', [256], 256, 3517, 'a'),
    ]
    for c in cases:
        layout, Ms, N, K, epi = c[:5]
        nseg = c[5] if len(c) > 5 else 1
        res = {}
        for lean in (0, 3):
            set_lean(lean)
            d, groups = make(layout, Ms, N, K, epi, nseg)
            if 'a' in epi:
                for g in groups:
                    g['C'].fill_(0.5)
            L.check(L.lib().mmnas_gemm(C.byref(d), L.stream()))
            torch.cuda.synchronize()
            outs = [g['C'].clone() for g in groups]
            t = time_desc(d)
            res[lean] = (outs, t)
        same = all(torch.equal(a, b) for a, b in zip(res[0][0], res[3][0]))
        err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(res[0][0], res[3][0]))
        if layout == 'TN':   # float atomics: the order of the adds is not fixed
            assert err < 1e-5, err
        flops = 2.0 * sum(Ms) * N * K * nseg
        print('%s M=%-22s N=%-5d K=%-5d seg=%d epi=%-4s | general %6.1f us %6.1f TF | lean %6.1f us %6.1f TF | lean/general %.3f | %s'
              % (layout, Ms, N, K, nseg, epi or '-', res[0][1], flops / res[0][1] * 1e-6, res[3][1], flops / res[3][1] * 1e-6,
                 res[3][1] / res[0][1], 'bit-equal' if same else 'max rel diff %.3g' % err), flush=True)
    # gradient pairs (one launch): dX = dY W (NN, + residual) and dW += dY^T X (TN)
    for (M, N, K, epi) in [(6400, 256, 256, 'r'), (6400, 256, 256, ''), (6400, 1024, 256, 'gc'), (6400, 256, 1024, 'r'), (896, 256, 256, 'r'),
                           (6400, 512, 512, 'r')]:
        res = {}
        for lean in (0, 3):
            set_lean(lean)
            dg, g0 = make('NN', [M], N, K, epi)                  # dX [M,N] = dY [M,K] W [K,N]
            wg, g1 = make('TN', [K], N, M, 'a')                  # dW [K,N] += dY^T [K rows of [M,K]^T] X [M,N]
            L.check(L.lib().mmnas_gemm_pair(C.byref(dg), C.byref(wg), L.stream()))
            torch.cuda.synchronize()
            outs = [g0[0]['C'].clone(), g1[0]['C'].clone()]
            res[lean] = (outs, time_desc(dg, d2=wg))
        err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(res[0][0], res[3][0]))
        assert err < 1e-5, err
        flops = 4.0 * M * N * K
        print('PAIR M=%d N=%d K=%d epi=%-2s | general %6.1f us %6.1f TF | lean %6.1f us %6.1f TF | lean/general %.3f | max rel diff %.2g'
              % (M, N, K, epi or '-', res[0][1], flops / res[0][1] * 1e-6, res[3][1], flops / res[3][1] * 1e-6, res[3][1] / res[0][1], err), flush=True)
    set_lean(3)


if __name__ == '__main__':
    main()

"""A/B of one mmnas_gemm tuning knob inside ONE build (tuning aid): the workloads' product shapes timed with the
environment variable set to each value in turn (mmnas_gemm_reload_tuning re-reads it), launches interleaved.

    python tools/gemm_knob.py MMNAS_GEMM_PF 1 2
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import _lib as L, ops  # noqa: E402


def main():
    knob, vals = sys.argv[1], sys.argv[2:]
    lib = L.lib()
    dev = 'cuda'
    shapes = []
    for d in (256, 512):
        for M in (6400, 896):
            shapes += [('NT', [M], d, d, 1), ('NT', [M] * 3, d, d, 1), ('NT', [M], 4 * d, d, 1), ('NT', [M], d, 4 * d, 1),
                       ('NN', [M], d, d, 1), ('NN', [M], d, d, 3), ('NN', [M], 4 * d, d, 1), ('NN', [M], d, 4 * d, 1),
                       ('TN', [d], d, M, 1), ('TN', [d] * 3, d, M, 1), ('TN', [4 * d], d, M, 1), ('TN', [d], 4 * d, M, 1)]
        shapes += [('NT', [6400, 896, 896], d, d, 1)]
    shapes += [('NT', [6400], 256, 2048, 1), ('NT', [6400], 512, 2048, 1)]
    for layout, Ms, N, K, nseg in shapes:
        lay = {'NT': L.GEMM_NT, 'NN': L.GEMM_NN, 'TN': L.GEMM_TN}[layout]
        groups, keep = [], []
        for M in Ms:
            if layout == 'NT':
                sa, sb, lda, ldb = (M, K), (N, K), K, K
            elif layout == 'NN':
                sa, sb, lda, ldb = (M, K), (K, N), K, N
            else:
                sa, sb, lda, ldb = (K, M), (K, N), M, N
            As = [torch.randn(*sa, device=dev) for _ in range(nseg)]
            Bs = [torch.randn(*sb, device=dev) for _ in range(nseg)]
            c = torch.zeros(M, N, device=dev)
            keep += As + Bs + [c]
            groups.append(dict(M=M, A=As, B=Bs, C=c))
        g = ops.gemm_desc(lay, groups, N, K, lda, ldb, N, nseg=nseg, accumulate=(layout == 'TN'))
        import ctypes as C
        res = {v: [] for v in vals}
        for rep in range(3):
            for v in vals:
                os.environ[knob] = v
                lib.mmnas_gemm_reload_tuning()
                for _ in range(3):
                    L.check(lib.mmnas_gemm(C.byref(g), L.stream()))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    lib.mmnas_gemm(C.byref(g), L.stream())
                e1.record()
                torch.cuda.synchronize()
                res[v].append(e0.elapsed_time(e1) * 1e3 / 30)
        flops = 2.0 * sum(Ms) * N * K * nseg
        best = {v: min(r) for v, r in res.items()}
        print('%-3s M=%-18s N=%-5d K=%-5d seg=%d | ' % (layout, Ms, N, K, nseg) +
              ' | '.join('%s=%s: %6.1f us %6.1f TF' % (knob[-6:], v, best[v], flops / best[v] / 1e6) for v in vals) +
              ' | last/first %.2f' % (best[vals[-1]] / best[vals[0]]), flush=True)


if __name__ == '__main__':
    main()

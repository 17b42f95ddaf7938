"""Microbenchmark of the dense sequence convolution (StdConv core, modules.py:472,480-481): the window-buffer-free
product (ops.ConvSeqFn: the zero-padded input read with overlapping rows, lda = d, K = k d)
against the explicit-window form it replaces (im2col -> one product; backward: product -> col2im).

    python tools/conv_bench.py > profiles/r04_conv_microbench.txt

Algorithmic work: forward 2 B S d^2 k flop; forward + backward 3x.  Minimum bytes forward: x and y once +
the weights = 4 (2 B S d + k d^2).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import ops  # noqa: E402


def timed(fn, reps):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / reps)
    return sorted(ts)[2]


def main():
    torch.manual_seed(0)
    dev = 'cuda'
    print('# StdConv core, B = 64.  us per call (median of 5 blocks of 30 calls); TF/s on the algorithmic flops 2 B S d^2 k (x3 with backward)')
    print('# seg = product on the padded input with overlapping rows (default)   i2c = im2col + product (MMNAS_CONV_IM2COL=1)   extra bytes: what each form moves beyond '
          'the minimum 4 (2 B S d + k d^2) forward')
    for S, d in ((100, 512), (100, 256), (14, 512)):
        for k in (3, 5, 7, 11):
            B = 64
            x = torch.randn(B, S, d, device=dev, requires_grad=True)
            w = (torch.randn(d, d, k, device=dev) * 0.05).requires_grad_(True)
            b = torch.zeros(d, device=dev, requires_grad=True)
            dy = torch.randn(B, S, d, device=dev)
            res = {}
            for mode in ('seg', 'i2c'):
                os.environ['MMNAS_CONV_IM2COL'] = '1' if mode == 'i2c' else '0'

                def fwd():
                    with torch.no_grad():
                        return ops.conv_seq(x, w, b)

                def fwdbwd():
                    y = ops.conv_seq(x, w, b)
                    torch.autograd.grad(y, (x, w, b), dy)
                res[mode] = (timed(fwd, 30), timed(fwdbwd, 30))
            fl = 2.0 * B * S * d * d * k
            minb = 4.0 * (2 * B * S * d + k * d * d)
            pad = k // 2
            seg_extra = 4.0 * (2 * B * (S + 2 * pad) * d + 2 * B * (S + 2 * pad) * d)   # padded copy of x written + read, padded y written + read back
            i2c_extra = 4.0 * (2 * B * S * k * d)                                          # the window buffer written + read
            print('S=%-3d d=%-3d k=%-2d | fwd: seg %7.1f us %6.1f TF  i2c %7.1f us %6.1f TF  x%.2f | fwd+bwd: seg %7.1f us %6.1f TF  i2c %7.1f us %6.1f TF  x%.2f '
                  '| min fwd bytes %.1f MB, extra seg %.1f MB, extra i2c %.1f MB'
                  % (S, d, k, res['seg'][0], fl / res['seg'][0] / 1e6, res['i2c'][0], fl / res['i2c'][0] / 1e6, res['i2c'][0] / res['seg'][0],
                     res['seg'][1], 3 * fl / res['seg'][1] / 1e6, res['i2c'][1], 3 * fl / res['i2c'][1] / 1e6, res['i2c'][1] / res['seg'][1],
                     minb / 1e6, seg_extra / 1e6, i2c_extra / 1e6))
            sys.stdout.flush()


if __name__ == '__main__':
    main()

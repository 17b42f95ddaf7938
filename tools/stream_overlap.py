"""Does running independent GEMM launches on several HIP streams shorten a latency-bound sequence?  (tuning aid)

The supernet's products are single-round launches (200-2000 workgroups, K <= 1024): a launch is ramp + 8-32 K-tiles +
drain and the next launch of the stream waits for the last workgroup.  This script times the same set of launches
(a) back to back on one stream and (b) dealt round-robin onto 2 / 4 streams, everything captured into one HIP graph per
variant so the host is out of the picture.  One line per case: microseconds for the set and the ratio to one stream.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import ops  # noqa: E402
import mmnas_amd._lib as L  # noqa: E402

DEV = 'cuda'
keep = []


def T(*shape):
    t = torch.randn(*shape, device=DEV)
    keep.append(t)
    return t


def nt(M, N, K):
    return ('one', ops.gemm_desc(L.GEMM_NT, [dict(M=M, A=[T(M, K)], B=[T(N, K)], C=T(M, N))], N, K, K, K, N))


def nn(M, N, K):   # dX[M, K] = dY[M, N] W[N, K]
    return ('one', ops.gemm_desc(L.GEMM_NN, [dict(M=M, A=[T(M, N)], B=[T(N, K)], C=T(M, K))], K, N, N, K, K))


def tn(M, N, K):   # dW[N, K] += dY[M, N]^T X[M, K]
    return ('one', ops.gemm_desc(L.GEMM_TN, [dict(M=N, A=[T(M, N)], B=[T(M, K)], C=T(N, K))], K, M, N, K, K, accumulate=True))


def pair(M, N, K):
    return ('pair', nn(M, N, K)[1], tn(M, N, K)[1])


def issue(item):
    if item[0] == 'one':
        L.check(L.lib().mmnas_gemm(C.byref(item[1]), L.stream()))
    else:
        L.check(L.lib().mmnas_gemm_pair(C.byref(item[1]), C.byref(item[2]), L.stream()))


def timed(lanes, iters=20):
    """lanes: list of lists of launches; lane i runs in order on stream i, the lanes run concurrently."""
    main = torch.cuda.Stream()
    side = [torch.cuda.Stream() for _ in lanes[1:]]

    def body():
        for s in side:
            s.wait_stream(torch.cuda.current_stream())
        for lane, s in zip(lanes, [None] + side):
            if s is None:
                for it in lane:
                    issue(it)
            else:
                with torch.cuda.stream(s):
                    for it in lane:
                        issue(it)
        for s in side:
            torch.cuda.current_stream().wait_stream(s)

    with torch.cuda.stream(main):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=main):
                body()
            run = g.replay
        except Exception as e:   # capture refused: time the eager issue instead
            print('  (graph capture failed: %s; eager timing)' % str(e)[:80])
            run = body
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def deal(items, n):
    return [items[i::n] for i in range(n)]


def case(name, items, lanes_list=(2, 4)):
    t1 = timed([items])
    out = ['%-58s 1 stream %8.1f us (%5.1f / launch)' % (name, t1, t1 / len(items))]
    for n in lanes_list:
        tn_ = timed(deal(items, n))
        out.append('%d streams %8.1f us (x%.2f)' % (n, tn_, tn_ / t1))
    print(' | '.join(out), flush=True)


if __name__ == '__main__':
    for d in (256, 512):
        M, Mt = 6400, 896
        print('--- d = %d' % d)
        case('16 x NT %dx%dx%d' % (M, d, d), [nt(M, d, d) for _ in range(16)])
        case('16 x NT %dx%dx%d' % (M, 4 * d, d), [nt(M, 4 * d, d) for _ in range(16)])
        case('16 x NT %dx%dx%d' % (M, d, 4 * d), [nt(M, d, 4 * d) for _ in range(16)])
        case('16 x NT %dx%dx%d (text side)' % (Mt, d, d), [nt(Mt, d, d) for _ in range(16)])
        case('16 x pair(NN + TN) %dx%dx%d' % (M, d, d), [pair(M, d, d) for _ in range(16)])
        case('16 x pair(NN + TN) %dx%dx%d' % (M, 4 * d, d), [pair(M, 4 * d, d) for _ in range(16)])
        # the same work with the weight gradients as their own launches on the second stream
        nns, tns = [nn(M, d, d) for _ in range(16)], [tn(M, d, d) for _ in range(16)]
        t_pair = timed([[pair(M, d, d) for _ in range(16)]])
        t_seq = timed([nns + tns])
        t_two = timed([nns, tns])
        print('%-58s pairs %8.1f us | NN then TN one stream %8.1f us | NN || TN two streams %8.1f us (x%.2f of pairs)'
              % ('16 x (NN, TN) %dx%dx%d' % (M, d, d), t_pair, t_seq, t_two, t_two / t_pair), flush=True)
        nns, tns = [nn(M, 4 * d, d) for _ in range(16)], [tn(M, 4 * d, d) for _ in range(16)]
        t_pair = timed([[pair(M, 4 * d, d) for _ in range(16)]])
        t_two = timed([nns, tns])
        print('%-58s pairs %8.1f us | NN || TN two streams %8.1f us (x%.2f of pairs)'
              % ('16 x (NN, TN) %dx%dx%d' % (M, 4 * d, d), t_pair, t_two, t_two / t_pair), flush=True)
        # image-side chain with the text-side chain beside it
        img, txt = [nt(M, d, d) for _ in range(16)], [nt(Mt, d, d) for _ in range(16)]
        ti, tt, tb = timed([img]), timed([txt]), timed([img, txt])
        print('%-58s image %8.1f us, text %8.1f us, one after the other %8.1f us | side by side %8.1f us'
              % ('16 x NT image chain || 16 x NT text chain', ti, tt, ti + tt, tb), flush=True)

"""Cycle stamps of one wave through the phases of the split-operand K loop (needs a -DMMNAS_DBG_STAMP=<workgroup> build of
gemm.hip loaded through MMNAS_LIB_PATH; tuning aid).  Prints cycles per K-tile pair spent in each phase of the first half-iteration."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmnas_amd import ops  # noqa: E402
import mmnas_amd._lib as L  # noqa: E402

NAMES = ['second half-iteration (whole)', 'load issue', 'fragment reads + MFMA issue', 'wait for the next tile\'s loads',
         'conversion + LDS store issue', 'LDS stores landed', 'barrier']


def run(M, N, K, launches=20):
    lib = C.CDLL(L.LIB_PATH)
    a, b, c = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda'), torch.zeros(M, N, device='cuda')
    d = ops.gemm_desc(L.GEMM_NT, [dict(M=M, A=[a], B=[b], C=c)], N, K, K, K, N)
    out = (C.c_ulonglong * 8)()
    for _ in range(3):
        L.check(L.lib().mmnas_gemm(C.byref(d), L.stream()))
    torch.cuda.synchronize()
    assert lib.mmnas_dbg_stamps(out, 1) == 0
    assert lib.mmnas_dbg_stamps(out, 3) == 0
    for _ in range(launches):
        L.check(L.lib().mmnas_gemm(C.byref(d), L.stream()))
    torch.cuda.synchronize()
    assert lib.mmnas_dbg_stamps(out, 1) == 0
    pairs = launches * max(K // 64, 1)
    print('NT %d x %d x %d (%d workgroups): cycles per pair of K-tiles' % (M, N, K, (M // 64) * (N // 64)))
    tot = 0
    for i, n in enumerate(NAMES):
        print('   %-40s %7.0f' % (n, out[i] / pairs))
        tot += out[i] / pairs
    print('   %-40s %7.0f' % ('sum', tot))
    assert lib.mmnas_dbg_stamps(out, 2) == 0
    life = ['kernel-argument warm-up', 'tile / group / offset set-up', 'first tile: loads -> conversion -> LDS -> barrier', 'K loop', 'epilogue, stores landed']
    print('   workgroup lifetime, cycles per launch: ' + ', '.join('%s %.0f' % (n, out[i] / launches) for i, n in enumerate(life))
          + '  (total %.0f)' % (sum(out[i] for i in range(5)) / launches))


if __name__ == '__main__':
    for w in (1, 2, 3):
        run(4096 * w, 256, 2048)
    run(64 * 38, 64, 32)
    run(6400, 256, 32)
    run(6400, 256, 256)
    run(6400, 1024, 256)

"""A/B microbenchmark of mmnas_gemm between two builds of the library (tuning aid).

    python tools/gemm_ab.py mmnas_amd/lib/libmmnas_hip_old.so mmnas_amd/lib/libmmnas_hip.so

The first library is driven with the descriptor layout of ABI revision 0 (no accumulate field, split_k
chosen by the caller as the operator layer used to), the second with the current one.  Launches are
interleaved so clock / box differences cancel.
"""
import ctypes as C
import os
import sys

import torch

_fp = C.c_void_p


def group_type(current):
    class Group(C.Structure):   # the current ABI has the colsum epilogue pointer
        _fields_ = [('M', C.c_int), ('A', _fp * 3), ('B', _fp * 3), ('C', _fp), ('bias', _fp), ('residual', _fp), ('gate', _fp)] + \
                   ([('colsum', _fp)] if current else [])
    return Group


def desc_type(with_acc):
    mid = [('accumulate', C.c_int), ('reserved', C.c_int)] if with_acc else []
    Group = group_type(with_acc)

    class Desc(C.Structure):
        _fields_ = [('layout', C.c_int), ('ngroups', C.c_int), ('nseg', C.c_int), ('N', C.c_int), ('K', C.c_int),
                    ('lda', C.c_int), ('ldb', C.c_int), ('ldc', C.c_int), ('ldres', C.c_int), ('ldgate', C.c_int),
                    ('relu', C.c_int), ('split_k', C.c_int)] + mid + [
                    ('alpha', C.c_float), ('gate_scale', C.c_float), ('drop_p', C.c_float), ('drop_site', C.c_uint32),
                    ('drop_seed', C.c_uint64), ('g', Group * 3)]
    return Desc


def old_split(M, N, groups, K):
    tiles = ((M + 63) // 64) * ((N + 63) // 64) * groups
    return max(1, min((1024 + tiles - 1) // tiles, max(K // 128, 1), 1024))


def main():
    # a path prefixed with "old:" is driven with the revision-0 descriptor (no accumulate field, caller-chosen split_k)
    # ("r2:": the round-2 descriptor -- three groups, no per-group dropout seed)
    paths = sys.argv[1:3]
    is_old = [p.startswith('old:') for p in paths]
    is_r2 = [p.startswith('r2:') for p in paths]
    libs = [C.CDLL(os.path.abspath(p.split(':', 1)[1] if (o or r) else p)) for p, o, r in zip(paths, is_old, is_r2)]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mmnas_amd._lib import GemmDesc   # the current descriptor (include/mmnas_hip.h)
    types = [desc_type(False) if o else (desc_type(True) if r else GemmDesc) for o, r in zip(is_old, is_r2)]
    dev = 'cuda'
    shapes = []
    for d in (512, 256):
        for M in (6400, 896):
            shapes += [('NT', [M], d, d, 1), ('NT', [M] * 3, d, d, 1), ('NT', [M], 4 * d, d, 1), ('NT', [M], d, 4 * d, 1),
                       ('NN', [M], d, d, 1), ('NN', [M], d, d, 3), ('NN', [M], 4 * d, d, 1), ('NN', [M], d, 4 * d, 1),
                       ('TN', [d], d, M, 1), ('TN', [d] * 3, d, M, 1), ('TN', [4 * d], d, M, 1), ('TN', [d], 4 * d, M, 1)]
        shapes += [('NT', [6400, 896, 896], d, d, 1)]
    if os.environ.get('GEMM_AB_ONLY'):
        shapes = [x for x in shapes if x[0] in os.environ['GEMM_AB_ONLY'].split(',')]
    st = torch.cuda.current_stream().cuda_stream
    for layout, Ms, N, K, nseg in shapes:
        lay = {'NT': 0, 'NN': 1, 'TN': 2}[layout]
        keep = []
        descs = []
        for li, T in enumerate(types):
            g = T()
            g.layout, g.ngroups, g.nseg, g.N, g.K = lay, len(Ms), nseg, N, K
            g.alpha, g.gate_scale, g.split_k = 1.0, 1.0, 1
            for i, M in enumerate(Ms):
                if layout == 'NT':
                    shp_a, shp_b, g.lda, g.ldb = (M, K), (N, K), K, K
                elif layout == 'NN':
                    shp_a, shp_b, g.lda, g.ldb = (M, K), (K, N), K, N
                else:
                    shp_a, shp_b, g.lda, g.ldb = (K, M), (K, N), M, N
                g.ldc = N
                g.g[i].M = M
                for s in range(nseg):
                    a, b = torch.randn(*shp_a, device=dev), torch.randn(*shp_b, device=dev)
                    keep += [a, b]
                    g.g[i].A[s], g.g[i].B[s] = a.data_ptr(), b.data_ptr()
                c = torch.zeros(M, N, device=dev)
                keep.append(c)
                g.g[i].C = c.data_ptr()
            if layout == 'TN':
                if is_old[li]:
                    g.split_k = old_split(Ms[0], N, len(Ms), K)
                else:
                    g.accumulate = 1
            descs.append(g)
        flops = 2.0 * sum(Ms) * N * K * nseg
        res = []
        for rep in range(2):
            for lib, g in zip(libs, descs):
                for _ in range(3):
                    assert lib.mmnas_gemm(C.byref(g), C.c_void_p(st)) == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    lib.mmnas_gemm(C.byref(g), C.c_void_p(st))
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) * 1e3 / 30)
        a, b = min(res[0], res[2]), min(res[1], res[3])
        print('%-3s M=%-18s N=%-5d K=%-5d seg=%d | A: %7.1f us %6.1f TF | B: %7.1f us %6.1f TF | B/A time %.2f'
              % (layout, Ms, N, K, nseg, a, flops / a / 1e6, b, flops / b / 1e6, b / a), flush=True)


if __name__ == '__main__':
    main()

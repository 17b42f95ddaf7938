"""Drives every HOST-ONLY entry point of the C ABI (the *_plan functions, their argument checks and error strings, the
scratch-size helpers) with realistic descriptors.  Imported by tests/test_abi.py against the product library and run as
a script by tests/test_asan.py against the AddressSanitizer build (`make -C mmnas_amd/csrc asan`; MMNAS_LIB_PATH selects
the library).  No device call is made: pointers are fake, non-null addresses."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FAKE = 0x7F0000001000   # never dereferenced by a plan function


def att_record(L, on_y, self_att, rel, d, node=0, cand=0, detached=0):
    r = L.ChainOp()
    r.kind, r.on_y = 0, on_y
    a = r.att
    a.di, a.dh, a.H = d, 64, d // 64
    a.flags = L.F_NORM | L.F_RESIDUAL | L.F_TRAIN | (L.F_SELF if self_att else 0) | ((L.F_REL | L.F_RELRAW) if rel else 0)
    a.drop_p, a.eps = 0.1, 1e-6
    for f in ('Wq', 'Wk', 'Wv', 'Wm', 'dWq', 'dWk', 'dWv', 'dWm', 'ln_a', 'ln_b', 'dln_a', 'dln_b'):
        setattr(a, f, FAKE)
    if rel:
        a.R, a.C = 64, 4
        for f in ('Wr', 'br', 'dWr', 'dbr', 'Wy', 'by', 'dWy', 'dby'):
            setattr(a, f, FAKE)
    r.node, r.cand, r.detached = node, cand, detached
    return r


def mlp_record(L, on_y, d, node=0, cand=0, detached=0):
    r = L.ChainOp()
    r.kind, r.on_y = 1, on_y
    m = r.mlp
    m.nl = 2
    m.dims[0], m.dims[1], m.dims[2] = d, 4 * d, d
    m.flags, m.drop_p, m.eps = L.F_NORM | L.F_RESIDUAL | L.F_TRAIN, 0.1, 1e-6
    for i in range(2):
        m.W[i] = m.dW[i] = m.b[i] = m.db[i] = FAKE
    m.ln_a = m.ln_b = m.dln_a = m.dln_b = FAKE
    r.node, r.cand, r.detached = node, cand, detached
    return r


def chain(L, records, B=64, Sx=14, Sy=100, d=256, mixed=False):
    arr = (L.ChainOp * len(records))(*records)
    ch = L.Chain()
    ch.n_ops, ch.ops = len(records), arr
    ch.B, ch.Sx, ch.Sy, ch.d = B, Sx, Sy, d
    ch.x_in = ch.y_in = ch.x_mask = ch.y_mask = ch.y_rel = ch.x_rel = FAKE
    if mixed:
        ch.mixed, ch.gate_width, ch.gate, ch.dgate = 1, 4, FAKE, FAKE
    return ch, arr


def run():
    from mmnas_amd import _lib as L
    lib = L.lib()
    err = lambda: lib.mmnas_last_error().decode()
    out = {}
    # operator plans + an argument error with its message
    op = L.AttOp()
    op.B, op.Sq, op.Sk, op.d, op.di, op.H, op.dh, op.R = 64, 100, 100, 512, 512, 8, 64, 64
    op.flags = L.F_NORM | L.F_RESIDUAL | L.F_MASK | L.F_REL | L.F_SELF | L.F_TRAIN
    op.drop_p = 0.1
    p = L.Plan()
    L.check(lib.mmnas_att_op_plan(C.byref(op), C.byref(p)))
    out['att_save'], out['att_ws'] = p.save_bytes, p.ws_bwd_bytes
    op.H = 7
    assert lib.mmnas_att_op_plan(C.byref(op), C.byref(p)) == -1 and 'H*dh' in err()
    m = L.MlpOp()
    m.M, m.nl = 6400, 2
    m.dims[0], m.dims[1], m.dims[2] = 512, 2048, 512
    m.flags = L.F_NORM | L.F_RESIDUAL
    L.check(lib.mmnas_mlp_op_plan(C.byref(m), C.byref(p)))
    out['mlp_save'] = p.save_bytes
    m.dims[2] = 256
    assert lib.mmnas_mlp_op_plan(C.byref(m), C.byref(p)) != 0 and 'width' in err()
    # backbone chain of a sampled architecture (weight step): 12 encoder + 18 decoder operators
    recs = []
    for i in range(12):
        recs.append(att_record(L, 0, True, False, 256, node=i) if i % 2 == 0 else mlp_record(L, 0, 256, node=i))
    for i in range(18):
        k = 12 + i
        recs.append([att_record(L, 1, True, True, 256, node=k), att_record(L, 1, False, False, 256, node=k), mlp_record(L, 1, 256, node=k)][i % 3])
    ch, keep = chain(L, recs)
    sz = C.c_size_t()
    L.check(lib.mmnas_chain_plan(C.byref(ch), C.byref(sz)))
    out['chain_bytes'] = sz.value
    assert sz.value > 30 * 6400 * 256 * 4
    # the same supernet's architecture step (mode 'full'): every candidate of every node
    recs = []
    for k in range(12):
        recs += [att_record(L, 0, True, False, 256, node=k, cand=0, detached=int(k % 2 != 0)),
                 mlp_record(L, 0, 256, node=k, cand=1, detached=int(k % 2 == 0))]
    for k in range(12, 30):
        a = k % 4
        recs += [att_record(L, 1, True, False, 256, node=k, cand=0, detached=int(a != 0)),
                 att_record(L, 1, True, True, 256, node=k, cand=1, detached=int(a != 1)),
                 att_record(L, 1, False, False, 256, node=k, cand=2, detached=int(a != 2)),
                 mlp_record(L, 1, 256, node=k, cand=3, detached=int(a != 3))]
    assert len(recs) == 96
    ch, keep = chain(L, recs, mixed=True)
    L.check(lib.mmnas_chain_plan(C.byref(ch), C.byref(sz)))
    out['mixed_chain_bytes'] = sz.value
    assert sz.value > out['chain_bytes']
    # malformed mixed chains are refused with a message, not walked
    recs[1].detached = 0          # node 0 with two differentiated candidates
    ch, keep = chain(L, recs, mixed=True)
    assert lib.mmnas_chain_plan(C.byref(ch), C.byref(sz)) != 0 and 'differentiated' in err()
    recs[1].detached = 1
    recs[5].cand = 9              # candidate index outside the gate row
    ch, keep = chain(L, recs, mixed=True)
    assert lib.mmnas_chain_plan(C.byref(ch), C.byref(sz)) != 0 and 'gate row' in err()
    recs[5].cand = 1
    ch, keep = chain(L, recs[:3] + [recs[0]] + recs[3:], mixed=True)    # a node id going backwards
    assert lib.mmnas_chain_plan(C.byref(ch), C.byref(sz)) != 0
    ch, keep = chain(L, [att_record(L, 0, False, False, 256)])            # guided attention on the language stream
    assert lib.mmnas_chain_plan(C.byref(ch), C.byref(sz)) != 0 and 'guided' in err()
    ch, keep = chain(L, recs * 2)                                            # more operators than the chain holds
    assert lib.mmnas_chain_plan(C.byref(ch), C.byref(sz)) != 0
    # answer head
    hd = L.Head()
    hd.B, hd.d, hd.MID, hd.G, hd.OUT, hd.ANS = 64, 256, 512, 1, 512, 3129
    hd.sx.S, hd.sy.S = 14, 100
    L.check(lib.mmnas_head_plan(C.byref(hd), C.byref(sz)))
    out['head_bytes'] = sz.value
    hd.MID = 511
    assert lib.mmnas_head_plan(C.byref(hd), C.byref(sz)) != 0
    # scratch-size helpers and capability queries
    out['ln_ws'] = lib.mmnas_layernorm_bwd_ws_floats(6400, 512)
    out['mix_ws'] = lib.mmnas_mixed_sum_ws_floats()
    out['emb_ws'] = lib.mmnas_embedding_bwd_det_ws_floats(896, 300)
    assert out['emb_ws'] == 896 * 300 and out['mix_ws'] > 0 and out['ln_ws'] > 0
    assert lib.mmnas_abi_version() >= 1
    # round 6: the row-panel product's switch and its argument checks (host-only: nothing is launched)
    prev = lib.mmnas_set_gemm_ln(1)
    assert prev in (0, 1) and lib.mmnas_set_gemm_ln(prev) == 1
    gd = L.GemmDesc()
    gd.layout, gd.ngroups, gd.nseg, gd.N, gd.K, gd.lda, gd.ldb, gd.ldc = L.GEMM_NT, 1, 1, 256, 256, 256, 256, 256
    assert lib.mmnas_gemm_ln(C.byref(gd), None, None, None, 1e-6, None) != 0 and 'null pointer' in err()
    gd.ngroups = 2
    one = C.c_void_p(16)
    assert lib.mmnas_gemm_ln(C.byref(gd), one, one, one, 1e-6, None) != 0 and 'one group' in err()
    return out


if __name__ == '__main__':
    print('HOST_PLAN_OK', run())

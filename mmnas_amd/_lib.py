"""ctypes binding of libmmnas_hip.so (include/mmnas_hip.h).

There is deliberately no fallback: if the shared library is missing, or a tensor is not a
contiguous fp32 HIP tensor, the call raises.  The product path never computes on the CPU.
"""
import ctypes as C
import logging
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MMNAS_LIB_PATH') or os.path.join(_HERE, 'lib', 'libmmnas_hip.so')   # (override: tuning builds)

log = logging.getLogger('mmnas_amd')


def _launch_configuration():
    """HIP_FORCE_DEV_KERNARG=1 (kernel arguments written straight into device memory) is worth 20-40 % of a supernet step
    on this pool (README: 6.1-7.3 ms instead of 5.1 with 0): a step is ~300 short dependent launches and each one's first
    instruction waits for its argument block.  The HIP runtime reads the variable when it initialises, i.e. at the process's
    first HIP call, so the LIBRARY sets it (when the user has not) as soon as it is imported -- `import mmnas.model...` in
    the reference's scripts comes before the first tensor reaches the GPU -- instead of leaving it to bench.py.  Imported
    after HIP is up with the variable unset, the library can only say so: one warning on the `mmnas_amd` logger.
    Returns (value, source); `ops.runtime_config()` reports both."""
    name = 'HIP_FORCE_DEV_KERNARG'
    if name in os.environ:
        return os.environ[name], 'inherited'
    if torch.cuda.is_initialized():
        log.warning('mmnas_amd was imported after HIP had initialised and %s is not set: kernel arguments go through host-visible '
                    'memory, the ~300 short launches of a step run 20-40 %% slower.  Export %s=1 or import mmnas / mmnas_amd '
                    'before the first CUDA call.', name, name)
        return None, 'unset_after_hip_init'
    os.environ[name] = '1'
    return '1', 'set_by_library'


KERNARG_VALUE, KERNARG_SOURCE = _launch_configuration()

F_NORM, F_RESIDUAL, F_MASK, F_REL, F_SELF, F_TRAIN, F_RELRAW = 1, 2, 4, 8, 16, 32, 64
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2

_fp = C.c_void_p  # device pointers travel as void*


class GemmGroup(C.Structure):
    _fields_ = [('M', C.c_int), ('A', _fp * 3), ('B', _fp * 3), ('C', _fp), ('bias', _fp),
                ('residual', _fp), ('gate', _fp), ('colsum', _fp), ('drop_seed', C.c_uint64)]


GEMM_MAX_GROUPS = 9


class GemmDesc(C.Structure):
    _fields_ = [('layout', C.c_int), ('ngroups', C.c_int), ('nseg', C.c_int), ('N', C.c_int), ('K', C.c_int),
                ('lda', C.c_int), ('ldb', C.c_int), ('ldc', C.c_int), ('ldres', C.c_int), ('ldgate', C.c_int),
                ('relu', C.c_int), ('split_k', C.c_int), ('accumulate', C.c_int), ('b_planes', C.c_int), ('alpha', C.c_float), ('gate_scale', C.c_float),
                ('drop_p', C.c_float), ('drop_site', C.c_uint32), ('drop_seed', C.c_uint64),
                ('g', GemmGroup * GEMM_MAX_GROUPS)]


class MhaDesc(C.Structure):
    _fields_ = [('B', C.c_int), ('H', C.c_int), ('Sq', C.c_int), ('Sk', C.c_int), ('dh', C.c_int),
                ('ldq', C.c_int), ('ldk', C.c_int), ('ldv', C.c_int), ('ldo', C.c_int),
                ('Q', _fp), ('K', _fp), ('V', _fp), ('mask', _fp), ('biasT', _fp), ('O', _fp), ('lse', _fp),
                ('drop_p', C.c_float), ('drop_site', C.c_uint32), ('drop_seed', C.c_uint64),
                ('dO', _fp), ('dQ', _fp), ('dK', _fp), ('dV', _fp), ('dbiasT', _fp), ('delta', _fp), ('q_off', _fp), ('k_off', _fp)]


class AttOp(C.Structure):
    _fields_ = [('B', C.c_int), ('Sq', C.c_int), ('Sk', C.c_int), ('d', C.c_int), ('di', C.c_int),
                ('H', C.c_int), ('dh', C.c_int), ('R', C.c_int), ('flags', C.c_int),
                ('drop_p', C.c_float), ('eps', C.c_float), ('seed', C.c_uint64),
                ('xq', _fp), ('xkv', _fp), ('mask', _fp), ('rel', _fp),
                ('Wq', _fp), ('Wk', _fp), ('Wv', _fp), ('Wm', _fp), ('Wr', _fp), ('br', _fp),
                ('ln_a', _fp), ('ln_b', _fp), ('y', _fp), ('save', _fp), ('ws', _fp),
                ('dy', _fp), ('dxq', _fp), ('dxkv', _fp), ('drel', _fp),
                ('dWq', _fp), ('dWk', _fp), ('dWv', _fp), ('dWm', _fp), ('dWr', _fp), ('dbr', _fp),
                ('dln_a', _fp), ('dln_b', _fp),
                ('C', C.c_int), ('Wy', _fp), ('by', _fp), ('dWy', _fp), ('dby', _fp),
                ('q_off', _fp), ('k_off', _fp), ('Mq', C.c_int), ('Mk', C.c_int), ('rel_tile_off', _fp), ('rel_ntiles', C.c_int), ('reserved2', C.c_int)]


class Plan(C.Structure):
    _fields_ = [('save_bytes', C.c_size_t), ('ws_fwd_bytes', C.c_size_t), ('ws_bwd_bytes', C.c_size_t)]


class MlpOp(C.Structure):
    _fields_ = [('M', C.c_int), ('nl', C.c_int), ('dims', C.c_int * 4), ('flags', C.c_int),
                ('drop_p', C.c_float), ('eps', C.c_float), ('seed', C.c_uint64),
                ('x', _fp), ('W', _fp * 3), ('b', _fp * 3), ('ln_a', _fp), ('ln_b', _fp), ('y', _fp),
                ('save', _fp), ('ws', _fp), ('dy', _fp), ('dx', _fp), ('dW', _fp * 3), ('db', _fp * 3),
                ('dln_a', _fp), ('dln_b', _fp)]


class ChainOp(C.Structure):
    _fields_ = [('kind', C.c_int), ('on_y', C.c_int), ('att', AttOp), ('mlp', MlpOp),
                ('node', C.c_int), ('cand', C.c_int), ('detached', C.c_int), ('reserved', C.c_int)]


class Chain(C.Structure):
    _fields_ = [('n_ops', C.c_int), ('ops', C.POINTER(ChainOp)), ('B', C.c_int), ('Sx', C.c_int), ('Sy', C.c_int),
                ('d', C.c_int), ('x_in', _fp), ('y_in', _fp), ('x_mask', _fp), ('y_mask', _fp), ('x_rel', _fp),
                ('y_rel', _fp), ('arena', _fp), ('x_out', _fp), ('y_out', _fp), ('dx_out', _fp), ('dy_out', _fp),
                ('dx_in', _fp), ('dy_in', _fp), ('use_side_stream', C.c_int), ('reserved', C.c_int),
                ('marks', C.c_void_p), ('mixed', C.c_int), ('gate_width', C.c_int), ('gate', _fp), ('dgate', _fp),
                ('y_off', _fp), ('y_tile_off', _fp), ('Ny', C.c_int), ('y_ntiles', C.c_int)]


CHAIN_MAX_OPS = 128


class AttFlatSide(C.Structure):
    _fields_ = [('S', C.c_int), ('reserved', C.c_int), ('x', _fp), ('mask', _fp), ('W1', _fp), ('b1', _fp), ('W2', _fp),
                ('b2', _fp), ('Wm', _fp), ('bm', _fp), ('dW1', _fp), ('db1', _fp), ('dW2', _fp), ('db2', _fp), ('dWm', _fp),
                ('dbm', _fp), ('seed', C.c_uint64), ('dx', _fp), ('off', _fp), ('M', C.c_int), ('reserved2', C.c_int)]


class Head(C.Structure):
    _fields_ = [('B', C.c_int), ('d', C.c_int), ('MID', C.c_int), ('G', C.c_int), ('OUT', C.c_int), ('ANS', C.c_int),
                ('flags', C.c_int), ('reserved', C.c_int), ('drop_p', C.c_float), ('eps', C.c_float),
                ('sx', AttFlatSide), ('sy', AttFlatSide), ('ln_a', _fp), ('ln_b', _fp), ('dln_a', _fp), ('dln_b', _fp),
                ('Wp', _fp), ('bp', _fp), ('dWp', _fp), ('dbp', _fp), ('logits', _fp), ('dlogits', _fp), ('arena', _fp)]


REL_MULTI_MAX = 32


class RelMulti(C.Structure):
    _fields_ = [('B', C.c_int), ('S', C.c_int), ('C', C.c_int), ('R', C.c_int), ('H', C.c_int), ('n_ops', C.c_int),
                ('raw', _fp), ('Wy', _fp), ('by', _fp), ('dWy', _fp), ('dby', _fp),
                ('Wr', _fp * REL_MULTI_MAX), ('br', _fp * REL_MULTI_MAX), ('biasT', _fp * REL_MULTI_MAX),
                ('dbiasT', _fp * REL_MULTI_MAX), ('dWr', _fp * REL_MULTI_MAX), ('dbr', _fp * REL_MULTI_MAX),
                ('off', _fp), ('tile_off', _fp), ('ntiles', C.c_int), ('reserved', C.c_int), ('ws', _fp)]


class ProfStat(C.Structure):
    _fields_ = [('ms', C.c_double), ('flops', C.c_double), ('bytes', C.c_double), ('launches', C.c_long)]


K_NAMES = ['gemm', 'mha_fwd', 'mha_bwd', 'rel_fwd', 'rel_bwd', 'rowops', 'lstm', 'small_ops']


class Segment(C.Structure):
    _fields_ = [('ptr', _fp), ('offset', C.c_uint64), ('n', C.c_uint64)]


# every symbol include/mmnas_hip.h declares: name -> (restype, argtypes)
_i, _f, _u32, _u64, _sz = C.c_int, C.c_float, C.c_uint32, C.c_uint64, C.c_size_t
SYMBOLS = {
    'mmnas_abi_version': (_i, []),
    'mmnas_last_error': (C.c_char_p, []),
    'mmnas_dropout_mask': (_i, [_fp, _sz, _f, _u64, _u32, _fp]),
    'mmnas_gemm': (_i, [C.POINTER(GemmDesc), _fp]),
    'mmnas_split_planes': (_i, [_fp, _fp, _sz, _fp]),
    'mmnas_gemm_ln': (_i, [C.POINTER(GemmDesc), _fp, _fp, _fp, _f, _fp]),
    'mmnas_set_gemm_ln': (_i, [_i]),
    'mmnas_gemm_pair': (_i, [C.POINTER(GemmDesc), C.POINTER(GemmDesc), _fp]),
    'mmnas_gemm_reload_tuning': (_i, []),
    'mmnas_lstm_supported': (_i, [_i, _i]),
    'mmnas_lstm_fwd': (_i, [_fp] * 9 + [_i, _i, _i, _i, _fp]),
    'mmnas_lstm_bwd': (_i, [_fp] * 7 + [_i, _i, _i, _fp]),
    'mmnas_lstm_seq_supported': (_i, [_i, _i]),
    'mmnas_set_small_ops': (_i, [_i]),
    'mmnas_set_small_bwd': (_i, [_i]),
    'mmnas_set_small_ffn': (_i, [_i]),
    'mmnas_set_chain_overlap': (_i, [_i]),
    'mmnas_lstm_seq_fwd': (_i, [_fp] * 7 + [_i, _i, _i, _fp]),
    'mmnas_lstm_seq_bwd': (_i, [_fp] * 5 + [_i, _i, _i, _fp]),
    'mmnas_lstm_seq_timed_out': (_i, [_fp]),
    'mmnas_layernorm_fwd': (_i, [_fp, _fp, _fp, _fp, _i, _i, _f, _fp]),
    'mmnas_layernorm_bwd': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _f, _u64, _u32, _i, _i, _f, _fp]),
    'mmnas_layernorm_bwd_ws_floats': (_sz, [_i, _i]),
    'mmnas_colsum': (_i, [_fp, _fp, _i, _i, _i, _fp]),
    'mmnas_eltwise_fwd': (_i, [_i, _fp, _fp, _sz, _fp]),
    'mmnas_eltwise_bwd': (_i, [_i, _fp, _fp, _fp, _sz, _fp]),
    'mmnas_drop_add': (_i, [_fp, _fp, _fp, _sz, _f, _u64, _u32, _fp]),
    'mmnas_glu_fwd': (_i, [_fp, _fp, _i, _i, _i, _f, _u64, _u32, _fp]),
    'mmnas_glu_bwd': (_i, [_fp, _fp, _fp, _i, _i, _i, _f, _u64, _u32, _fp]),
    'mmnas_rel_bias_fwd': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp]),
    'mmnas_rel_bias_bwd': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp]),
    'mmnas_row_is_zero': (_i, [_fp, _fp, C.c_long, _i, _fp]),
    'mmnas_glimpse1_supported': (_i, [_i]),
    'mmnas_glimpse1_bwd_ws_floats': (_sz, [C.c_long, _i]),
    'mmnas_glimpse1_fwd': (_i, [_fp, _fp, _fp, _fp, C.c_long, _i, _fp]),
    'mmnas_glimpse1_bwd': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_long, _i, _fp]),
    'mmnas_attflat_pool_fwd': (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp]),
    'mmnas_attflat_pool_bwd': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp]),
    'mmnas_relation_embedding': (_i, [_fp, _fp, _fp, _i, _i, _fp]),
    'mmnas_semantic_embedding': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, C.c_long, _fp]),
    'mmnas_onehot_rows': (_i, [_fp, _i, _i, C.POINTER(C.c_int), _fp]),
    'mmnas_mixed_sum_ws_floats': (_sz, []),
    'mmnas_mixed_sum_fwd': (_i, [C.POINTER(_fp), _i, _fp, _fp, _sz, _fp]),
    'mmnas_mixed_sum_bwd': (_i, [C.POINTER(_fp), _i, _fp, _fp, _fp, _i, _fp, _fp, _sz, _fp]),
    'mmnas_alpha_full_step': (_i, [_fp, _fp, _fp, _fp, _fp, _i, _i, _f, _f, _f, _f, _i, _fp]),
    'mmnas_embedding_bwd': (_i, [_fp, _fp, _fp, C.c_long, _i, C.c_long, _fp]),
    'mmnas_node_mix_fwd': (_i, [_fp, _fp, _fp, _i, _fp, _fp, _i, _i, _f, _fp]),
    'mmnas_node_mix_bwd': (_i, [_fp, _fp, _fp, _i, _fp, _fp, _fp, _i, _fp, _fp, _i, _i, _f, _fp]),
    'mmnas_embedding_bwd_det': (_i, [_fp, _fp, _fp, _fp, C.c_long, _i, C.c_long, _f, _fp]),
    'mmnas_embedding_bwd_det_ws_floats': (_sz, [C.c_long, _i]),
    'mmnas_rel_fused_supported': (_i, [_i, _i, _i]),
    'mmnas_rel_fused_fwd': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp]),
    'mmnas_rel_fused_bwd_ws_floats': (_sz, [_i, _i, _i]),
    'mmnas_rel_fused_bwd': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp]),
    'mmnas_rel_fused_fwd_ragged': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp]),
    'mmnas_rel_fused_bwd_ragged': (_i, [_fp] * 11 + [_i, _i, _i, _i, _i, _fp, _fp, _i, _fp]),
    'mmnas_rel_multi_supported': (_i, [_i, _i, _i]),
    'mmnas_rel_multi_bwd_ws_floats': (_sz, [_i, _i]),
    'mmnas_rel_multi_fwd': (_i, [C.POINTER(RelMulti), _fp]),
    'mmnas_rel_multi_bwd': (_i, [C.POINTER(RelMulti), _fp]),
    'mmnas_set_rel_hoist': (_i, [_i]),
    'mmnas_set_guided_hoist': (_i, [_i]),
    'mmnas_set_rel_overlap': (_i, [_i]),
    'mmnas_mha_core_fwd': (_i, [C.POINTER(MhaDesc), _fp]),
    'mmnas_mha_core_bwd': (_i, [C.POINTER(MhaDesc), _fp]),
    'mmnas_att_op_plan': (_i, [C.POINTER(AttOp), C.POINTER(Plan)]),
    'mmnas_att_op_fwd': (_i, [C.POINTER(AttOp), _fp]),
    'mmnas_att_op_bwd': (_i, [C.POINTER(AttOp), _fp]),
    'mmnas_mlp_op_plan': (_i, [C.POINTER(MlpOp), C.POINTER(Plan)]),
    'mmnas_mlp_op_fwd': (_i, [C.POINTER(MlpOp), _fp]),
    'mmnas_mlp_op_bwd': (_i, [C.POINTER(MlpOp), _fp]),
    'mmnas_chain_plan': (_i, [C.POINTER(Chain), C.POINTER(C.c_size_t)]),
    'mmnas_chain_fwd': (_i, [C.POINTER(Chain), _fp]),
    'mmnas_chain_bwd': (_i, [C.POINTER(Chain), _fp]),
    'mmnas_chain_join': (_i, [_fp, _fp]),
    'mmnas_head_plan': (_i, [C.POINTER(Head), C.POINTER(C.c_size_t)]),
    'mmnas_head_fwd': (_i, [C.POINTER(Head), _fp]),
    'mmnas_head_bwd': (_i, [C.POINTER(Head), _fp]),
    'mmnas_bce_logits_sum_fwd': (_i, [_fp, _fp, _fp, _sz, _fp]),
    'mmnas_bce_logits_bwd': (_i, [_fp, _fp, _fp, _fp, _sz, _fp]),
    'mmnas_im2col_seq': (_i, [_fp, _fp, _i, _i, _i, _i, _fp]),
    'mmnas_pack_rows': (_i, [_fp, _fp, _fp, _i, _i, _i, _fp]),
    'mmnas_unpack_rows': (_i, [_fp, _fp, _fp, _i, _i, _i, _fp]),
    'mmnas_pad_seq': (_i, [_fp, _fp, _i, _i, _i, _i, _i, C.c_long, _fp]),
    'mmnas_col2im_seq': (_i, [_fp, _fp, _i, _i, _i, _i, _fp]),
    'mmnas_dwconv_seq_fwd': (_i, [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp]),
    'mmnas_dwconv_seq_bwd': (_i, [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp]),
    'mmnas_pack_segments': (_i, [_fp, _i, _fp, _f, _i, _fp]),
    'mmnas_pack_segments_host': (_i, [C.POINTER(Segment), _i, _fp, _f, _i, _fp]),
    'mmnas_adam_step': (_i, [_fp, _fp, _fp, _fp, _sz, _f, _f, _f, _f, _f, _fp, _f, _i, _fp]),
    'mmnas_sumsq': (_i, [_fp, _sz, _fp, _fp]),
    'mmnas_prof_enable': (_i, [_i]),
    'mmnas_prof_collect': (_i, [C.POINTER(ProfStat)]),
}

_lib = None


class MMNasHipError(RuntimeError):
    pass


def lib():
    """The loaded shared library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMNasHipError(
                'libmmnas_hip.so not found at %s -- build it with `python -c "import __graft_entry__ as g; '
                'g.build()"` or `make -C mmnas_amd/csrc`.  There is no CPU fallback.' % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        if l.mmnas_abi_version() != 1:
            raise MMNasHipError('libmmnas_hip.so ABI version %d != 1' % l.mmnas_abi_version())
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise MMNasHipError('libmmnas_hip: %s (code %d)' % (lib().mmnas_last_error().decode(), rc))


def ptr(t):
    """Device pointer of a contiguous tensor, or None (NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MMNasHipError('mmnas_amd operators run on the MI355X only: got a %s tensor (no CPU fallback)'
                            % t.device)
    if not t.is_contiguous():
        raise MMNasHipError('non-contiguous tensor passed to the HIP boundary')
    return t.data_ptr()


def fptr(t):
    if t is not None and t.dtype != torch.float32:
        raise MMNasHipError('expected float32, got %s' % t.dtype)
    return ptr(t)


def stream():
    """Raw handle of torch's current stream on the current device (the C calls take a hipStream_t).  The private
    accessors cost ~0.3 us; torch.cuda.current_stream() builds a Stream object (~10 us, ~45 times a step)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())

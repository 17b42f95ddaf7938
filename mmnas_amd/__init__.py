"""mmnas_amd: MI355X-native implementation of the MMNas candidate-operator hot path.

    mmnas_amd.csrc/           hand-written HIP kernels + the C ABI (include/mmnas_hip.h)
    mmnas_amd._lib / .ops     ctypes binding and the autograd bridge
    mmnas_amd.model / .utils  host-side mirror of mmnas.model / mmnas.utils (same names and keys)
    mmnas_amd.dp              data-parallel gradient exchange over RCCL

`import mmnas` (the alias package at the repo root) resolves to these modules, so the reference's
search_vqa.py / train_vqa.py import lines work unchanged.
"""
__version__ = '0.1.0'

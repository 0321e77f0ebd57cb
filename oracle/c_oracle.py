"""ctypes binding of oracle/libts2d_ref.so (the C restatement, checker only)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _Arch(ctypes.Structure):
    _fields_ = [('input_channels', ctypes.c_int), ('num_classes', ctypes.c_int), ('n_stages', ctypes.c_int),
                ('features', ctypes.c_int * 16), ('n_conv_enc', ctypes.c_int * 16), ('n_conv_dec', ctypes.c_int * 16),
                ('eps', ctypes.c_float), ('slope', ctypes.c_float), ('strides', (ctypes.c_int * 2) * 16)]


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libts2d_ref.so')
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.ts2d_ref_forward.restype = ctypes.c_int
        _LIB.ts2d_ref_forward.argtypes = [ctypes.POINTER(_Arch), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        _LIB.ts2d_ref_mask.restype = None
        _LIB.ts2d_ref_mask.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    return _LIB


def _c_arch(arch) -> _Arch:
    arch.validate()
    a = _Arch()
    a.input_channels, a.num_classes, a.n_stages = arch.input_channels, arch.num_classes, arch.n_stages
    for i, f in enumerate(arch.features_per_stage):
        a.features[i] = f
    for i, c in enumerate(arch.n_conv_per_stage):
        a.n_conv_enc[i] = c
    for i, c in enumerate(arch.n_conv_per_stage_decoder):
        a.n_conv_dec[i] = c
    a.eps, a.slope = arch.norm_eps, arch.leaky_slope
    for i, st in enumerate(arch.strides):
        a.strides[i][0], a.strides[i][1] = int(st[0]), int(st[1])
    return a


def unet_forward(arch, blob: np.ndarray, x: np.ndarray, acc64: bool = False) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    blob = np.ascontiguousarray(blob, dtype=np.float32)
    B, C, H, W = x.shape
    assert C == arch.input_channels and blob.size == arch.n_params()
    out = np.empty((B, arch.num_classes, H, W), dtype=np.float32)
    a = _c_arch(arch)
    rc = lib().ts2d_ref_forward(ctypes.byref(a), blob.ctypes.data, x.ctypes.data, B, H, W, out.ctypes.data, int(acc64))
    if rc != 0:
        raise RuntimeError(f"ts2d_ref_forward failed with code {rc}")
    return out


def logits_to_mask(logits: np.ndarray) -> np.ndarray:
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    m = np.empty(logits.shape, dtype=np.uint8)
    lib().ts2d_ref_mask(logits.ctypes.data, logits.size, m.ctypes.data)
    return m

"""Loader for ``libts2d_engine.so`` (the C-ABI of include/ts2d_engine.h).

north_star asks for "a thin C-ABI cffi layer"; ``cffi`` is used in ABI mode when it is importable and ``ctypes``
otherwise (cffi is absent from this image).  There is deliberately NO CPU fallback: if the HIP library is missing
or cannot be loaded the import of the product path fails loudly.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libts2d_engine.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'ts2d_engine.h')
ABI_VERSION = 7
MAX_STAGES = 16
PRECISION_F32_EXACT = 0
PRECISION_F32_SPLIT_F16X3 = 1
PRECISION_F16 = 2

# every symbol include/ts2d_engine.h declares
SYMBOLS = ('ts2d_engine_create', 'ts2d_engine_load_weights', 'ts2d_engine_weight_buffer', 'ts2d_engine_weights_ready',
           'ts2d_engine_forward', 'ts2d_engine_check', 'ts2d_engine_predict_tiled', 'ts2d_engine_tiled_inf_flag', 'ts2d_engine_set_tile_dtype', 'ts2d_engine_set_keep_activations', 'ts2d_project_coronal', 'ts2d_project_coronal_zscore', 'ts2d_synth_slices', 'ts2d_engine_reserve', 'ts2d_engine_workspace_bytes', 'ts2d_engine_set_workspace', 'ts2d_engine_set_precision', 'ts2d_engine_set_option', 'ts2d_engine_set_profiling', 'ts2d_engine_num_ops',
           'ts2d_engine_op_name', 'ts2d_engine_op_kernel', 'ts2d_engine_op_times', 'ts2d_engine_debug_tensor', 'ts2d_engine_device_bytes', 'ts2d_engine_destroy',
           'ts2d_last_error', 'ts2d_abi_version')


class ArchDesc(ctypes.Structure):
    """``ts2d_arch_desc``."""
    _fields_ = [('input_channels', ctypes.c_int32), ('num_classes', ctypes.c_int32), ('n_stages', ctypes.c_int32),
                ('features', ctypes.c_int32 * MAX_STAGES), ('n_conv_enc', ctypes.c_int32 * MAX_STAGES),
                ('n_conv_dec', ctypes.c_int32 * MAX_STAGES), ('norm_eps', ctypes.c_float), ('leaky_slope', ctypes.c_float),
                ('strides', (ctypes.c_int32 * 2) * MAX_STAGES)]


class EngineLibraryError(RuntimeError):
    pass


_lib = None


def build(verbose: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess
    subprocess.check_call(['make', '-C', os.path.join(_HERE, 'csrc')] + ([] if verbose else ['-s']))
    return LIB_PATH


def load():
    """dlopen the engine and declare its signatures.  Raises EngineLibraryError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineLibraryError(
            f"{LIB_PATH} is missing: the MI355X HIP engine has not been built (run `python -c 'import "
            f"__graft_entry__ as g; g.build()'` or `make -C totalsegmentator2d_amd/csrc`). There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as ex:
        raise EngineLibraryError(f"failed to load {LIB_PATH}: {ex}") from ex
    c = ctypes
    lib.ts2d_abi_version.restype = c.c_int
    lib.ts2d_abi_version.argtypes = []
    if lib.ts2d_abi_version() != ABI_VERSION:
        raise EngineLibraryError(f"ABI mismatch: library {lib.ts2d_abi_version()}, binding {ABI_VERSION}")
    lib.ts2d_last_error.restype = c.c_char_p
    lib.ts2d_last_error.argtypes = []
    lib.ts2d_engine_create.restype = c.c_int
    lib.ts2d_engine_create.argtypes = [c.POINTER(ArchDesc), c.c_void_p, c.c_size_t, c.c_int, c.POINTER(c.c_void_p)]
    lib.ts2d_engine_load_weights.restype = c.c_int
    lib.ts2d_engine_load_weights.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t]
    lib.ts2d_engine_weight_buffer.restype = c.c_int
    lib.ts2d_engine_weight_buffer.argtypes = [c.c_void_p, c.POINTER(c.c_void_p), c.POINTER(c.c_size_t)]
    lib.ts2d_engine_weights_ready.restype = c.c_int
    lib.ts2d_engine_weights_ready.argtypes = [c.c_void_p]
    lib.ts2d_engine_forward.restype = c.c_int
    lib.ts2d_engine_forward.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_void_p,
                                        c.c_int, c.c_void_p]
    lib.ts2d_engine_check.restype = c.c_int
    lib.ts2d_engine_check.argtypes = [c.c_void_p]
    lib.ts2d_engine_predict_tiled.restype = c.c_int
    lib.ts2d_engine_predict_tiled.argtypes = [c.c_void_p, c.c_void_p, c.c_int, c.c_int, c.c_int, c.c_int, c.c_int, c.c_void_p, c.c_void_p,
                                              c.c_int, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.ts2d_project_coronal.restype = c.c_int
    lib.ts2d_project_coronal.argtypes = [c.c_int, c.c_void_p, c.c_size_t, c.c_int, c.c_int, c.c_int, c.c_int, c.c_longlong, c.c_longlong,
                                         c.c_longlong, c.c_longlong, c.c_void_p, c.c_void_p]
    lib.ts2d_project_coronal_zscore.restype = c.c_int
    lib.ts2d_project_coronal_zscore.argtypes = [c.c_int, c.c_void_p, c.c_size_t, c.c_int, c.c_int, c.c_int, c.c_int, c.c_longlong, c.c_longlong,
                                         c.c_longlong, c.c_longlong, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.ts2d_synth_slices.restype = c.c_int
    lib.ts2d_synth_slices.argtypes = [c.c_int, c.c_ulonglong, c.c_ulonglong, c.c_ulonglong, c.c_void_p, c.c_void_p]
    lib.ts2d_engine_reserve.restype = c.c_int
    lib.ts2d_engine_reserve.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int]
    lib.ts2d_engine_workspace_bytes.restype = c.c_int
    lib.ts2d_engine_workspace_bytes.argtypes = [c.c_void_p, c.c_int, c.c_int, c.c_int, c.POINTER(c.c_size_t)]
    lib.ts2d_engine_set_workspace.restype = c.c_int
    lib.ts2d_engine_set_workspace.argtypes = [c.c_void_p, c.c_void_p, c.c_size_t]
    lib.ts2d_engine_set_precision.restype = c.c_int
    lib.ts2d_engine_set_precision.argtypes = [c.c_void_p, c.c_int]
    lib.ts2d_engine_set_option.restype = c.c_int
    lib.ts2d_engine_set_option.argtypes = [c.c_void_p, c.c_char_p, c.c_int]
    lib.ts2d_engine_set_tile_dtype.restype = c.c_int
    lib.ts2d_engine_set_tile_dtype.argtypes = [c.c_void_p, c.c_int]
    lib.ts2d_engine_set_keep_activations.restype = c.c_int
    lib.ts2d_engine_set_keep_activations.argtypes = [c.c_void_p, c.c_int]
    lib.ts2d_engine_set_profiling.restype = c.c_int
    lib.ts2d_engine_set_profiling.argtypes = [c.c_void_p, c.c_int]
    lib.ts2d_engine_num_ops.restype = c.c_int
    lib.ts2d_engine_num_ops.argtypes = [c.c_void_p]
    lib.ts2d_engine_op_name.restype = c.c_char_p
    lib.ts2d_engine_op_name.argtypes = [c.c_void_p, c.c_int]
    lib.ts2d_engine_op_kernel.restype = c.c_char_p
    lib.ts2d_engine_op_kernel.argtypes = [c.c_void_p, c.c_int]
    lib.ts2d_engine_op_times.restype = c.c_int
    lib.ts2d_engine_op_times.argtypes = [c.c_void_p, c.c_void_p, c.c_int]
    lib.ts2d_engine_debug_tensor.restype = c.c_int
    lib.ts2d_engine_debug_tensor.argtypes = [c.c_void_p, c.c_char_p, c.c_void_p, c.c_size_t, c.POINTER(c.c_int32 * 4)]
    lib.ts2d_engine_tiled_inf_flag.restype = c.c_int
    lib.ts2d_engine_tiled_inf_flag.argtypes = [c.c_void_p]
    lib.ts2d_engine_device_bytes.restype = c.c_size_t
    lib.ts2d_engine_device_bytes.argtypes = [c.c_void_p]
    lib.ts2d_engine_destroy.restype = c.c_int
    lib.ts2d_engine_destroy.argtypes = [c.c_void_p]
    _lib = lib
    return lib


def load_cffi():
    """Same library through cffi (ABI mode) when cffi is installed; returns (ffi, lib) or None."""
    try:
        import cffi
    except ImportError:
        return None
    import re
    ffi = cffi.FFI()
    src = open(HEADER_PATH).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    src = '\n'.join(l for l in src.splitlines() if not l.strip().startswith('#') and 'extern "C"' not in l
                    and l.strip() not in ('}',))
    src = src.replace('TS2D_MAX_STAGES', str(MAX_STAGES))
    ffi.cdef(src)
    return ffi, ffi.dlopen(LIB_PATH)


def last_error() -> str:
    return load().ts2d_last_error().decode('utf-8', 'replace')


def check(rc: int, what: str):
    """status code -> RuntimeError (reference error convention: Python exceptions,
    ``ts2d/core/inference/prediction_worker.py:211-212``)."""
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {last_error()}")

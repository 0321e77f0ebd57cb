"""``Engine`` - Python handle on one HIP engine (one sub-model / fold on one GPU).

Thin wrapper over the C-ABI (include/ts2d_engine.h).  It replaces ``nnUNetPredictor.network`` (built at reference
``ts2d/core/inference/nnu.py:164-165``, called inside ``predict_logits_from_preprocessed_data``, reference
``ts2d/core/inference/prediction_worker.py:209``).  torch is optional: numpy host arrays always work, torch CUDA
tensors are consumed zero-copy through their ``data_ptr()``.
"""
from __future__ import annotations

import ctypes
import weakref
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib
from .arch import UNetArch


def _desc(arch: UNetArch) -> _lib.ArchDesc:
    arch.validate()
    d = _lib.ArchDesc()
    d.input_channels, d.num_classes, d.n_stages = arch.input_channels, arch.num_classes, arch.n_stages
    for i, f in enumerate(arch.features_per_stage):
        d.features[i] = int(f)
    for i, c in enumerate(arch.n_conv_per_stage):
        d.n_conv_enc[i] = int(c)
    for i, c in enumerate(arch.n_conv_per_stage_decoder):
        d.n_conv_dec[i] = int(c)
    d.norm_eps, d.leaky_slope = arch.norm_eps, arch.leaky_slope
    for i, st in enumerate(arch.strides):
        d.strides[i][0], d.strides[i][1] = int(st[0]), int(st[1])
    return d


def _is_torch(x) -> bool:
    return type(x).__module__.startswith('torch') and hasattr(x, 'data_ptr')


class Engine:
    default_options: Dict[str, int] = {}     # dispatch options every new handle starts with (empty in the product; test modules that address
                                             # the large-batch kernels at small B set {'sbk': 0})

    def __init__(self, arch: UNetArch, blob: Optional[np.ndarray], device: int = 0, options: Optional[Dict[str, int]] = None):
        """blob: fp32 weight blob (:func:`weights.pack_blob`) or None for a replica to be filled by broadcast.
        options: kernel-dispatch options (:meth:`set_option`), e.g. ``{'upc': 0}`` - tests and A/B scripts."""
        self.lib = _lib.load()
        self.arch = arch
        self.device = int(device)
        self._keep = False           # one buffer per activation: switched on by the USER (keep_activations)
        self._auto_keep = False      # ... or by debug_tensor, until the next forward (which returns to the shared arena)
        self._last = None            # (weak reference to the input, logits?, mask?) of the last forward: debug_tensor re-runs it with
                                     # private buffers.  Weak: a production call must not pin the caller's batch in HBM.
        self._ran = False            # a forward has run on this handle (distinguishes "no forward yet" from "its input is gone")
        self._ws_need = {}           # (B, H, W) -> workspace_bytes under the CURRENT mode / options / keep flag (cleared when those change)
        self._h = ctypes.c_void_p()
        d = _desc(arch)
        if blob is not None:
            blob = np.ascontiguousarray(blob, dtype=np.float32)
            _lib.check(self.lib.ts2d_engine_create(ctypes.byref(d), blob.ctypes.data, blob.size, self.device,
                                                   ctypes.byref(self._h)), 'ts2d_engine_create')
        else:
            _lib.check(self.lib.ts2d_engine_create(ctypes.byref(d), None, 0, self.device, ctypes.byref(self._h)),
                       'ts2d_engine_create')
        for k, v in {**Engine.default_options, **(options or {})}.items():
            self.set_option(k, v)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, '_h', None) is not None and self._h.value:
            self.lib.ts2d_engine_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------ weights
    def load_weights(self, blob: np.ndarray):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        _lib.check(self.lib.ts2d_engine_load_weights(self._h, blob.ctypes.data, blob.size), 'ts2d_engine_load_weights')

    def weight_buffer(self) -> Tuple[int, int]:
        """(device pointer, bytes) of the packed weight arena - the RCCL broadcast payload."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        _lib.check(self.lib.ts2d_engine_weight_buffer(self._h, ctypes.byref(p), ctypes.byref(n)), 'ts2d_engine_weight_buffer')
        return int(p.value), int(n.value)

    def weights_ready(self):
        _lib.check(self.lib.ts2d_engine_weights_ready(self._h), 'ts2d_engine_weights_ready')

    def set_precision(self, mode):
        """'exact' (fp32 MFMA), 'split' (fp16 hi/lo x3 MFMA, fp32-equivalent accuracy; the default) or 'f16' (fp16 storage,
        one fp16 MFMA product, fp32 accumulate/statistics: BASELINE configs 3/5, outside the fp32 parity tolerance)."""
        m = {'exact': _lib.PRECISION_F32_EXACT, 'split': _lib.PRECISION_F32_SPLIT_F16X3, 'f16': _lib.PRECISION_F16}.get(mode, mode)
        _lib.check(self.lib.ts2d_engine_set_precision(self._h, int(m)), 'ts2d_engine_set_precision')
        self._ws_need.clear()

    def set_option(self, name: str, value: int):
        """Kernel-dispatch option of this handle (C-ABI ts2d_engine_set_option, include/ts2d_engine.h lists the names): picks
        between two parity-tested kernels for the ops it names; takes effect at the next forward."""
        _lib.check(self.lib.ts2d_engine_set_option(self._h, name.encode(), int(value)), f'ts2d_engine_set_option({name})')
        self._ws_need.clear()

    def set_tile_dtype(self, mode):
        """Blend order of :meth:`predict_tiled`: 'float' (reference CPU path: fp32 tile, one rounding into the half buffer;
        the default) or 'half' (CUDA autocast path: half tile, half product, half sum)."""
        m = {'float': 0, 'half': 1}.get(mode, mode)
        _lib.check(self.lib.ts2d_engine_set_tile_dtype(self._h, int(m)), 'ts2d_engine_set_tile_dtype')

    def keep_activations(self, on: bool = True):
        """One buffer per activation (C-ABI ts2d_engine_set_keep_activations) instead of liveness-based sharing; needed to read
        intermediate tensors back with :meth:`debug_tensor` (which switches it on itself and re-runs the last forward)."""
        _lib.check(self.lib.ts2d_engine_set_keep_activations(self._h, int(bool(on))), 'ts2d_engine_set_keep_activations')
        self._ws_need.clear()
        self._keep = bool(on)
        self._auto_keep = False

    # ------------------------------------------------------------------ forward
    def reserve(self, B: int, H: int, W: int):
        _lib.check(self.lib.ts2d_engine_reserve(self._h, B, H, W), 'ts2d_engine_reserve')

    def workspace_bytes(self, B: int, H: int, W: int) -> int:
        """Bytes of activation workspace :meth:`reserve` would allocate for (B, H, W) in the current precision mode.
        (The C-ABI call re-plans every op; the answer is remembered until the mode, an option or the keep flag changes -
        ``SubModelSet.forward_masks`` asks before every run.)"""
        key = (int(B), int(H), int(W))
        if key not in self._ws_need:
            n = ctypes.c_size_t()
            _lib.check(self.lib.ts2d_engine_workspace_bytes(self._h, B, H, W, ctypes.byref(n)), 'ts2d_engine_workspace_bytes')
            self._ws_need[key] = int(n.value)
        return self._ws_need[key]

    def set_workspace(self, dev_ptr: Optional[int], n_bytes: int = 0):
        """Run inside caller-owned device memory (C-ABI ts2d_engine_set_workspace; None: back to the engine's own allocation).
        Engines sharing one workspace must be driven on one stream."""
        _lib.check(self.lib.ts2d_engine_set_workspace(self._h, ctypes.c_void_p(dev_ptr) if dev_ptr else None, int(n_bytes)),
                   'ts2d_engine_set_workspace')

    def forward(self, x, logits=True, mask=False, out_logits=None, out_mask=None, stream: int = 0, _debug_rerun: bool = False):
        """x: [B,C,H,W] fp32, numpy (host) or torch CUDA tensor (device, zero-copy).
        Returns (logits or None, packed mask or None) of the same kind as x."""
        K = self.arch.num_classes
        if self._auto_keep and not _debug_rerun:      # debug_tensor left private buffers behind: a production call returns to the shared arena
            _lib.check(self.lib.ts2d_engine_set_keep_activations(self._h, 0), 'ts2d_engine_set_keep_activations')
            self._ws_need.clear()
            self._auto_keep = False
        self._ran = True
        try:
            self._last = (weakref.ref(x), logits, mask)
        except TypeError:
            self._last = None
        if _is_torch(x):
            import torch
            if not x.is_cuda:
                raise RuntimeError("torch input must live on the GPU (pass numpy for host data)")
            x = x.contiguous()
            if x.dtype != torch.float32:
                raise RuntimeError(f"input must be float32, found {x.dtype}")
            B, C, H, W = x.shape
            self._check_shape(C, W, mask)
            if logits and out_logits is None:
                out_logits = torch.empty((B, K, H, W), dtype=torch.float32, device=x.device)
            if mask and out_mask is None:
                out_mask = torch.empty((B, K, H, W // 32), dtype=torch.int32, device=x.device)
            side = None
            if stream == 0:
                # Order the kernels with the caller's torch work.  A non-default current stream is used directly; torch's DEFAULT
                # stream has the null handle, which the C-ABI reads as "the engine's own stream" - so run on a side stream that
                # waits for the current stream and make the current stream wait for it afterwards (no host synchronisation).
                cur = torch.cuda.current_stream(x.device)
                if cur.cuda_stream != 0:
                    stream = cur.cuda_stream
                else:
                    if getattr(self, '_side_stream', None) is None:
                        self._side_stream = torch.cuda.Stream(device=x.device)
                    side = self._side_stream
                    side.wait_stream(cur)
                    stream = side.cuda_stream
                    for t in (x, out_logits if logits else None, out_mask if mask else None):
                        if t is not None:
                            t.record_stream(side)
            _lib.check(self.lib.ts2d_engine_forward(
                self._h, x.data_ptr(), B, H, W, out_logits.data_ptr() if logits else None,
                out_mask.data_ptr() if mask else None, 1, ctypes.c_void_p(stream)), 'ts2d_engine_forward')
            if side is not None:
                torch.cuda.current_stream(x.device).wait_stream(side)
            return (out_logits if logits else None), (out_mask if mask else None)
        x = np.ascontiguousarray(x, dtype=np.float32)
        B, C, H, W = x.shape
        self._check_shape(C, W, mask)
        if logits and out_logits is None:
            out_logits = np.empty((B, K, H, W), dtype=np.float32)
        if mask and out_mask is None:
            out_mask = np.empty((B, K, H, W // 32), dtype=np.uint32)
        _lib.check(self.lib.ts2d_engine_forward(
            self._h, x.ctypes.data, B, H, W, out_logits.ctypes.data if logits else None,
            out_mask.ctypes.data if mask else None, 0, None), 'ts2d_engine_forward')
        return (out_logits if logits else None), (out_mask if mask else None)

    def check(self):
        """Synchronise the last forward and raise RuntimeError if it produced inf / NaN logits, naming the first layer whose
        output is non-finite (C-ABI ts2d_engine_check).  numpy forwards and predict_tiled run it themselves."""
        _lib.check(self.lib.ts2d_engine_check(self._h), 'ts2d_engine_check')

    def predict_tiled(self, image: np.ndarray, patch, tiles, mirror_axes=None, gaussian: Optional[np.ndarray] = None,
                      want_logits: bool = True, want_seg: bool = False, out_logits: Optional[np.ndarray] = None):
        """Device-side sliding window for one padded 2-D image [C,Hp,Wp] (C-ABI ts2d_engine_predict_tiled).
        tiles: [(y, x), ...] in upstream order; gaussian: float16 [ph,pw] or None.  Returns (float16 [K,Hp,Wp] or None,
        uint8 [K,Hp,Wp] or None)."""
        image = np.ascontiguousarray(image, dtype=np.float32)
        C, Hp, Wp = image.shape
        if C != self.arch.input_channels:
            raise RuntimeError(f"input has {C} channels, the model expects {self.arch.input_channels}")
        ty = np.ascontiguousarray([t[0] for t in tiles], dtype=np.int32)
        tx = np.ascontiguousarray([t[1] for t in tiles], dtype=np.int32)
        mask = 0
        for a in (mirror_axes or ()):
            mask |= 1 << int(a)
        g = None if gaussian is None else np.ascontiguousarray(gaussian, dtype=np.float16)
        K = self.arch.num_classes
        if out_logits is not None and not (out_logits.dtype == np.float16 and out_logits.shape == (K, Hp, Wp) and out_logits.flags.c_contiguous):
            raise RuntimeError("out_logits must be a C-contiguous float16 [K,Hp,Wp] array")
        out16 = (out_logits if out_logits is not None else np.empty((K, Hp, Wp), dtype=np.float16)) if want_logits else None
        seg = np.empty((K, Hp, Wp), dtype=np.uint8) if want_seg else None
        _lib.check(self.lib.ts2d_engine_predict_tiled(
            self._h, image.ctypes.data, Hp, Wp, int(patch[0]), int(patch[1]), len(tiles), ty.ctypes.data, tx.ctypes.data, mask,
            None if g is None else g.ctypes.data, None if out16 is None else out16.ctypes.data,
            None if seg is None else seg.ctypes.data), 'ts2d_engine_predict_tiled')
        self.last_tiled_inf = bool(self.lib.ts2d_engine_tiled_inf_flag(self._h))     # upstream's inf check, done on the device
        return out16, seg

    def _check_shape(self, C, W, mask):
        if C != self.arch.input_channels:
            raise RuntimeError(f"input has {C} channels, the model expects {self.arch.input_channels}")
        if mask and W % 32:
            raise RuntimeError("packed mask output needs W % 32 == 0")

    # ------------------------------------------------------------------ profiling
    def set_profiling(self, on: bool):
        _lib.check(self.lib.ts2d_engine_set_profiling(self._h, int(on)), 'ts2d_engine_set_profiling')

    def op_times(self) -> Dict[str, float]:
        """ms per launch of the LAST forward (HIP events on the launch stream), program order."""
        n = self.lib.ts2d_engine_num_ops(self._h)
        ms = (ctypes.c_float * max(n, 1))()
        _lib.check(self.lib.ts2d_engine_op_times(self._h, ms, n), 'ts2d_engine_op_times')
        return {self.lib.ts2d_engine_op_name(self._h, i).decode(): float(ms[i]) for i in range(n)}

    def op_kernels(self) -> dict:
        """op name -> name of the kernel that served it in the last profiled forward."""
        n = self.lib.ts2d_engine_num_ops(self._h)
        return {self.lib.ts2d_engine_op_name(self._h, i).decode(): self.lib.ts2d_engine_op_kernel(self._h, i).decode() for i in range(n)}

    def debug_tensor(self, name: str, capacity: int = 1 << 26) -> np.ndarray:
        """Test accessor: activation `name` of the last forward as torch would hold it (NCHW, norm+act applied).  Activations
        share buffers by liveness: the first call switches the engine to private buffers and runs the last forward again - on the
        SAME input object, which the engine holds only weakly: keep a reference to the array / tensor you passed to forward()
        (``e.forward(x[None])`` or ``e.forward(a.astype(np.float32))`` leave nothing to re-run), or call ``keep_activations(True)``
        before the forward.  With the first block fused into the second (split mode) ``enc0.c0`` is not materialised: create the
        engine with ``options={'fuse0': 0}`` to read it."""
        if not (self._keep or self._auto_keep):
            if not self._ran:
                raise RuntimeError("debug_tensor: no forward has run on this engine")
            x = self._last[0]() if self._last is not None else None
            if x is None:
                raise RuntimeError("debug_tensor: the input of the last forward has been garbage-collected (the engine holds it weakly so "
                                   "that a production call does not pin the batch): keep a reference to the object passed to forward(), "
                                   "or call keep_activations(True) before the forward")
            _lib.check(self.lib.ts2d_engine_set_keep_activations(self._h, 1), 'ts2d_engine_set_keep_activations')
            self._ws_need.clear()
            self._auto_keep = True
            self.forward(x, logits=self._last[1], mask=self._last[2], _debug_rerun=True)
            if _is_torch(x):
                import torch
                torch.cuda.synchronize(x.device)
        out = np.empty(capacity, dtype=np.float32)
        dims = (ctypes.c_int32 * 4)()
        _lib.check(self.lib.ts2d_engine_debug_tensor(self._h, name.encode(), out.ctypes.data, out.size, ctypes.byref(dims)),
                   'ts2d_engine_debug_tensor')
        shp = tuple(int(d) for d in dims)
        return out[:int(np.prod(shp))].reshape(shp).copy()

    def materialised(self, name: str) -> bool:
        """Test accessor: False if the last forward composed the op that produces `name` into its consumer (a transposed conv
        folded into the next block, ``csrc/kernels_upc.h``; the first block recomputed inside the second), so that the tensor was
        never written.  Like :meth:`debug_tensor` it re-runs the last forward: keep a reference to its input."""
        try:
            self.debug_tensor(name)
            return True
        except RuntimeError as ex:
            if 'not materialised' in str(ex):
                return False
            raise

    def device_bytes(self) -> int:
        return int(self.lib.ts2d_engine_device_bytes(self._h))


def unpack_mask(packed: np.ndarray, W: int) -> np.ndarray:
    """[.., W/32] uint32 -> [.., W] uint8 {0,1}."""
    p = np.asarray(packed).view(np.uint32)
    bits = (p[..., None] >> np.arange(32, dtype=np.uint32)) & np.uint32(1)
    return bits.reshape(p.shape[:-1] + (W,)).astype(np.uint8)

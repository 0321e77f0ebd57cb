"""TEST INFRASTRUCTURE (never imported by the product): the host-side restatement of what ``ts2d_engine_predict_tiled`` does on the
device - tiles x mirror variants, mirror average, upstream's float16 Gaussian aggregation in both blend orders - as a subclass of the
product's drop-in predictor whose network is an INJECTED callable (the torch oracle on machines without a GPU; the engine's own
``forward`` where device and host aggregation are compared bit for bit).  The product class has no such hook: it always builds HIP
engines and always aggregates on the device.

Restates upstream ``predict_sliding_window_return_logits`` / ``_internal_maybe_mirror_and_predict`` (reached from reference
``ts2d/core/inference/prediction_worker.py:209``)."""
from __future__ import annotations

import numpy as np

from totalsegmentator2d_amd import sliding_window as sw
from totalsegmentator2d_amd.predictor import HIPnnUNetPredictor


def _any_inf_f16(x: np.ndarray) -> bool:
    """np.any(np.isinf(x)) for float16 arrays on the bit pattern (exponent all ones, mantissa zero): numpy's float16
    isinf converts element-wise and costs ~10 ms on an 18x644x512 logits array; this is the same predicate."""
    if x.dtype != np.float16:
        return bool(np.any(np.isinf(x)))
    v = np.ascontiguousarray(x).view(np.uint16)
    return bool(np.any((v & np.uint16(0x7FFF)) == np.uint16(0x7C00)))



class HostLogicPredictor(HIPnnUNetPredictor):
    def __init__(self, network=None, **kw):
        """network: callable ([B,C,h,w] float32[, fold]) -> [B,K,h,w]; None: only the initialisation paths are exercised."""
        super().__init__(**kw)
        self._network = network

    def _create_engines(self):               # no GPU, no engine
        self.engines = []

    def _run_network(self, fold: int, batch: np.ndarray) -> np.ndarray:
        net = self._network
        return np.asarray(net(batch, fold) if net.__code__.co_argcount > 1 else net(batch), dtype=np.float32)

    def predict_sliding_window_return_logits(self, data: np.ndarray, fold: int = 0) -> np.ndarray:
        patch = tuple(self.configuration_manager.patch_size)
        data = np.asarray(data, dtype=np.float32)
        if data.ndim != 4:
            raise AssertionError('input_image must be a 4D np.ndarray or torch.Tensor (c, x, y, z)')
        padded, revert = sw.pad_nd_image(data, patch)
        C, Z, H, W = padded.shape
        slicers = sw.tile_slicers((H, W), patch, self.tile_step_size, Z)
        combos = sw.mirror_combos(self.allowed_mirroring_axes if self.use_mirroring else None)
        if self.use_mirroring and self.allowed_mirroring_axes and max(self.allowed_mirroring_axes) > 1:
            raise AssertionError('mirror_axes does not match the dimension of the input!')
        nv = len(combos)
        batch = np.empty((len(slicers) * nv, C, patch[0], patch[1]), dtype=np.float32)
        for t, (d, sx, sy) in enumerate(slicers):
            x = padded[:, d, sx:sx + patch[0], sy:sy + patch[1]]
            for v, c in enumerate(combos):
                batch[t * nv + v] = np.flip(x, [a - 1 for a in c]) if c else x      # tensor dim a of [1,C,h,w] = dim a-1 here
        y = self._run_network(fold, batch)
        K = y.shape[1]
        g = sw.compute_gaussian(patch) if self.use_gaussian else np.ones(patch, dtype=np.float16)
        logits = np.zeros((K, Z, H, W), dtype=np.float16)
        n_pred = np.zeros((Z, H, W), dtype=np.float16)
        for t, (d, sx, sy) in enumerate(slicers):
            p = y[t * nv].copy()
            for v in range(1, nv):
                p += np.flip(y[t * nv + v], [a - 1 for a in combos[v]])
            if nv > 1:
                p /= np.float32(nv)
            sl = (slice(None), d, slice(sx, sx + patch[0]), slice(sy, sy + patch[1]))
            if self.tile_dtype == 'half':          # CUDA autocast order: half tile, half product, half sum (three roundings)
                p = p.astype(np.float16)
                if self.use_gaussian:
                    p = p * g
                logits[sl] += p
            else:                                  # reference CPU path: fp32 tile * float(g) in fp32, ONE rounding into the half buffer
                if self.use_gaussian:
                    p = p * g.astype(np.float32)
                logits[sl] = (logits[sl].astype(np.float32) + p).astype(np.float16)
            n_pred[d, sx:sx + patch[0], sy:sy + patch[1]] += g
        logits = logits / n_pred
        if _any_inf_f16(logits):
            raise RuntimeError('Encountered inf in predicted array. Aborting... If this problem persists, reduce '
                               'value_scaling_factor in compute_gaussian or increase the dtype of predicted_logits to fp32')
        return logits[(slice(None),) + revert[1:]]

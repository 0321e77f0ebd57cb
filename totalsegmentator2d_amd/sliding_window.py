"""Host-side tiling logic of nnU-Net's sliding-window inference, re-designed for
batched submission: every tile x mirror variant of a case becomes one row of a single engine batch instead of the
reference's B=1 ``network()`` calls.

The algorithm lives in the third-party wheel ``nnunetv2ml==2.6.2`` (reference ``pyproject.toml:25``; entered from
``predictor.predict_logits_from_preprocessed_data`` at reference ``ts2d/core/inference/prediction_worker.py:209``);
upstream names are kept: ``compute_steps_for_sliding_window``, ``compute_gaussian``, ``pad_nd_image``.
"""
from __future__ import annotations

import itertools
from typing import List, Optional, Sequence, Tuple

import numpy as np


def compute_steps_for_sliding_window(image_size: Sequence[int], tile_size: Sequence[int], tile_step_size: float) -> List[List[int]]:
    """num = ceil((img - tile) / (tile * step)) + 1 tiles per axis at positions round(i * (img - tile) / (num - 1))."""
    if not all(i >= j for i, j in zip(image_size, tile_size)):
        raise AssertionError("image size must be as large or larger than patch_size")
    if not 0 < tile_step_size <= 1:
        raise AssertionError('step_size must be larger than 0 and smaller or equal to 1')
    target = [i * tile_step_size for i in tile_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, tile_size)]
    steps = []
    for dim in range(len(tile_size)):
        max_step_value = image_size[dim] - tile_size[dim]
        actual = max_step_value / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


def _gaussian_kernel1d(sigma: float, radius: int) -> np.ndarray:
    """scipy.ndimage._filters._gaussian_kernel1d(order=0): exp(-0.5 / sigma^2 * x^2) normalised to sum 1."""
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


_GAUSS_CACHE = {}


def compute_gaussian(tile_size: Sequence[int], sigma_scale: float = 1. / 8, value_scaling_factor: float = 10,
                     dtype=np.float16) -> np.ndarray:
    """Gaussian importance map: ``gaussian_filter`` (truncate 4 sigma, mode constant) of a centre delta, divided by
    max / value_scaling_factor, cast to ``dtype`` (float16 upstream), zeros replaced by the smallest non-zero value.
    Filtering a delta is separable, so the map is the outer product of the 1-D kernels - no scipy needed."""
    key = (tuple(int(t) for t in tile_size), float(sigma_scale), float(value_scaling_factor), np.dtype(dtype).str)
    if key in _GAUSS_CACHE:
        return _GAUSS_CACHE[key]
    axes = []
    for n in tile_size:
        sigma = n * sigma_scale
        radius = int(4.0 * sigma + 0.5)
        k = _gaussian_kernel1d(sigma, radius)
        c = n // 2
        line = np.zeros(n, dtype=np.float64)
        for i in range(n):
            d = i - c
            if -radius <= d <= radius:
                line[i] = k[d + radius]
        axes.append(line)
    g = axes[0]
    for a in axes[1:]:
        g = np.multiply.outer(g, a)
    g = g / (g.max() / value_scaling_factor)
    g = g.astype(dtype)
    mask = g == 0
    if mask.any():
        g[mask] = g[~mask].min()
    g.setflags(write=False)
    _GAUSS_CACHE[key] = g
    return g


def pad_nd_image(data: np.ndarray, new_shape: Sequence[int]) -> Tuple[np.ndarray, Tuple[slice, ...]]:
    """``pad_nd_image(data, patch, 'constant', {'value': 0}, True)`` on the trailing dims: symmetric zero pad up to the
    patch (below = diff // 2, above = diff // 2 + diff % 2); returns (padded, slicer that undoes it)."""
    nd = len(new_shape)
    old = data.shape[-nd:]
    new = [max(n, o) for n, o in zip(new_shape, old)]
    diff = [n - o for n, o in zip(new, old)]
    below = [d // 2 for d in diff]
    above = [d // 2 + d % 2 for d in diff]
    pad = [(0, 0)] * (data.ndim - nd) + list(zip(below, above))
    out = np.pad(data, pad, mode='constant', constant_values=0) if any(diff) else data
    slicer = tuple([slice(None)] * (data.ndim - nd) + [slice(b, b + o) for b, o in zip(below, old)])
    return out, slicer


def mirror_combos(mirror_axes: Optional[Sequence[int]]) -> List[Tuple[int, ...]]:
    """() plus every non-empty subset of the allowed mirroring axes (spatial axis m -> tensor dim m + 2)."""
    if not mirror_axes:
        return [()]
    axes = [m + 2 for m in mirror_axes]
    return [()] + [c for i in range(len(axes)) for c in itertools.combinations(axes, i + 1)]


def tile_slicers(padded_hw: Sequence[int], patch: Sequence[int], step: float, Z: int) -> List[Tuple[int, int, int]]:
    """``_internal_get_sliding_window_slicers`` for a 2-D net on [C,Z,H,W]: (z, sx, sy) in upstream order."""
    steps = compute_steps_for_sliding_window(padded_hw, patch, step)
    return [(d, sx, sy) for d in range(Z) for sx in steps[0] for sy in steps[1]]

"""ORACLE (test infrastructure, never shipped or timed as the product): torch-CPU restatement of the hot path.

PARITY PINNING: the reference pins no numerical result for this path (its tests assert type / file existence
only: reference ``test/test_020_predict_default.py:37-38``, ``test/test_030_cli.py:14-15``) and the arithmetic
lives in third-party wheels that are absent offline (``nnunetv2ml==2.6.2`` ``pyproject.toml:25`` ->
``dynamic_network_architectures`` -> ``torch``).  This file therefore restates the PUBLISHED upstream algorithm
with the SAME ATen CPU kernels the reference dispatches to on its CPU path (``ts2d/core/inference/nnu.py:161-163``:
``device=cpu`` when ``torch.cuda.is_available()`` is false): ``F.conv2d``, ``F.instance_norm``, ``F.leaky_relu``, ``F.conv_transpose2d``, ``torch.cat``.
Golden fixtures generated from it live in tests/golden/ (script tests/gen_golden.py).  Status: "parity pinned to the
reference's arithmetic kernels, unpinned w.r.t. reference-produced vectors" (DESIGN.md section 3).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Rows restated:
  A6/K1-K7  ``PlainConvUNet.forward``                                     -> :func:`unet_forward`
  A5        ``_internal_maybe_mirror_and_predict``                        -> :func:`mirror_and_predict`
  A3        ``compute_steps_for_sliding_window`` / slicers                -> :func:`sliding_window_steps`
  A4        ``compute_gaussian`` + fp16 accumulation                      -> :func:`compute_gaussian`, :func:`predict_sliding_window`
  A2        fold ensembling                                               -> :func:`predict_logits`
  (tests)   one block from given inputs                                   -> :func:`layer_forward`
  A7        multilabel export ``sigmoid(logits.float()) > 0.5``           -> :func:`logits_to_mask`
  A1        ZScoreNormalization                                           -> :func:`zscore`
"""
from __future__ import annotations

import itertools
from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))


# ----------------------------------------------------------------------------- K1-K7 / A6
def _h(t):
    """Round to IEEE half and back: what a value becomes when the 16-bit mode stores it (or feeds it to an fp16 MFMA operand)."""
    return t.to(torch.float16).to(torch.float32)


def conv_block(x, w, b, g, be, stride, eps=1e-5, slope=0.01, emulate=None, round_w=True, storage_view=False):
    """``ConvDropoutNormReLU``: Conv2d(3x3, pad 1, stride = int or (sy, sx)) -> InstanceNorm2d(affine, eps, biased var) -> LeakyReLU.

    ``emulate='f16'`` restates the arithmetic CONTRACT of the engine's 16-bit mode (include/ts2d_engine.h, TS2D_PRECISION_F16; to first
    order also the reference's CUDA path, which runs the network under fp16 autocast: ``ts2d/core/inference/nnu.py:159-163``):
    weights rounded to fp16 once, products accumulated in fp32 (F.conv2d on fp16-representable fp32 tensors), bias added in fp32, the
    conv output STORED as fp16, InstanceNorm statistics in fp32 over the stored values, the normalised value rounded to fp16 and the
    LeakyReLU done in fp16 (``max(y, fp16(y * fp16(slope)))``) - the operand the next conv multiplies.  ``round_w=False``: the first
    block, which the engine computes from the fp32 network input with fp32 weights (exact fp32 MFMA) and only stores as fp16.
    ``storage_view=True``: the block's output as the engine's debug accessor shows it - the STORED fp16 conv output normalised and
    activated in fp32, without the operand rounding of the consumer - so that a per-layer comparison is not blurred by half an fp16
    ulp on every element."""
    if emulate is None:
        y = F.conv2d(x, w, b, stride=stride, padding=1)
        y = F.instance_norm(y, None, None, g, be, use_input_stats=True, momentum=0.1, eps=eps)
        return F.leaky_relu(y, slope)
    assert emulate == 'f16'
    y = _h(F.conv2d(x, _h(w) if round_w else w, b, stride=stride, padding=1))
    y = F.instance_norm(y, None, None, g, be, use_input_stats=True, momentum=0.1, eps=eps)
    if storage_view:
        return F.leaky_relu(y, slope)
    y = _h(y)
    s16 = float(torch.tensor(slope, dtype=torch.float16))
    return torch.maximum(y, _h(y * s16))


def unet_forward(arch, sd: Dict[str, np.ndarray], x, return_intermediates: bool = False, emulate=None):
    """``PlainConvUNet.forward`` (deep supervision off): encoder stages (strided first conv), then per decoder
    stage ``transpconv -> cat((up, skip), 1) -> conv blocks``, finally ``seg_layers[-1]``.  x: [B,C,H,W] fp32.
    ``emulate='f16'``: the 16-bit mode's contract (see :func:`conv_block`); the transposed conv's output is stored as fp16 without
    normalisation, the head multiplies fp16 weights with the fp16 activations and returns fp32 logits."""
    sd = {k: _t(v) for k, v in sd.items()}
    x = _t(x).to(torch.float32)
    f16 = emulate == 'f16'
    inter = {}
    skips = []
    with torch.no_grad():
        for s in range(arch.n_stages):
            for i in range(arch.n_conv_per_stage[s]):
                k = f'encoder.stages.{s}.0.convs.{i}'
                x = conv_block(x, sd[f'{k}.conv.weight'], sd[f'{k}.conv.bias'], sd[f'{k}.norm.weight'],
                               sd[f'{k}.norm.bias'], tuple(arch.strides[s]) if (i == 0 and s > 0) else 1, arch.norm_eps, arch.leaky_slope,
                               emulate=emulate, round_w=not (s == 0 and i == 0))
                inter[f'enc{s}.c{i}'] = x
            skips.append(x)
        for j in range(arch.n_stages - 1):
            lvl = arch.n_stages - 2 - j
            k = f'decoder.transpconvs.{j}'
            st = tuple(arch.strides[lvl + 1])          # upstream UNetDecoder: kernel = stride = the stride of the stage below
            if f16:
                x = _h(F.conv_transpose2d(x, _h(sd[f'{k}.weight']), sd[f'{k}.bias'], stride=st))
            else:
                x = F.conv_transpose2d(x, sd[f'{k}.weight'], sd[f'{k}.bias'], stride=st)
            inter[f'dec{lvl}.up'] = x
            x = torch.cat((x, skips[lvl]), 1)
            for i in range(arch.n_conv_per_stage_decoder[j]):
                k = f'decoder.stages.{j}.convs.{i}'
                x = conv_block(x, sd[f'{k}.conv.weight'], sd[f'{k}.conv.bias'], sd[f'{k}.norm.weight'],
                               sd[f'{k}.norm.bias'], 1, arch.norm_eps, arch.leaky_slope, emulate=emulate)
                inter[f'dec{lvl}.c{i}'] = x
        k = f'decoder.seg_layers.{arch.n_stages - 2}'
        x = F.conv2d(x, _h(sd[f'{k}.weight']) if f16 else sd[f'{k}.weight'], sd[f'{k}.bias'])
    return (x, inter) if return_intermediates else x


def layer_forward(arch, sd: Dict[str, np.ndarray], name: str, src, skip=None, emulate=None, storage_view=False):
    """ONE block of :func:`unet_forward` in isolation, from the activated tensors it reads (as torch holds them after the previous
    block): ``encS.cI`` / ``decL.cI`` (I > 0) from ``src``; ``decL.c0`` from ``src`` = the COARSE tensor the transposed conv reads and
    ``skip`` (transpconv -> cat((up, skip), 1) -> conv block); ``head`` from ``src``.  Per-layer parity tests feed it the engine's own
    inputs of that layer, so that an error cannot hide behind the accumulated difference of the layers in front of it.
    With ``emulate='f16'`` the given inputs are rounded to fp16 first (the operand rounding of the consumer);
    ``storage_view``: see :func:`conv_block`."""
    sd = {k: _t(v) for k, v in sd.items()}
    x = _t(src).to(torch.float32)
    f16 = emulate == 'f16'
    if f16 and name != 'enc0.c0':
        # the engine's debug accessor returns activations BEFORE the consumer's operand rounding: round them here (a no-op for
        # inputs taken from unet_forward(emulate='f16'); on the negative LeakyReLU branch fp16(y * 0.01f) stands in for the
        # engine's fp16(fp16(y) * fp16(0.01)): values 100x smaller than the rest, relative difference 7e-4)
        x = _h(x)
        if skip is not None:
            skip = _h(_t(skip).to(torch.float32))
    with torch.no_grad():
        if name == 'head':
            k = f'decoder.seg_layers.{arch.n_stages - 2}'
            return F.conv2d(x, _h(sd[f'{k}.weight']) if f16 else sd[f'{k}.weight'], sd[f'{k}.bias'])
        kind, lvl, i = name[:3], int(name[3:name.index('.')]), int(name[name.index('.c') + 2:])
        if kind == 'enc':
            k = f'encoder.stages.{lvl}.0.convs.{i}'
            stride = tuple(arch.strides[lvl]) if (i == 0 and lvl > 0) else 1
            rw = not (lvl == 0 and i == 0)
        else:
            j = arch.n_stages - 2 - lvl
            k = f'decoder.stages.{j}.convs.{i}'
            stride, rw = 1, True
            if i == 0:
                kt = f'decoder.transpconvs.{j}'
                up = F.conv_transpose2d(x, _h(sd[f'{kt}.weight']) if f16 else sd[f'{kt}.weight'], sd[f'{kt}.bias'],
                                        stride=tuple(arch.strides[lvl + 1]))
                x = torch.cat((_h(up) if f16 else up, _t(skip).to(torch.float32)), 1)
        return conv_block(x, sd[f'{k}.conv.weight'], sd[f'{k}.conv.bias'], sd[f'{k}.norm.weight'], sd[f'{k}.norm.bias'],
                          stride, arch.norm_eps, arch.leaky_slope, emulate=emulate, round_w=rw, storage_view=storage_view)


# ----------------------------------------------------------------------------- A7
def logits_to_mask(logits) -> torch.Tensor:
    """multilabel export: ``sigmoid(logits.float()) > 0.5`` per channel -> uint8."""
    return (torch.sigmoid(_t(logits).float()) > 0.5).to(torch.uint8)


def pack_mask(mask_u8: np.ndarray) -> np.ndarray:
    """[.., W] uint8 {0,1} -> [.., W/32] uint32, bit i of word j = pixel 32*j+i (the engine's packed-mask layout)."""
    m = np.asarray(mask_u8, dtype=np.uint32)
    assert m.shape[-1] % 32 == 0
    m = m.reshape(m.shape[:-1] + (m.shape[-1] // 32, 32))
    return (m << np.arange(32, dtype=np.uint32)).sum(-1).astype(np.uint32)


# ----------------------------------------------------------------------------- A5
def mirror_and_predict(net, x, mirror_axes: Sequence[int] | None):
    """``_internal_maybe_mirror_and_predict``: y = net(x) + sum over non-empty subsets of (axes+2) of
    flip(net(flip(x))) ; y /= 2^len(axes)."""
    y = net(x)
    if mirror_axes:
        axes = [m + 2 for m in mirror_axes]
        combos = [c for i in range(len(axes)) for c in itertools.combinations(axes, i + 1)]
        for c in combos:
            y = y + torch.flip(net(torch.flip(x, c)), c)
        y = y / (len(combos) + 1)
    return y


# ----------------------------------------------------------------------------- A3
def sliding_window_steps(image_size: Sequence[int], tile_size: Sequence[int], tile_step_size: float) -> List[List[int]]:
    """``compute_steps_for_sliding_window``: num = ceil((img - tile) / (tile*step)) + 1; positions round(i * actual)."""
    assert all(i >= j for i, j in zip(image_size, tile_size)), "image size must be >= patch size in all dims"
    assert 0 < tile_step_size <= 1
    target = [i * tile_step_size for i in tile_size]
    num = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, tile_size)]
    steps = []
    for d in range(len(tile_size)):
        max_step = image_size[d] - tile_size[d]
        actual = max_step / (num[d] - 1) if num[d] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num[d])])
    return steps


def pad_to_patch(data: torch.Tensor, patch: Sequence[int]):
    """``pad_nd_image(data, patch, 'constant', {'value': 0}, True, None)`` on the trailing dims: symmetric pad
    (below = diff // 2, above = diff // 2 + diff % 2); returns (padded, slicer to undo)."""
    shp = data.shape[-len(patch):]
    new = [max(p, s) for p, s in zip(patch, shp)]
    diff = [n - s for n, s in zip(new, shp)]
    below = [d // 2 for d in diff]
    above = [d // 2 + d % 2 for d in diff]
    pads = []
    for b, a in zip(reversed(below), reversed(above)):
        pads += [b, a]
    out = F.pad(data, pads, mode='constant', value=0) if any(diff) else data
    slicer = tuple([slice(None)] * (data.ndim - len(patch)) + [slice(b, b + s) for b, s in zip(below, shp)])
    return out, slicer


# ----------------------------------------------------------------------------- A4
def compute_gaussian(tile_size: Sequence[int], sigma_scale: float = 1. / 8, value_scaling_factor: float = 10,
                     dtype=torch.float16) -> torch.Tensor:
    """``compute_gaussian``: delta at the centre -> scipy gaussian_filter(sigma = tile*sigma_scale) -> / (max /
    value_scaling_factor) -> cast -> zeros replaced by the minimum non-zero value."""
    from scipy.ndimage import gaussian_filter
    tmp = np.zeros(tile_size)
    tmp[tuple(i // 2 for i in tile_size)] = 1
    g = gaussian_filter(tmp, [i * sigma_scale for i in tile_size], 0, mode='constant', cval=0)
    g = torch.from_numpy(g)
    g = g / (torch.max(g) / value_scaling_factor)
    g = g.to(dtype)
    mask = g == 0
    g[mask] = torch.min(g[~mask])
    return g


def predict_sliding_window(net, data: torch.Tensor, patch: Sequence[int], step: float = 0.5,
                           mirror_axes: Sequence[int] | None = (0, 1), use_gaussian: bool = True,
                           tile_dtype: str = 'float') -> torch.Tensor:
    """``predict_sliding_window_return_logits`` for a 2-D net on ``[C,Z,H,W]`` data: pad to the patch, for every z
    and every (sx, sy) tile: p = A5(x[None])[0]; p *= g; logits[sl] += p; n[sl[1:]] += g; logits /= n; un-pad.
    Accumulators and the gaussian are float16 exactly as upstream (results_device = cpu).

    ``tile_dtype`` is the dtype the tile prediction ``p`` has when it meets the half buffers:
      'float' (default) - the reference's CPU path (``nnu.py:161-163``: device=cpu when ``torch.cuda.is_available()`` is false - no autocast): ``network(x)`` is
                fp32, ``.to(results_device)`` moves device only, so ``p *= g`` is fp32 x half -> fp32 and
                ``logits[sl] += p`` is ONE rounding into the half buffer (ATen computes half += float in float);
      'half'  - the CUDA path (fp16 autocast): ``p`` is half, so ``p * g`` and ``logits += p`` each round to half.
    Both orders are written with the plain ATen statements upstream uses, so ATen itself defines the rounding."""
    assert data.ndim == 4
    data, revert = pad_to_patch(data, patch)
    C, Z, H, W = data.shape
    steps = sliding_window_steps((H, W), patch, step)
    g = compute_gaussian(tuple(patch)) if use_gaussian else torch.ones(tuple(patch), dtype=torch.half)
    K = None
    logits = n_pred = None
    for d in range(Z):
        for sx in steps[0]:
            for sy in steps[1]:
                x = data[:, d, sx:sx + patch[0], sy:sy + patch[1]][None].float()
                p = mirror_and_predict(net, x, mirror_axes)[0]
                if logits is None:
                    K = p.shape[0]
                    logits = torch.zeros((K, Z, H, W), dtype=torch.half)
                    n_pred = torch.zeros((Z, H, W), dtype=torch.half)
                # prediction = self._internal_maybe_mirror_and_predict(...)[0].to(results_device): a device move, no cast
                p = p.to(torch.half) if tile_dtype == 'half' else p.clone()
                if use_gaussian:
                    p *= g                    # fp32 *= half computes in fp32 ('float'); half *= half rounds to half ('half')
                logits[:, d, sx:sx + patch[0], sy:sy + patch[1]] += p      # half += float: float add, one cast to half
                n_pred[d, sx:sx + patch[0], sy:sy + patch[1]] += g
    logits = logits / n_pred
    if torch.any(torch.isinf(logits)):
        raise RuntimeError('Encountered inf in predicted array.')
    return logits[(slice(None),) + revert[1:]]


# ----------------------------------------------------------------------------- A2
def predict_logits(arch, fold_state_dicts: List[dict], data, patch, step=0.5, mirror_axes=(0, 1),
                   tile_dtype: str = 'float') -> torch.Tensor:
    """``predict_logits_from_preprocessed_data``: per fold sliding-window prediction, summed, / n_folds."""
    data = _t(data).float()
    pred = None
    for sd in fold_state_dicts:
        net = lambda x, sd=sd: unet_forward(arch, sd, x)
        p = predict_sliding_window(net, data, patch, step, mirror_axes, tile_dtype=tile_dtype)
        pred = p if pred is None else pred + p
    if len(fold_state_dicts) > 1:
        pred = pred / len(fold_state_dicts)
    return pred


# ----------------------------------------------------------------------------- A1
def zscore(img: np.ndarray) -> np.ndarray:
    """``ZScoreNormalization.run`` without mask: ``image.astype(float32)``; ``(image - mean) / max(std, 1e-8)``."""
    img = img.astype(np.float32, copy=True)
    mean = img.mean()
    std = img.std()
    img -= mean
    img /= max(std, 1e-8)
    return img

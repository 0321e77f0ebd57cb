"""``HIPnnUNetPredictor`` - the drop-in for the duck-typed predictor the reference worker consumes.

The reference accepts any object whose type name contains ``nnUNetPredictor`` (or a zero-argument callable returning
one): ``ts2d/core/inference/prediction_worker.py:103-114``.  The members it touches are mirrored here with the same
names and argument meaning:

* ``.configuration_manager.patch_size`` / ``.spacing`` / ``.preprocessor_class(verbose=)``  (prediction_worker.py:76-77,194)
* ``.dataset_json`` (``channel_names``, ``file_ending``, ``labels``)                           (prediction_worker.py:78,180,230)
* ``.plans_manager``, ``.verbose``, ``.device.type``                                         (prediction_worker.py:194-207)
* ``.predict_logits_from_preprocessed_data(data[C,1,H,W]) -> Tensor[K,1,H,W]`` with ``.cpu()`` (prediction_worker.py:206-209)
* ``initialize_from_trained_model_folder(model_dir, folds, checkpoint_name)``                (nnu.py:165)
* ctor kwargs ``tile_step_size, use_mirroring, verbose, allow_tqdm, device, perform_everything_on_device`` (nnu.py:152-163)

What changes underneath: ``network(x)`` is the HIP engine (C-ABI, include/ts2d_engine.h); all tiles x mirror variants
of a case are submitted as ONE batch instead of B=1 calls; the Gaussian/fp16 aggregation follows upstream bit for
bit (float16 accumulators) so that end-of-pipeline logits are comparable with the reference's.
"""
from __future__ import annotations

import json
import os
import re
from types import SimpleNamespace
from typing import List, Optional, Sequence

import numpy as np

from . import sliding_window as sw
from .arch import UNetArch


class _Device:
    """``torch.device``-like (``.type``) without importing torch."""
    def __init__(self, index: int):
        self.type, self.index = 'cuda', index

    def __repr__(self):
        return f"device(type='cuda', index={self.index})"


class HIPnnUNetPredictor:
    def __init__(self, tile_step_size: float = 0.5, use_gaussian: bool = True, use_mirroring: bool = True,
                 perform_everything_on_device: bool = True, device=None, verbose: bool = False,
                 verbose_preprocessing: bool = False, allow_tqdm: bool = True, max_batch: int = 64, precision: str = 'split',
                 tile_dtype: Optional[str] = None):
        """``tile_dtype``: dtype of a tile prediction when it is blended into upstream's float16 buffers - 'float' (the
        reference's CPU path, taken when ``torch.cuda.is_available()`` is false, ``nnu.py:161-163``: fp32 tile x half gaussian in
        fp32, ONE rounding per ``logits[sl] += p``) or 'half' (the reference's CUDA path under fp16 autocast: the tile is half, the
        product and the sum each round to half).  Default (None): the blend order of the reference path the chosen arithmetic
        mirrors - 'float' for the fp32-parity modes ('split', 'exact': BASELINE.json compares with the CPU path), 'half' for
        ``precision='f16'`` (the autocast-like mode).
        (The network is always the HIP engine: the numpy restatement of the tiling / aggregation that the CPU tests use lives in
        tests/host_predictor.py, a subclass - nothing in this module can route around the engine.)"""
        self.tile_step_size = tile_step_size
        self.use_gaussian = use_gaussian
        self.use_mirroring = use_mirroring
        self.perform_everything_on_device = perform_everything_on_device
        self.verbose = verbose
        self.verbose_preprocessing = verbose_preprocessing
        self.allow_tqdm = allow_tqdm
        self.max_batch = int(max_batch)
        if precision not in ('split', 'exact', 'f16'):
            raise ValueError("precision must be 'split' (fp32-equivalent, default), 'exact' (fp32 MFMA) or 'f16' (like the reference's CUDA autocast path)")
        self.precision = precision
        if tile_dtype is None:
            tile_dtype = 'half' if precision == 'f16' else 'float'
        if tile_dtype not in ('float', 'half'):
            raise ValueError("tile_dtype must be 'float' (reference CPU path), 'half' (CUDA autocast path) or None (by precision)")
        self.tile_dtype = tile_dtype
        idx = 0
        if device is not None:
            idx = getattr(device, 'index', device)
            idx = 0 if idx is None else int(idx)
            if getattr(device, 'type', 'cuda') != 'cuda':
                raise RuntimeError("HIPnnUNetPredictor runs on an MI355X only; there is no CPU fallback")
        self.device = _Device(idx)
        self.engines: list = []
        self.arch: Optional[UNetArch] = None
        self.plans_manager = None
        self.configuration_manager = None
        self.dataset_json = None
        self.allowed_mirroring_axes = None
        self.list_of_parameters: List[np.ndarray] = []
        self.label_manager = None

    # ------------------------------------------------------------------ initialisation
    def manual_initialization(self, arch: UNetArch, fold_blobs: Sequence[np.ndarray], patch_size: Sequence[int],
                              spacing: Sequence[float] = (1.5, 1.5), dataset_json: Optional[dict] = None,
                              plans: Optional[dict] = None, configuration: str = '2d',
                              inference_allowed_mirroring_axes: Optional[Sequence[int]] = (0, 1)):
        from .preprocess import DefaultPreprocessor
        self.arch = arch
        self.list_of_parameters = [np.ascontiguousarray(b, dtype=np.float32) for b in fold_blobs]
        self.allowed_mirroring_axes = tuple(inference_allowed_mirroring_axes) if inference_allowed_mirroring_axes else None
        self.dataset_json = dataset_json or {
            'channel_names': {str(i): f'ch{i}' for i in range(arch.input_channels)},
            'labels': {'background': 0, **{f'label{i + 1}': i + 1 for i in range(arch.num_classes)}},
            'file_ending': '.nrrd', 'multilabel': True}
        self.plans_manager = SimpleNamespace(plans=plans or {}, transpose_forward=[0, 1, 2], transpose_backward=[0, 1, 2])
        self.configuration_manager = SimpleNamespace(
            patch_size=list(patch_size), spacing=list(spacing), preprocessor_class=DefaultPreprocessor,
            normalization_schemes=['ZScoreNormalization'] * arch.input_channels,
            use_mask_for_norm=[False] * arch.input_channels, configuration=configuration)
        self._create_engines()

    def initialize_from_trained_model_folder(self, model_training_output_dir: str, use_folds, checkpoint_name: str = 'checkpoint_final.pth'):
        """Reads ``dataset.json`` / ``plans.json`` / ``fold_X/<checkpoint_name>`` exactly where upstream does."""
        from . import weights as W
        with open(os.path.join(model_training_output_dir, 'dataset.json')) as f:
            dataset_json = json.load(f)
        with open(os.path.join(model_training_output_dir, 'plans.json')) as f:
            plans = json.load(f)
        if use_folds is None or use_folds == 'auto':
            use_folds = sorted(int(m.group(1)) for m in (re.match(r'fold_(\d+)$', d) for d in os.listdir(model_training_output_dir)) if m)
        if isinstance(use_folds, (str, int)):
            use_folds = [use_folds]
        blobs, mirror, configuration = [], None, None
        n_in = len(dataset_json['channel_names'])
        labels = dataset_json['labels']
        multilabel = bool(dataset_json.get('multilabel', dataset_json.get('multiclass', False)))
        n_fg = len([k for k, v in labels.items() if k != 'background' and v != 0])
        n_heads = n_fg if multilabel else len(labels)
        for i, f in enumerate(use_folds):
            f = int(f) if f != 'all' else f
            sd, mirror_axes, init_args = W.load_checkpoint(os.path.join(model_training_output_dir, f'fold_{f}', checkpoint_name))
            if i == 0:
                configuration = init_args.get('configuration', '2d')
                mirror = mirror_axes
                arch = UNetArch.from_plans(plans, configuration, n_in, n_heads)
            blobs.append(W.pack_blob(arch, sd))
        cfg = plans['configurations'][configuration]
        self.manual_initialization(arch, blobs, cfg['patch_size'], cfg.get('spacing', (1.0, 1.0)), dataset_json, plans,
                                   configuration, mirror)
        self.plans_manager.transpose_forward = plans.get('transpose_forward', [0, 1, 2])
        self.plans_manager.transpose_backward = plans.get('transpose_backward', [0, 1, 2])
        self.configuration_manager.normalization_schemes = cfg.get('normalization_schemes', self.configuration_manager.normalization_schemes)
        self.configuration_manager.use_mask_for_norm = cfg.get('use_mask_for_norm', self.configuration_manager.use_mask_for_norm)

    def _create_engines(self):
        from .engine import Engine            # raises loudly if libts2d_engine.so is missing - no fallback
        for e in self.engines:
            e.close()
        self.engines = [Engine(self.arch, blob, self.device.index) for blob in self.list_of_parameters]
        for e in self.engines:
            e.set_precision(self.precision)
            e.set_tile_dtype(self.tile_dtype)

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []

    # ------------------------------------------------------------------ inference
    def predict_sliding_window_return_logits(self, data: np.ndarray, fold: int = 0) -> np.ndarray:
        """One fold: tiles x mirror variants -> one engine batch -> upstream's fp16 Gaussian aggregation.
        data [C,Z,H,W] float32 -> float16 [K,Z,H,W]."""
        patch = tuple(self.configuration_manager.patch_size)
        data = np.asarray(data, dtype=np.float32)
        if data.ndim != 4:
            raise AssertionError('input_image must be a 4D np.ndarray or torch.Tensor (c, x, y, z)')
        padded, revert = sw.pad_nd_image(data, patch)
        C, Z, H, W = padded.shape
        slicers = sw.tile_slicers((H, W), patch, self.tile_step_size, Z)
        if self.use_mirroring and self.allowed_mirroring_axes and max(self.allowed_mirroring_axes) > 1:
            raise AssertionError('mirror_axes does not match the dimension of the input!')
        # gather (with mirroring), network, mirror-average and fp16 Gaussian aggregation all on the device
        g = sw.compute_gaussian(patch) if self.use_gaussian else None
        K = self.arch.num_classes
        logits = np.empty((K, Z, H, W), dtype=np.float16)
        axes = self.allowed_mirroring_axes if self.use_mirroring else None
        any_inf = False
        for d in range(Z):
            tiles = [(sx, sy) for (dd, sx, sy) in slicers if dd == d]
            # (Z == 1, the 2-D case: logits[:, d] is contiguous and the engine writes into it directly)
            out16, _ = self.engines[fold].predict_tiled(padded[:, d], patch, tiles, axes, g, want_logits=True,
                                                        out_logits=logits[:, d] if Z == 1 else None)
            if Z != 1:
                logits[:, d] = out16
            any_inf = any_inf or self.engines[fold].last_tiled_inf
        if any_inf:
            raise RuntimeError('Encountered inf in predicted array. Aborting... If this problem persists, reduce '
                               'value_scaling_factor in compute_gaussian or increase the dtype of predicted_logits to fp32')
        return logits[(slice(None),) + revert[1:]]

    def predict_segmentation_from_preprocessed_data(self, data):
        """Fast path of the product surface (not part of the reference's duck-typed seam): the multilabel segmentation
        ``sigmoid(float(half logits)) > 0.5`` thresholded ON THE DEVICE by the aggregation kernel (kernels_sw.h: the same predicate as
        export.py's bit-pattern test, verified on all 65 536 half values), so that K uint8 planes travel to the host instead of K float16
        ones and the host never thresholds.  One fold, one z-slice (the 2-D models of ts2d); returns uint8 [K, 1, H, W] in the
        preprocessed geometry, or None when the case needs the logits (fold ensembles average logits first; 3-D stacks)."""
        if hasattr(data, 'detach'):
            data = data.detach().cpu().numpy()
        data = np.asarray(data, dtype=np.float32)
        if len(self.list_of_parameters) != 1 or len(self.engines) != 1 or data.ndim != 4 or data.shape[1] != 1:
            return None
        patch = tuple(self.configuration_manager.patch_size)
        padded, revert = sw.pad_nd_image(data, patch)
        C, Z, H, W = padded.shape
        slicers = sw.tile_slicers((H, W), patch, self.tile_step_size, Z)
        if self.use_mirroring and self.allowed_mirroring_axes and max(self.allowed_mirroring_axes) > 1:
            raise AssertionError('mirror_axes does not match the dimension of the input!')
        g = sw.compute_gaussian(patch) if self.use_gaussian else None
        axes = self.allowed_mirroring_axes if self.use_mirroring else None
        tiles = [(sx, sy) for (dd, sx, sy) in slicers if dd == 0]
        _, seg = self.engines[0].predict_tiled(padded[:, 0], patch, tiles, axes, g, want_logits=False, want_seg=True)
        if self.engines[0].last_tiled_inf:
            raise RuntimeError('Encountered inf in predicted array. Aborting... If this problem persists, reduce '
                               'value_scaling_factor in compute_gaussian or increase the dtype of predicted_logits to fp32')
        return seg[:, None][(slice(None),) + revert[1:]]

    def predict_logits_from_preprocessed_data(self, data):
        """Fold ensemble (upstream: sum over ``list_of_parameters`` then ``/= n``).  Accepts numpy or torch [C,1,H,W];
        returns a torch CPU tensor (float16) when torch is importable so that the caller's ``.cpu()`` works."""
        if hasattr(data, 'detach'):
            data = data.detach().cpu().numpy()
        n = max(1, len(self.list_of_parameters))
        pred = None
        for f in range(n):
            p = self.predict_sliding_window_return_logits(data, f)
            pred = p if pred is None else pred + p
        if n > 1:
            pred = pred / np.float16(n)
        try:
            import torch
            return torch.from_numpy(np.ascontiguousarray(pred))
        except ImportError:
            return pred

"""In-process model handle: the replacement of the reference's ``NNUProcessModel`` (``ts2d/core/inference/nnu.py:98-241``)
and its ``ParallelPredictor`` worker pool (``ts2d/core/inference/predictor.py``).  Same surface - ``start / await_startup /
apply / stop``, ``channels``, ``multilabel``, ``revision`` - but no worker process, no Manager queue, no temp NRRD files:
``apply`` runs preprocess -> HIP engine -> export in the caller's process and reports per-stage timestamps like
``PredictTask.timestamps`` (``prediction_worker.py:57-58``).
"""
from __future__ import annotations

import json
import os
import re
import time
from typing import Dict, List, Optional, Union

import numpy as np

from . import nrrd
from .export import export_prediction_from_logits
from .predictor import HIPnnUNetPredictor


def parse_int(v):
    try:
        return int(v)
    except (TypeError, ValueError):
        return v


class HIPModel:
    def __init__(self, config: dict):
        """config keys (reference ``model.json`` + loader additions): ``root``, ``model``, ``revision``, ``folds``, ``param``
        (``nnu.*`` dotted keys, reference wrapper.py:53-71), or ``synthetic`` = dict(arch, blobs, patch_size, dataset_json)."""
        self._config = dict(config)
        self._param = dict(config.get('param', {}))
        self._predictor: Optional[HIPnnUNetPredictor] = None
        self.timestamps: Dict[str, float] = {}
        self.labels: Optional[Dict[int, str]] = None
        self.colors = self._param.get('nnu.result.colors')
        self._dataset_json: Optional[dict] = None
        self.device_threshold = True     # thresholded segmentation straight from the device where the export needs no logits (_apply_one)
        self._discover()

    # ------------------------------------------------------------------ configuration (reference wrapper.py:113-162)
    def _discover(self):
        syn = self._config.get('synthetic')
        if syn is not None:
            self._dataset_json = syn['dataset_json']
            self._data_dir = None
        else:
            root = self._config['root']
            task = next((d for d in sorted(os.listdir(root)) if re.match(r'Dataset\d+_', d)), None)
            if task is None:
                raise RuntimeError(f"no nnU-Net v2 'Dataset###_*' directory found in {root}")
            trainer = '__'.join([self._param.get('nnu.trainer', 'nnUNetTrainer'), self._param.get('nnu.plans', 'nnUNetPlans'),
                                 self._param.get('nnu.configuration', '3d_fullres')])
            self._data_dir = os.path.join(root, task, trainer)
            with open(os.path.join(self._data_dir, 'dataset.json')) as f:
                self._dataset_json = json.load(f)
        lab = self._dataset_json.get('labels', {})
        self.labels = {int(v): k for k, v in lab.items() if k != 'background'}

    @property
    def name(self):
        return self._config.get('model', 'model')

    @property
    def revision(self):
        r = self._config.get('revision', 0)
        return f'r{r:03d}' if isinstance(r, int) else r

    @property
    def folds(self):
        f = self._param.get('nnu.folds', self._config.get('folds'))
        return tuple(f) if f else (0,)

    @property
    def channels(self) -> Dict[int, str]:
        return {int(k): v for k, v in self._dataset_json['channel_names'].items()}

    @property
    def multilabel(self) -> bool:
        return bool(self._dataset_json.get('multilabel', self._dataset_json.get('multiclass', False)))

    # ------------------------------------------------------------------ lifetime (reference nnu.py:118-137)
    def start(self, wait: bool = True):
        p = self._param
        kw = {}
        if p.get('nnu.predict.stepsize') is not None:
            kw['tile_step_size'] = float(p['nnu.predict.stepsize'])
        kw['use_mirroring'] = bool(p.get('nnu.predict.augment', True))        # reference default: True (wrapper.py:65)
        kw['verbose'] = bool(p.get('nnu.verbose', False))
        kw['device'] = self._config.get('device')
        kw['precision'] = p.get('hip.precision', self._config.get('precision', 'split'))     # engine arithmetic mode (include/ts2d_engine.h)
        pred = self._make_predictor(kw)
        syn = self._config.get('synthetic')
        if syn is not None:
            pred.manual_initialization(syn['arch'], syn['blobs'], syn['patch_size'], syn.get('spacing', (1.5, 1.5)),
                                       syn['dataset_json'], inference_allowed_mirroring_axes=syn.get('mirror_axes', (0, 1)))
        else:
            ck = p.get('nnu.predict.checkpoint', 'final')
            pred.initialize_from_trained_model_folder(self._data_dir, self.folds, f'checkpoint_{ck}.pth')
        self._predictor = pred
        if wait:
            self.await_startup()

    def _make_predictor(self, kw: dict):
        """The predictor object behind this model: always the HIP one (the CPU surface tests subclass the model, tests/surface_util.py)."""
        return HIPnnUNetPredictor(**kw)

    def await_startup(self):
        """Warm-up on a zero patch (reference prediction_worker.py:74-96): allocates the workspace, loads the kernels."""
        p = self._predictor
        ps = tuple(p.configuration_manager.patch_size)
        p.predict_logits_from_preprocessed_data(np.zeros((len(self.channels), 1) + ps, np.float32))

    def stop(self):
        if self._predictor is not None:
            self._predictor.close()
            self._predictor = None

    # ------------------------------------------------------------------ apply (reference nnu.py:169-241)
    def apply(self, inputs: Union[str, nrrd.Image, List, Dict], result_dir: Optional[str] = None, override: bool = True):
        if self._predictor is None:
            raise RuntimeError("model is not started")
        single = isinstance(inputs, (str, nrrd.Image))
        if single:
            inputs = [inputs]
        if isinstance(inputs, (list, tuple)):
            inputs = {f'image{i + 1}': img for i, img in enumerate(inputs)}
        results = {}
        for name, img in inputs.items():
            try:
                results[name] = self._apply_one(name, img, result_dir, override)
            except Exception as ex:
                raise RuntimeError(f"Prediction failed for: {name}: {ex}") from ex
        return next(iter(results.values())) if single else results

    @staticmethod
    def _preprocess_key(p, props) -> str:
        """Everything DefaultPreprocessor.run_case_npy reads besides the image itself: two sub-models with the same key preprocess identically."""
        cm, pm = p.configuration_manager, p.plans_manager
        dz = props.get('device_zscore')
        return repr((list(getattr(pm, 'transpose_forward', [0, 1, 2])), list(cm.spacing), list(getattr(cm, 'normalization_schemes', None) or []),
                     list(getattr(cm, 'use_mask_for_norm', None) or []), sorted((p.dataset_json.get('channel_names') or {}).items()),
                     (getattr(pm, 'plans', None) or {}).get('foreground_intensity_properties_per_channel', {}),
                     None if dz is None else tuple(dz.get('order', ()))))

    def _apply_one(self, name, img, result_dir, override):
        """The reference worker's four stages (``prediction_worker.py:177-242``), each failing under its own name -
        ``"<Stage> failed for <name>: <cause>"`` - so that a HIP error (``ts2d_last_error``) tells which stage raised it."""
        p = self._predictor
        ts = self.timestamps = {'start': time.time()}
        ofile = None
        try:
            if result_dir is not None:
                os.makedirs(result_dir, exist_ok=True)
                ofile = os.path.join(result_dir, name)
                if not override and os.path.exists(ofile + '.nrrd'):
                    return ofile + '.nrrd'
        except Exception as ex:
            raise RuntimeError(f"Could not create output directory: {ex}") from ex
        try:
            ref = nrrd.read(img) if isinstance(img, str) else img
            from .preprocess import image_to_array
            data, props = image_to_array(ref)
            if getattr(ref, 'device_zscore', None) is not None:
                props['device_zscore'] = ref.device_zscore      # z-score done on the device behind the projection (image.py)
            pre = p.configuration_manager.preprocessor_class(verbose=p.verbose)
            shared = getattr(ref, 'preprocess_cache', None)      # set by TS2D.predict: the sub-models of one case mostly share channels and plan
            if shared is None:
                data, _, props = pre.run_case_npy(data, None, props, p.plans_manager, p.configuration_manager, p.dataset_json)
            else:
                key = self._preprocess_key(p, props)
                with shared['lock']:                             # (the first sub-model computes, its siblings wait for the result instead of repeating it)
                    hit = shared['items'].get(key)
                    if hit is None:
                        d2, _, p2 = pre.run_case_npy(data, None, props, p.plans_manager, p.configuration_manager, p.dataset_json)
                        d2.setflags(write=False)
                        hit = shared['items'][key] = (d2, p2)
                data, props = hit[0], dict(hit[1])
            ts['preprocessed'] = time.time()
        except Exception as ex:
            raise RuntimeError(f"Preprocessing failed for {name}: {ex}") from ex
        try:
            logits = None
            # product fast path: a multilabel 2-D case whose export does not resample gets its segmentation thresholded on the device
            # (K uint8 planes to the host instead of K float16 ones; the predicate is the export step's, bit for bit) - the reference's seam
            # (predict_logits_from_preprocessed_data + export_prediction_from_logits) stays as it is and serves every other case
            if self.device_threshold and bool(p.dataset_json.get('multilabel', p.dataset_json.get('multiclass', False))) \
                    and hasattr(p, 'predict_segmentation_from_preprocessed_data'):
                from .export import needs_logits
                if not needs_logits(props, np.asarray(data).shape[1:]):
                    logits = p.predict_segmentation_from_preprocessed_data(data)
            if logits is None:
                logits = p.predict_logits_from_preprocessed_data(data)
                logits = logits.cpu().numpy() if hasattr(logits, 'cpu') else logits
            ts['predicted'] = time.time()
        except Exception as ex:
            raise RuntimeError(f"Prediction failed for {name}: {ex}") from ex
        try:
            seg = export_prediction_from_logits(logits, props, p.configuration_manager, p.plans_manager, p.dataset_json, ofile,
                                                ref_image=ref, labels=self.labels,
                                                colors=self.colors if isinstance(self.colors, dict) else None)
            ts['exported'] = ts['done'] = time.time()
        except Exception as ex:
            raise RuntimeError(f"Export failed for {name}: {ex}") from ex
        return (ofile + '.nrrd') if result_dir is not None else seg

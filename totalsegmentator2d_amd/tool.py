"""``TS2D`` - the public API surface of the reference (``ts2d/tool.py:19-311``) on top of the MI355X engine.

Same constructor, ``predict(input, collapse, merge) -> TS2D.Result`` and ``Result`` accessors / ``save`` naming
(``<name>.seg.nrrd``, ``<name>-<group>.seg.nrrd``, ``<name>_<channel>.nrrd``; reference tool.py:235-311,
test/test_030_cli.py:46-50).  Differences: models run in-process (no worker pool, no temp files); the sub-models of one case
run CONCURRENTLY (one host thread per sub-model, each engine on its own HIP stream: a case is 2 tiles x 4 mirror passes = 8 slices
per sub-model, which leaves most of the GPU idle when the five are driven one after the other as the reference does,
tool.py:110-112; ``concurrent_models=False`` restores that order - the results are identical); PNG visualisation is not
implemented (``content='visual'`` is skipped with a warning; SURVEY.md marks rendering out of scope).
"""
from __future__ import annotations

import os
import warnings
from typing import Dict, List, Optional, Union

import numpy as np

from . import nrrd
from .image import (cast, combine_segmentations, compose, get_actual_dimension, project, reduce_dimensions, reorient_image,
                    restore_dimension, split_channels)
from .model import HIPModel
from .zoo import LocalZoo, decompose_model_key, get_label_colors


def _as_list(v):
    return list(v) if isinstance(v, (list, tuple, set)) else [v]


class TS2D:
    def __init__(self, key: str = "ts2d", use_remote: bool = True, fetch_remote: bool = True,
                 models: Optional[Dict[str, HIPModel]] = None, zoo_root: Optional[str] = None, device=None,
                 concurrent_models: bool = True):
        """``models``: pre-built ``{id: HIPModel}`` (synthetic-weight models in tests / bench); otherwise `key` is resolved
        against the local zoo.  ``use_remote`` / ``fetch_remote`` are accepted for signature compatibility (no network)."""
        self.models: Dict[str, HIPModel] = {}
        self.concurrent_models = bool(concurrent_models)
        if models is None:
            self.zoo = LocalZoo(zoo_root)
            ids = self.zoo.resolve(key, unique_model=True)
            if not ids:
                raise RuntimeError(f"No models were resolved for key: {key}")
            models = {}
            for mid in ids:
                try:
                    cfg = self.zoo.load_config(mid, {'server.workers': 1, 'nnu.result.colors': get_label_colors()})
                    cfg['device'] = device
                    models[mid] = HIPModel(cfg)
                except Exception as ex:
                    raise RuntimeError(f"Failed to load model {mid}" + (f" (resolved from {key})" if key != mid else "")) from ex
        for mid, model in models.items():
            try:
                if getattr(model, 'colors', None) is None:      # reference tool.py:29-33: every model gets the packaged label colours
                    model.colors = get_label_colors()
                model.start(wait=False)
                if not model.multilabel:
                    warnings.warn(f"The loaded model {mid} is not configured for multilabel inference.")
                self.models[mid] = model
            except Exception as ex:
                self.close()
                raise RuntimeError(f"Failed to load model {mid}") from ex
        for model in self.models.values():
            model.await_startup()
        # the projections run on the GPU when every model is backed by HIP engines (always, on the product path)
        self._gpu_projection = all(bool(getattr(getattr(m, '_predictor', None), 'engines', None)) for m in self.models.values())

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.close()

    def close(self):
        pool, self._pool = getattr(self, '_pool', None), None
        if pool is not None:
            pool.shutdown(wait=True)
        for model in self.models.values():
            model.stop()
        self.models = {}

    def _executor(self, n: int):
        """One worker thread per sub-model for the life of this object (not five new threads per case)."""
        pool = getattr(self, '_pool', None)
        if pool is None or pool._max_workers < n:
            from concurrent.futures import ThreadPoolExecutor
            if pool is not None:
                pool.shutdown(wait=True)
            pool = self._pool = ThreadPoolExecutor(max_workers=n, thread_name_prefix='ts2d-submodel')
        return pool

    def predict(self, input: Union[nrrd.Image, str], collapse: bool = False, merge: bool = True) -> "TS2D.Result":
        if isinstance(input, str):
            input = nrrd.read(input)
        if not isinstance(input, nrrd.Image):
            raise RuntimeError(f"input must be a string path or an image, found: {type(input).__name__}")
        result: dict = {}
        cache: dict = {}
        order = sorted(self.models)
        # input side per sub-model (projections are computed once and cached), then the networks - concurrently: the C-ABI's handles
        # are independent (include/ts2d_engine.h: one caller thread per engine), every engine runs on its own stream, and ctypes
        # releases the interpreter lock for the duration of a call
        prepared = {mid: self._prepare_model_input(mid, input, cache) for mid in order}
        if self.concurrent_models and len(order) > 1:
            pool = self._executor(len(order))
            futs = {mid: pool.submit(self._apply_model, mid, prepared[mid], collapse) for mid in order}
            done = {}
            try:
                for mid in order:
                    done[mid] = futs[mid].result()                       # (the first failure is raised, in sub-model order)
            except BaseException:
                for f in futs.values():                                  # sub-models that have not started yet are not run for a failed case
                    f.cancel()
                raise
        else:
            done = {mid: self._apply_model(mid, prepared[mid], collapse) for mid in order}
        for mid in order:
            result.setdefault('models', {})[mid] = done[mid]
        if merge:
            segs = [r['segmentation'] for _, r in sorted(result['models'].items())]
            result['segmentation'] = segs[0] if len(segs) == 1 else combine_segmentations(segs)
        result['input'] = input
        if cache.get('projections'):
            result['projections'] = cache['projections']
        return TS2D.Result(result)

    def _predict_model(self, mid: str, input: nrrd.Image, collapse: bool, cache: dict) -> dict:
        return self._apply_model(mid, self._prepare_model_input(mid, input, cache), collapse)

    def _prepare_model_input(self, mid: str, input: nrrd.Image, cache: dict):
        """Reference tool.py:145-171: the 2-D multi-channel input of one sub-model (projections cached across sub-models)."""
        model = self.models[mid]
        channels = sorted(model.channels.items())
        projections = cache.setdefault('projections', {})
        if get_actual_dimension(input) > 2:
            need = [n for _, n in channels if n not in projections]
            if need and self._gpu_projection and input.components == 1 and input.array.dtype.name in ('int16', 'uint8', 'float32', 'uint16', 'int32') \
                    and all(n.lower() in ('max', 'mip', 'mean', 'avg') for n in need):
                # product path: both projections in one pass over the volume on the GPU, reorientation folded into the strides
                from .image import project_coronal_gpu
                pr = project_coronal_gpu(input, getattr(model._predictor.device, 'index', 0) or 0, zscore=True)
                for n in need:
                    projections[n] = pr['max' if n.lower() in ('max', 'mip') else 'mean']
                cache['device_zscore'] = pr['zscore']      # both projections normalised on the device (preprocess.py uses it)
            else:
                input = reorient_image(input, 'RAI')
            chs = []
            for _, ch_name in channels:                      # channel NAME = projection mode (reference tool.py:156-158)
                if ch_name not in projections:
                    projections[ch_name] = cast(project(input, mode=ch_name, axis='coronal'), np.float32)
                chs.append(projections[ch_name])
            input = compose(chs) if len(chs) > 1 else chs[0]
        else:
            if len(channels) != input.components:
                raise RuntimeError(f"The number of channels in the input image does not match the models channel definition "
                                   f"({len(channels)} vs {input.components}).")
            projections.update((f"ch{i}", ch) for i, ch in enumerate(split_channels(input)))
        native_2d = input.dimension < 3
        input2d = input if native_2d else reduce_dimensions(input)
        dz = cache.get('device_zscore')
        if dz is not None and not native_2d and all(n.lower() in ('max', 'mip', 'mean', 'avg') for _, n in channels):
            # channel order of THIS model over the (max, mean) planes the device normalised
            input2d.device_zscore = dict(dz, order=tuple(0 if n.lower() in ('max', 'mip') else 1 for _, n in channels))
        # one preprocessing per distinct (channels, plan) of the case, shared by the sub-models (they run on threads: model.py takes the lock)
        import threading
        input2d.preprocess_cache = cache.setdefault('preprocessed', {'lock': threading.Lock(), 'items': {}})
        return input, input2d, native_2d

    def _apply_model(self, mid: str, prepared, collapse: bool) -> dict:
        """Reference tool.py:172-174: ``model.apply`` + restoring the 3-D geometry."""
        model = self.models[mid]
        input, input2d, native_2d = prepared
        res = {'id': mid, 'revision': model.revision}
        res['model'], res['group'] = decompose_model_key(mid)
        seg = model.apply(input2d)
        if not (collapse or native_2d):
            seg = restore_dimension(seg, input)
        res['input'] = input2d if collapse else input
        res['segmentation'] = seg
        res['timestamps'] = dict(model.timestamps)
        return res

    class Result:
        def __init__(self, data: dict):
            self.data = data

        @property
        def models(self) -> List[str]:
            return sorted(self.data.get('models', {}).keys())

        def get_input(self, model: Optional[str] = None):
            return self.data.get('models', {}).get(model, {}).get('input') if model is not None else self.data.get('input')

        def get_segmentation(self, model: Optional[str] = None):
            return self.data.get('models', {}).get(model, {}).get('segmentation') if model is not None else self.data.get('segmentation')

        def get_projection(self, channel: Optional[str] = None):
            pr = self.data.get('projections', {})
            return pr.get(channel) if channel is not None else pr

        def save(self, dest: str, name: str = 'result', ext: str = 'nrrd', models='final', targets='all', content: str = 'all',
                 naming: str = 'group'):
            assert ext.lower() != 'png', "PNG is not a valid export format for the 'file' content type."
            assert naming in {'group', 'model'} and content in {'file', 'visual', 'all'}
            if content in ('visual', 'all'):
                warnings.warn("PNG visualisation is not implemented in the MI355X build; only files are written.")
            mset = {str(t).strip().lower() for t in _as_list(models)}
            keys: set = set()
            if 'all' in mset:
                keys |= set(self.models) | {None}
            if 'final' in mset:
                keys |= {None}
            keys |= {m for m in self.models if m.lower() in mset}
            tset = {str(t).strip().lower() for t in _as_list(targets)}

            def fname(base, key):
                if key is not None and naming == 'group':
                    return f"{base}-{decompose_model_key(key)[1]}"
                return base if key is None else f"{base}-{key}"

            os.makedirs(dest, exist_ok=True)
            written = []
            if {'all', 'input'} & tset:
                for key in keys:
                    img = self.get_input(key)
                    if img is not None:
                        written.append(os.path.join(dest, f"{fname(name, key)}.{ext}"))
                        nrrd.write(img, written[-1], True)
            if {'all', 'segmentation'} & tset:
                for key in keys:
                    img = self.get_segmentation(key)
                    if img is not None:
                        written.append(os.path.join(dest, f"{fname(name, key)}.seg.{ext}"))
                        nrrd.write(img, written[-1], True)
            if {'all', 'projection'} & tset:
                for channel, img in self.get_projection().items():
                    written.append(os.path.join(dest, f"{name}_{channel}.{ext}"))
                    nrrd.write(img, written[-1], True)
            return written

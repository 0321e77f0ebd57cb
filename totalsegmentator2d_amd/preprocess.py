"""Input side of the hot path: nnU-Net's ``DefaultPreprocessor.run_case`` for the 2-D multi-channel inputs ts2d feeds.

Reference call site: ``ts2d/core/inference/prediction_worker.py:194-199`` (``preprocessor.run_case(task.filenames, None,
plans_manager, configuration_manager, dataset_json)`` -> ``(data[C,1,H,W] float32, None, properties)``).  The
algorithm is third-party (nnunetv2ml==2.6.2): read -> float32 -> transpose_forward ->
crop_to_nonzero -> per-channel normalisation (channel names ``mean`` / ``max`` are not ``ct`` => ZScoreNormalization)
-> resample to the plan spacing when the rounded target shape differs.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import nrrd


def sitk_stuff(img, files=()) -> dict:
    """``properties['sitk_stuff']`` as upstream's ``SimpleITKIO.read_images`` fills it [UPSTREAM-RECALL]: the geometry of the
    FIRST input image in SimpleITK order (x, y[, z]; direction flattened row-major).  Upstream's ``SimpleITKIO.write_seg``
    reads exactly these three keys back when the reference's unchanged exporter (``export_prediction_from_logits``,
    ``ts2d/core/inference/prediction_worker.py:215-221``) writes the result of THIS preprocessor's properties (INTEGRATION.md
    section 1)."""
    return {'spacing': tuple(float(v) for v in img.spacing), 'origin': tuple(float(v) for v in img.origin),
            'direction': tuple(float(v) for v in img.direction), 'files': list(files)}


def image_to_array(img) -> Tuple[np.ndarray, dict]:
    """One in-memory image -> (``[c, z, y, x]`` float32, properties); 2-D images get a unit z axis and nnU-Net's
    ``999`` pseudo-spacing."""
    a = np.asarray(img.array)
    if img.dimension == 2:
        a = a[None] if img.components == 1 else np.moveaxis(a, -1, 0)
        a = a[:, None]
        sp = (999.0, float(img.spacing[1]), float(img.spacing[0]))
    elif img.dimension == 3:
        a = a[None] if img.components == 1 else np.moveaxis(a, -1, 0)
        sp = (float(img.spacing[2]), float(img.spacing[1]), float(img.spacing[0]))
    else:
        raise RuntimeError(f"unsupported image dimension {img.dimension}")
    return a.astype(np.float32), {'spacing': sp, 'sitk_stuff': sitk_stuff(img)}


def read_images(files: Sequence[str]) -> Tuple[np.ndarray, dict]:
    """The fork's reader for one multi-component 2-D file (reference ``ts2d/tool.py:160,170-172`` hands the model ONE
    vector image): ``[y, x, c]`` -> ``[c, 1, y, x]``; spacing reported nnU-Net style ``(999, sy, sx)`` for 2-D."""
    arrs, spacing, stuff = [], None, None
    for fp in files:
        img = nrrd.read(fp)
        a = np.asarray(img.array)
        if img.dimension == 2:
            a = a[None] if img.components == 1 else np.moveaxis(a, -1, 0)      # [c, y, x]
            a = a[:, None]                                                      # [c, 1, y, x]
            sp = (999.0, float(img.spacing[1]), float(img.spacing[0]))
        elif img.dimension == 3:
            a = a[None] if img.components == 1 else np.moveaxis(a, -1, 0)      # [c, z, y, x]
            sp = (float(img.spacing[2]), float(img.spacing[1]), float(img.spacing[0]))
        else:
            raise RuntimeError(f"unsupported image dimension {img.dimension} in {fp}")
        arrs.append(a)
        if stuff is None:                       # upstream keeps the first image's geometry (and checks the others against it)
            spacing, stuff = sp, sitk_stuff(img, files)
    data = np.concatenate(arrs, 0).astype(np.float32)
    return data, {'spacing': spacing, 'sitk_stuff': stuff}


def crop_to_nonzero(data: np.ndarray):
    """``crop_to_nonzero``: bounding box of voxels that are non-zero in ANY channel."""
    nz = np.any(data != 0, axis=0)
    if not nz.any():
        bbox = [[0, s] for s in data.shape[1:]]
    else:
        bbox = []
        for ax in range(nz.ndim):
            other = tuple(i for i in range(nz.ndim) if i != ax)
            idx = np.where(nz.any(axis=other))[0]
            bbox.append([int(idx[0]), int(idx[-1]) + 1])
    sl = (slice(None),) + tuple(slice(b[0], b[1]) for b in bbox)
    return data[sl], bbox


def zscore(img: np.ndarray) -> np.ndarray:
    """``ZScoreNormalization.run`` (no mask): float32, ``(x - mean) / max(std, 1e-8)``."""
    img = img.astype(np.float32, copy=True)
    mean, std = img.mean(), img.std()
    img -= mean
    img /= max(std, 1e-8)
    return img


def _device_zscore_applies(dz, data, bbox, tf, schemes, use_mask) -> bool:
    """The device result is nnU-Net's result only if nothing sits between projection and normalisation: identity transpose,
    crop-to-nonzero = whole image (checked on both sides: the device's non-zero box and the host's bbox), plain z-score on
    every channel, one channel per projection."""
    nz, nx = dz['shape']
    return (list(tf) == [0, 1, 2] and data.shape[1:] == (1, nz, nx) and len(dz['order']) == data.shape[0]
            and tuple(dz['box']) == (0, nz - 1, 0, nx - 1) and [list(b) for b in bbox] == [[0, 1], [0, nz], [0, nx]]
            and all(s == 'ZScoreNormalization' for s in schemes[:data.shape[0]]) and not any(use_mask[:data.shape[0]]))


class DefaultPreprocessor:
    def __init__(self, verbose: bool = True):
        self.verbose = verbose

    def run_case_npy(self, data: np.ndarray, seg, properties: dict, plans_manager, configuration_manager, dataset_json):
        data = data.astype(np.float32)
        tf = list(getattr(plans_manager, 'transpose_forward', [0, 1, 2]))
        data = data.transpose([0] + [i + 1 for i in tf])
        original_spacing = [properties['spacing'][i] for i in tf]
        properties['shape_before_cropping'] = data.shape[1:]
        data, bbox = crop_to_nonzero(data)
        properties['bbox_used_for_cropping'] = bbox
        properties['shape_after_cropping_and_before_resampling'] = data.shape[1:]
        target_spacing = list(configuration_manager.spacing)
        if len(target_spacing) < len(data.shape[1:]):
            target_spacing = [original_spacing[0]] + target_spacing
        new_shape = [int(round(i / j * k)) for i, j, k in zip(original_spacing, target_spacing, data.shape[1:])]
        schemes = getattr(configuration_manager, 'normalization_schemes', None) or ['ZScoreNormalization'] * data.shape[0]
        use_mask = getattr(configuration_manager, 'use_mask_for_norm', None) or [False] * data.shape[0]
        dz = properties.pop('device_zscore', None)
        if dz is not None and not _device_zscore_applies(dz, data, bbox, tf, schemes, use_mask):
            dz = None
        for c in range(data.shape[0]):
            if dz is not None:          # normalised on the device behind the projection (ts2d_project_coronal_zscore): no host pass
                data[c, 0] = dz['norm'][dz['order'][c]]
                continue
            if schemes[c] == 'ZScoreNormalization' and c < len(use_mask) and use_mask[c]:
                # upstream then takes mean/std inside the nonzero mask only and leaves the outside at 0 - refuse rather than
                # normalise differently in silence
                raise NotImplementedError(f"use_mask_for_norm is set for channel {c}: masked ZScoreNormalization is not implemented")
            if schemes[c] not in ('ZScoreNormalization', 'NoNormalization'):
                raise NotImplementedError(f"normalization scheme {schemes[c]} is not implemented")
            if schemes[c] == 'ZScoreNormalization':
                data[c] = zscore(data[c])
        if list(new_shape) != list(data.shape[1:]):
            raise NotImplementedError(f"resampling {list(data.shape[1:])} -> {new_shape} (spacing {original_spacing} -> "
                                      f"{target_spacing}) is not implemented")
        return data, None, properties

    def run_case(self, image_files: List[str], seg_file: Optional[str], plans_manager, configuration_manager, dataset_json):
        if isinstance(image_files, str):
            image_files = [image_files]
        data, props = read_images(image_files)
        return self.run_case_npy(data, None, props, plans_manager, configuration_manager, dataset_json)

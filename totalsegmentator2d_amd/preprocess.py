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


def crop_to_nonzero(data: np.ndarray, return_mask: bool = False):
    """``crop_to_nonzero``: bounding box of voxels that are non-zero in ANY channel.  ``return_mask``: also the non-zero mask inside
    the box - upstream writes it into the segmentation (``seg = where(nonzero_mask, 0, -1)``) and the masked normalisers use
    ``seg >= 0``.  Upstream fills the holes of the mask (``create_nonzero_mask``: ``binary_fill_holes`` on the [Z, H, W] mask, default
    3-D structure) - a no-op for the [C, 1, H, W] inputs of the 2-D path (with Z = 1 every voxel lies on the array's border, so no
    background region is enclosed), applied here for Z > 1 so that a volume input is masked like upstream masks it [UPSTREAM-RECALL]."""
    nz = np.any(data != 0, axis=0)
    if nz.ndim == 3 and nz.shape[0] > 1:
        from scipy.ndimage import binary_fill_holes
        nz = binary_fill_holes(nz)
    if not nz.any():
        bbox = [[0, s] for s in data.shape[1:]]
    else:
        bbox = []
        for ax in range(nz.ndim):
            other = tuple(i for i in range(nz.ndim) if i != ax)
            idx = np.where(nz.any(axis=other))[0]
            bbox.append([int(idx[0]), int(idx[-1]) + 1])
    sl = (slice(None),) + tuple(slice(b[0], b[1]) for b in bbox)
    if return_mask:
        return data[sl], bbox, nz[sl[1:]]
    return data[sl], bbox


def zscore(img: np.ndarray) -> np.ndarray:
    """``ZScoreNormalization.run`` (no mask): float32, ``(x - mean) / max(std, 1e-8)``."""
    img = img.astype(np.float32, copy=True)
    mean, std = img.mean(), img.std()
    img -= mean
    img /= max(std, 1e-8)
    return img


def normalize_channel(img: np.ndarray, scheme: str, use_mask: bool, mask: Optional[np.ndarray], props: Optional[dict]) -> np.ndarray:
    """nnU-Net ``default_normalization_schemes`` [UPSTREAM-RECALL nnunetv2 2.6], float32 like upstream's ``target_dtype``:
    ZScoreNormalization (optionally inside the non-zero mask only: outside stays 0), CTNormalization (clip to the dataset's
    0.5 / 99.5 percentiles, then the dataset's mean / std: ``plans['foreground_intensity_properties_per_channel'][str(c)]``),
    NoNormalization, RescaleTo01Normalization, RGBTo01Normalization."""
    img = img.astype(np.float32, copy=True)
    if scheme == 'ZScoreNormalization':
        if use_mask:
            if mask is None:
                raise RuntimeError("masked ZScoreNormalization needs the non-zero mask")
            m = mask.astype(bool)
            mean, std = img[m].mean(), img[m].std()
            img[m] = (img[m] - mean) / max(std, 1e-8)
            return img
        return zscore(img)
    if scheme == 'CTNormalization':
        if not props:
            raise RuntimeError("CTNormalization needs plans['foreground_intensity_properties_per_channel']")
        lo, hi = props['percentile_00_5'], props['percentile_99_5']
        np.clip(img, lo, hi, out=img)
        img -= np.float32(props['mean'])
        img /= np.float32(max(props['std'], 1e-8))
        return img
    if scheme == 'NoNormalization':
        return img
    if scheme == 'RescaleTo01Normalization':
        img -= img.min()
        img /= np.clip(img.max(), a_min=1e-8, a_max=None)
        return img
    if scheme == 'RGBTo01Normalization':
        if img.min() < 0 or img.max() > 255:
            raise RuntimeError("RGB images are uint 8, for whatever reason I found pixel values outside [0, 255]")
        return img / np.float32(255.0)
    raise NotImplementedError(f"normalization scheme {scheme} is not implemented")


def resize_like_skimage(img2d: np.ndarray, new_shape, order: int) -> np.ndarray:
    """``skimage.transform.resize(img, new_shape, order, mode='edge', anti_aliasing=False)`` of scikit-image >= 0.19 (nnU-Net's
    resampling primitive) restated with scipy, which is what skimage calls itself on this path [UPSTREAM-RECALL]:
    ``ndi.zoom(img, out / in, order, mode='nearest', grid_mode=True)`` followed by a clip to the input's value range
    (``clip=True``).  skimage is not installed here."""
    from scipy import ndimage as ndi
    img2d = np.asarray(img2d)
    if tuple(img2d.shape) == tuple(new_shape):
        return img2d
    zoom = [n / o for n, o in zip(new_shape, img2d.shape)]
    out = ndi.zoom(img2d, zoom, order=order, mode='nearest', grid_mode=True)
    if order > 0 and out.size:
        np.clip(out, img2d.min(), img2d.max(), out=out)
    return out.astype(img2d.dtype, copy=False)


def resample_data_to_shape(data: np.ndarray, new_shape, order: int = 3) -> np.ndarray:
    """``resample_data_or_seg_to_shape(data, new_shape, current_spacing, new_spacing, is_seg=False, order=3, order_z=0)`` for the
    2-D configurations ts2d uses ([C, 1, H, W] with the 999 pseudo-spacing): upstream finds the 999 axis anisotropic and
    resamples every (channel, slice) in-plane with ``resize``; the slice count does not change [UPSTREAM-RECALL]."""
    new_shape = tuple(int(v) for v in new_shape)
    if tuple(data.shape[1:]) == new_shape:
        return data
    if data.shape[1] != new_shape[0]:
        raise NotImplementedError(f"resampling along the slice axis ({data.shape[1]} -> {new_shape[0]} slices) is not implemented (2-D configurations only)")
    out = np.empty((data.shape[0],) + new_shape, dtype=data.dtype)
    for c in range(data.shape[0]):
        for z in range(data.shape[1]):
            out[c, z] = resize_like_skimage(data[c, z], new_shape[1:], order)
    return out


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
        data, bbox, nzmask = crop_to_nonzero(data, return_mask=True)
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
        fip = (getattr(plans_manager, 'plans', None) or {}).get('foreground_intensity_properties_per_channel', {})
        for c in range(data.shape[0]):
            if dz is not None:          # normalised on the device behind the projection (ts2d_project_coronal_zscore): no host pass
                data[c, 0] = dz['norm'][dz['order'][c]]
                continue
            data[c] = normalize_channel(data[c], schemes[c], bool(c < len(use_mask) and use_mask[c]), nzmask, fip.get(str(c)))
        if list(new_shape) != list(data.shape[1:]):       # the plan's spacing differs from the image's: resample (order 3), AFTER normalising
            data = resample_data_to_shape(data, new_shape, order=3)
        return data, None, properties

    def run_case(self, image_files: List[str], seg_file: Optional[str], plans_manager, configuration_manager, dataset_json):
        if isinstance(image_files, str):
            image_files = [image_files]
        data, props = read_images(image_files)
        return self.run_case_npy(data, None, props, plans_manager, configuration_manager, dataset_json)

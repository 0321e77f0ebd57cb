"""Output side of the hot path: logits -> segmentation image.

Reference: ``export_prediction_from_logits(prediction, data_properties, configuration_manager, plans_manager, dataset_json,
ofile_truncated, save_probabilities)`` called at ``ts2d/core/inference/prediction_worker.py:215-221`` (third-party
nnunetv2ml fork: multilabel => ``sigmoid(logits.float()) > 0.5`` per channel), followed by the metadata restamp
``set_annotation_meta(img, labels, colors)`` at ``prediction_worker.py:226-240``.  Here both happen in memory; the
file is written once.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from . import nrrd
from .image import set_annotation_meta

SIGMOID_HALF_THRESHOLD = np.float32(1.5 * 2.0 ** -24)     # sigmoid(float32 x) > 0.5  <=>  x > 1.5 * 2^-24 (tests/test_oracle.py)


def needs_logits(properties: dict, shape_khw) -> bool:
    """Does the export resample the prediction back to another shape (then it needs logits, not a thresholded segmentation)?"""
    return tuple(properties.get('shape_after_cropping_and_before_resampling', tuple(shape_khw))) != tuple(shape_khw)


def convert_predicted_logits_to_segmentation_with_correct_shape(logits, properties: dict, multilabel: bool = True,
                                                                transpose_backward=(0, 1, 2)) -> np.ndarray:
    """[K, Z, H, W] logits (any float dtype) -> uint8 segmentation in the ORIGINAL (pre-crop) array shape:
    multilabel: [K, Z0, H0, W0] of {0,1}; otherwise a label map [Z0, H0, W0] (argmax).  Upstream first resamples the logits back to
    ``properties['shape_after_cropping_and_before_resampling']`` (``resampling_fn_probabilities``: order 1, per slice for the 2-D
    configurations [UPSTREAM-RECALL]) - a no-op when the plan's spacing is the image's."""
    lg = np.asarray(logits)
    tgt = tuple(properties.get('shape_after_cropping_and_before_resampling', lg.shape[1:]))
    if tuple(lg.shape[1:]) != tgt:
        from .preprocess import resample_data_to_shape
        lg = resample_data_to_shape(lg.astype(np.float32), tgt, order=1)
    shape0 = tuple(properties['shape_before_cropping'])
    bbox = properties['bbox_used_for_cropping']
    sl = tuple(slice(b[0], b[1]) for b in bbox)
    if multilabel:
        if lg.dtype == np.uint8:
            seg = lg          # already thresholded on the device (HIPnnUNetPredictor.predict_segmentation_from_preprocessed_data): same predicate
        elif lg.dtype == np.float16:
            # float32(x) > 1.5 * 2^-24 on the fp16 bit pattern: positive, at least the SECOND subnormal (the first, 2^-24, is
            # below the threshold), +inf included, NaN excluded - the same predicate without a float32 copy of the array
            v = np.ascontiguousarray(lg).view(np.uint16)
            seg = ((v >= np.uint16(2)) & (v <= np.uint16(0x7C00))).view(np.uint8)
        else:
            seg = (lg.astype(np.float32) > SIGMOID_HALF_THRESHOLD).astype(np.uint8)
        out = np.zeros((lg.shape[0],) + shape0, dtype=np.uint8)
        out[(slice(None),) + sl] = seg
        return out.transpose([0] + [i + 1 for i in transpose_backward])
    seg = lg.astype(np.float32).argmax(0).astype(np.uint8)
    out = np.zeros(shape0, dtype=np.uint8)
    out[sl] = seg
    return out.transpose(list(transpose_backward))


def segmentation_to_image(seg: np.ndarray, ref: nrrd.Image, multilabel: bool, labels: Optional[Dict[int, str]] = None,
                          colors: Optional[dict] = None) -> nrrd.Image:
    """uint8 array from :func:`convert_...` -> image with the geometry of the (2-D) input image `ref` and Slicer metadata."""
    if multilabel:
        arr = np.moveaxis(seg, 0, -1)                  # [Z, H, W, K]
        if ref.dimension == 2:
            arr = arr[0]                               # [H, W, K]
        # (the interleaved [.., K] VIEW of the plane-major array: scattering K x H x W bytes at stride K costs 3 ms per sub-model of a
        #  644 x 337 case; consumers index it like sitk's vector image, nrrd.write serialises the logical order)
        img = nrrd.Image(arr, ref.spacing, ref.origin, ref.direction, seg.shape[0], {}, ref.space)
    else:
        arr = seg[0] if ref.dimension == 2 else seg
        img = nrrd.Image(np.ascontiguousarray(arr), ref.spacing, ref.origin, ref.direction, 1, {}, ref.space)
    if labels or colors:
        set_annotation_meta(img, {int(k): v for k, v in (labels or {}).items()}, colors)
    return img


def export_prediction_from_logits(logits, properties: dict, configuration_manager, plans_manager, dataset_json: dict,
                                  ofile_truncated: str, save_probabilities: bool = False, ref_image: Optional[nrrd.Image] = None,
                                  labels: Optional[Dict[int, str]] = None, colors: Optional[dict] = None) -> nrrd.Image:
    multilabel = bool(dataset_json.get('multilabel', dataset_json.get('multiclass', False)))
    tb = getattr(plans_manager, 'transpose_backward', [0, 1, 2])
    seg = convert_predicted_logits_to_segmentation_with_correct_shape(logits, properties, multilabel, tb)
    if ref_image is None:
        ref_image = nrrd.read(properties['sitk_stuff']['files'][0])
    img = segmentation_to_image(seg, ref_image, multilabel, labels, colors)
    if ofile_truncated:
        nrrd.write(img, ofile_truncated + dataset_json.get('file_ending', '.nrrd'), True)
    return img

"""ORACLE (test infrastructure, never shipped or timed as the product): numpy-float64 restatement of the INPUT side of the
hot path - the coronal projections and the per-channel z-score that sit between ``TS2D.predict`` and the network.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Rows restated, reference file:line followed:
  ``ts2d/tool.py:152-160``            reorient to RAI, one projection per channel NAME, ``sitk.Compose``
  ``ts2d/core/util/image.py:32-43``   ``reorient_image`` = ``sitk.DICOMOrient(img, 'RAI')``          -> :func:`dicom_orient_rai`
  ``ts2d/core/util/image.py:78-79``   ``sitk.MeanProjectionImageFilter`` / ``MaximumProjection...``  -> :func:`project`
  ``ts2d/tool.py:182-185``            ``sitk.Cast(res, sitk.sitkFloat32)``                           -> :func:`project_f32`
  ``prediction_worker.py:194-199``    nnU-Net ``ZScoreNormalization`` (third party)                  -> :func:`zscore`

PARITY PINNING of the projection: SimpleITK is absent offline, but the reference's own sample assets are outputs of this very
step and pin its arithmetic (tests/test_oracle.py::test_mean_projection_is_pinned_by_the_reference_assets):
``assets/sample_s0616.nrrd`` (stored as double, 261 coronal slices) holds in channel 0 exactly ``double(S) / 261`` for an
integer S in every pixel, ``assets/sample_s0332.nrrd`` (Float32, 269 slices) exactly ``float32(double(S) / 269)``, and channel 1
(maximum) is integer-valued in both: the mean of an integer volume is REAL-valued (ITK accumulates in
``NumericTraits<T>::RealType`` = double and divides in double), not truncated back to the integer type.
"""
from __future__ import annotations

import numpy as np


def dicom_orient_rai(arr: np.ndarray, direction) -> np.ndarray:
    """Voxel array of ``sitk.DICOMOrient(img, 'RAI')``: ``arr`` is indexed [z, y, x] (numpy order of a 3-D sitk image),
    ``direction`` the 9 row-major direction cosines (LPS).  Output index k grows along physical axis k (Right->Left,
    Anterior->Posterior, Inferior->Superior): for each physical axis take the image axis with the largest |cosine| not yet
    used, flipped when the cosine is negative."""
    D = np.asarray(direction, dtype=np.float64).reshape(3, 3)
    used, perm, flip = [], [], []
    for phys in range(3):
        best = max((a for a in range(3) if a not in used), key=lambda a: abs(D[phys, a]))
        used.append(best); perm.append(best); flip.append(D[phys, best] < 0)
    out = np.transpose(arr, [2 - perm[2 - k] for k in range(3)])      # numpy axis k <-> sitk axis 2 - k
    for k in range(3):
        if flip[k]:
            out = np.flip(out, axis=2 - k)
    return out


def project(vol: np.ndarray, mode: str, np_axis: int = 1) -> np.ndarray:
    """ITK projection of ``vol`` along numpy axis ``np_axis`` (coronal = sitk axis 1 = numpy axis 1 of [z, y, x]), axis dropped.
    max / min keep the input type; mean is float64: the accumulator adds the slices in index order in double and divides
    by the count in double (``itk::Functor::MeanAccumulator``)."""
    mode = mode.lower()
    v = np.moveaxis(vol, np_axis, 0)
    if mode in ('max', 'mip'):
        return v.max(axis=0)
    if mode == 'min':
        return v.min(axis=0)
    if mode in ('mean', 'avg'):
        acc = np.zeros(v.shape[1:], np.float64)
        for k in range(v.shape[0]):                                   # index order, one double addition per slice
            acc += v[k].astype(np.float64)
        return acc / np.float64(v.shape[0])
    raise ValueError(mode)


def project_f32(vol: np.ndarray, mode: str, np_axis: int = 1) -> np.ndarray:
    """``TS2D._project``: projection, then ``sitk.Cast(..., sitkFloat32)`` (one rounding of the double mean)."""
    return project(vol, mode, np_axis).astype(np.float32)


def coronal_projections_f32(arr: np.ndarray, direction) -> dict:
    """What ``TS2D._predict_model`` feeds the model for a 3-D volume: {'max', 'mean'} float32 [z, x] planes of the RAI volume."""
    r = dicom_orient_rai(arr, direction)
    return {m: project_f32(r, m, 1) for m in ('max', 'mean')}


def zscore(img: np.ndarray) -> np.ndarray:
    """nnU-Net ``ZScoreNormalization.run`` without mask: ``image.astype(float32)``; ``(image - mean) / max(std, 1e-8)`` with
    numpy's float32 ``mean()`` / ``std()``."""
    img = img.astype(np.float32, copy=True)
    mean = img.mean()
    std = img.std()
    img -= mean
    img /= max(std, 1e-8)
    return img


def zscore_stats64(img: np.ndarray):
    """mean and population standard deviation in float64 (what the device kernel reports beside the normalised planes)."""
    x = img.astype(np.float64)
    return float(x.mean()), float(x.std())

"""Image operations either side of the hot path, on :class:`nrrd.Image` (the stand-in for ``sitk.Image``).

Mirrors the subset of the reference's ``ts2d/core/util/image.py`` and ``ts2d/core/util/meta.py`` that ``TS2D.predict`` /
``Result.save`` use: ``reorient_image`` (:32-43), ``project`` (:46-101, modes max/mip/min/mean/avg), ``reduce_dimensions``
(:241-258), ``combine_segmentations`` (:490-510), ``split_channels`` (:512-520), ``get_label_mask`` (meta.py:51-62),
``set_annotation_meta`` / ``get_annotation_labels`` (meta.py:172-240,289-330).  SimpleITK is not installed here, so the ITK
filters are restated in numpy; where ITK behaviour could not be verified offline the docstring says [UPSTREAM-RECALL].
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from .nrrd import Image

_AXES = {'a': 2, 'ax': 2, 'axial': 2, 's': 0, 'sag': 0, 'sagittal': 0, 'c': 1, 'cor': 1, 'coronal': 1}


def axis_name_to_index(name: str) -> int:
    return _AXES[name.lower()]


def get_actual_dimension(img: Image) -> int:
    """Dimensions of size > 1 (reference image.py ``get_actual_dimension``)."""
    return sum(s > 1 for s in img.size)


def _np_axis(img: Image, sitk_axis: int) -> int:
    return img.dimension - 1 - sitk_axis


def reorient_image(img: Image, orient: str = 'RAI') -> Image:
    """``sitk.DICOMOrient(img, 'RAI')`` for 3-D images (2-D images pass through): permute / flip the axes so that the
    direction matrix becomes the identity in LPS space (ITK's 'RAI' = indices grow from Right, Anterior, Inferior).
    Oblique directions are snapped to the nearest axis like ITK's orientation filter does."""
    if img.dimension != 3 or orient.upper() != 'RAI':
        return img
    D = np.asarray(img.direction, dtype=np.float64).reshape(3, 3)          # columns = image axes in physical space
    perm, flips, used = [], [], set()
    for phys in range(3):                                                   # output axis `phys` <- input axis with the largest component
        cand = [(abs(D[phys, ax]), ax) for ax in range(3) if ax not in used]
        ax = max(cand)[1]
        used.add(ax)
        perm.append(ax)
        flips.append(D[phys, ax] < 0)
    arr = img.array
    np_perm = [_np_axis(img, perm[_np_axis(img, k)]) for k in range(3)] + ([3] if img.components > 1 else [])
    arr = np.transpose(arr, np_perm)
    size = list(img.size)
    origin = np.asarray(img.origin, dtype=np.float64)
    spacing = [img.spacing[ax] for ax in perm]
    new_size = [size[ax] for ax in perm]
    for k in range(3):
        if flips[k]:
            arr = np.flip(arr, axis=_np_axis(img, k))
            ax = perm[k]
            origin = origin + D[:, ax] * img.spacing[ax] * (size[ax] - 1)
    newD = np.stack([D[:, perm[k]] * (-1.0 if flips[k] else 1.0) for k in range(3)], axis=1)
    out = Image(np.ascontiguousarray(arr), tuple(spacing), tuple(float(o) for o in origin),
                tuple(float(v) for v in newD.reshape(-1)), img.components, dict(img.meta), img.space)
    assert list(out.size) == new_size
    return out


def project(img: Image, mode: str = 'max', axis=-1) -> Image:
    """``sitk.{Maximum,Minimum,Mean}ProjectionImageFilter`` along ``axis``: the projected axis keeps size 1 and its
    origin (reference image.py:97-100).  The mean is REAL-valued (float64: ITK accumulates and divides in double), also for
    integer volumes - pinned by the reference's own pre-projected assets (``sample_s0332`` / ``sample_s0616``: channel 0 is
    exactly ``double(sum) / n``; tests/test_oracle.py); the reference casts to Float32 afterwards (tool.py:182-185)."""
    ax = axis_name_to_index(axis) if isinstance(axis, str) else list(range(img.dimension))[axis]
    mode = str(mode).lower().strip()
    npax = _np_axis(img, ax)
    a = img.array
    if mode in ('max', 'mip'):
        r = a.max(axis=npax, keepdims=True)
    elif mode == 'min':
        r = a.min(axis=npax, keepdims=True)
    elif mode in ('avg', 'mean'):
        if np.issubdtype(a.dtype, np.integer):      # exact integer sum (15x faster than a float64 copy of a CT volume)
            r = a.sum(axis=npax, keepdims=True, dtype=np.int64).astype(np.float64) / np.float64(a.shape[npax])
        else:                                       # ITK's accumulator: slices added in index order, in double
            v = np.moveaxis(a, npax, 0)
            acc = np.zeros(v.shape[1:], np.float64)
            for k in range(v.shape[0]):
                acc += v[k]
            r = np.expand_dims(acc / np.float64(v.shape[0]), npax)
    else:
        raise RuntimeError(f"Unsupported filter mode: {mode}")
    return Image(np.ascontiguousarray(r), img.spacing, img.origin, img.direction, img.components, dict(img.meta), img.space)


_GPU_DTYPES = {'int16': 0, 'uint8': 1, 'float32': 2, 'uint16': 3, 'int32': 4}


def _reorient_plan(img: Image):
    """Axis permutation / flips of :func:`reorient_image` as (perm, flips, new geometry) without touching the voxels."""
    D = np.asarray(img.direction, dtype=np.float64).reshape(3, 3)
    perm, flips, used = [], [], set()
    for phys in range(3):
        ax = max((abs(D[phys, a]), a) for a in range(3) if a not in used)[1]
        used.add(ax); perm.append(ax); flips.append(D[phys, ax] < 0)
    return perm, flips


def project_coronal_gpu(img: Image, device: int = 0, zscore: bool = False):
    """max and mean coronal projections of a 3-D volume on the MI355X (C-ABI ``ts2d_project_coronal``): equals
    ``project(reorient_image(img), mode, 'coronal')`` for mode in (max, mean), as float32 images, without the host-side
    reorientation copy.  Returns {'max': Image, 'mean': Image} with size (nx, 1, nz).

    ``zscore=True`` (C-ABI ``ts2d_project_coronal_zscore``) also normalises both projections on the device - nnU-Net's
    ZScoreNormalization, float64 statistics - and returns them under ``'zscore'``: ``{'norm': [2, nz, nx] float32 in
    (max, mean) order, 'stats': (mean, std) x 2, 'box': non-zero bounding box}``; ``DefaultPreprocessor.run_case_npy`` uses it
    in place of its host pass when nnU-Net's crop-to-nonzero is the identity (``preprocess.py``)."""
    from . import _lib
    if img.dimension != 3 or img.components != 1 or img.array.dtype.name not in _GPU_DTYPES:
        raise RuntimeError(f"GPU projection needs a scalar 3-D volume of type {sorted(_GPU_DTYPES)}, found {img.array.dtype}")
    perm, flips = _reorient_plan(img)
    a = np.ascontiguousarray(img.array)
    view = np.transpose(a, [2 - perm[2 - k] for k in range(3)])          # numpy axis k = sitk axis 2 - k
    for k in range(3):
        if flips[k]:
            view = np.flip(view, axis=2 - k)
    nz, ny, nx = view.shape
    it = a.dtype.itemsize
    base = (view.__array_interface__['data'][0] - a.__array_interface__['data'][0]) // it
    sz, sy, sx = (s // it for s in view.strides)
    omax = np.empty((nz, nx), np.float32); omean = np.empty((nz, nx), np.float32)
    lib = _lib.load()
    zs = None
    if zscore:
        norm = np.empty((2, nz, nx), np.float32); stats = np.empty(4, np.float64); box = np.empty(4, np.int32)
        _lib.check(lib.ts2d_project_coronal_zscore(int(device), a.ctypes.data, a.size, _GPU_DTYPES[a.dtype.name], nz, ny, nx, sz, sy, sx, base,
                                                   omax.ctypes.data, omean.ctypes.data, norm.ctypes.data, stats.ctypes.data, box.ctypes.data),
                   'ts2d_project_coronal_zscore')
        zs = {'norm': norm, 'stats': tuple(float(v) for v in stats), 'box': tuple(int(v) for v in box), 'shape': (nz, nx)}
    else:
        _lib.check(lib.ts2d_project_coronal(int(device), a.ctypes.data, a.size, _GPU_DTYPES[a.dtype.name], nz, ny, nx, sz, sy, sx, base,
                                            omax.ctypes.data, omean.ctypes.data), 'ts2d_project_coronal')
    sp = tuple(img.spacing[ax] for ax in perm)
    geo = _reoriented_geometry(img, perm, flips)
    out = {m: Image(arr.reshape(nz, 1, nx), sp, geo[0], geo[1], 1, dict(img.meta), img.space) for m, arr in (('max', omax), ('mean', omean))}
    if zs is not None:
        out['zscore'] = zs
    return out


def _reoriented_geometry(img: Image, perm, flips):
    D = np.asarray(img.direction, dtype=np.float64).reshape(3, 3)
    size = list(img.size)
    origin = np.asarray(img.origin, dtype=np.float64)
    for k in range(3):
        if flips[k]:
            ax = perm[k]
            origin = origin + D[:, ax] * img.spacing[ax] * (size[ax] - 1)
    newD = np.stack([D[:, perm[k]] * (-1.0 if flips[k] else 1.0) for k in range(3)], axis=1)
    return tuple(float(o) for o in origin), tuple(float(v) for v in newD.reshape(-1))


def cast(img: Image, dtype) -> Image:
    return Image(img.array.astype(dtype), img.spacing, img.origin, img.direction, img.components, dict(img.meta), img.space)


def compose(channels: Sequence[Image]) -> Image:
    """``sitk.Compose``: scalar images -> one vector image (geometry of the first)."""
    first = channels[0]
    arr = np.stack([c.array for c in channels], axis=-1)
    return Image(arr, first.spacing, first.origin, first.direction, len(channels), dict(first.meta), first.space)


def split_channels(img: Image) -> List[Image]:
    if img.components == 1:
        return [img]
    return [Image(np.ascontiguousarray(img.array[..., i]), img.spacing, img.origin, img.direction, 1, dict(img.meta), img.space)
            for i in range(img.components)]


def reduce_dimensions(img: Image) -> Image:
    """``sitk.Extract`` collapsing the axes of size 1 (reference image.py:241-258, DIRECTIONCOLLAPSETOGUESS)."""
    size = img.size
    keep = [i for i, s in enumerate(size) if s > 1]
    if len(keep) == img.dimension:
        return img
    npkeep = sorted(_np_axis(img, k) for k in keep)
    shape = [img.array.shape[k] for k in npkeep] + ([img.components] if img.components > 1 else [])
    arr = img.array.reshape(shape)
    D = np.asarray(img.direction, dtype=np.float64).reshape(img.dimension, img.dimension)
    sub = D[np.ix_(keep, keep)]
    if abs(np.linalg.det(sub)) < 1e-6:
        sub = np.eye(len(keep))
    return Image(np.ascontiguousarray(arr), tuple(img.spacing[k] for k in keep), tuple(img.origin[k] for k in keep),
                 tuple(float(v) for v in sub.reshape(-1)), img.components, dict(img.meta), None)


def restore_dimension(seg: Image, ref: Image) -> Image:
    """``TS2D._restore_dimension`` (reference tool.py:188-193): put the collapsed axis back and copy ref's geometry."""
    shape = list(reversed(ref.size)) + ([seg.components] if seg.components > 1 else [])
    return Image(seg.array.reshape(shape), ref.spacing, ref.origin, ref.direction, seg.components,      # (a unit axis: a view, also of a
                 dict(seg.meta), ref.space)                                                             #  plane-major segmentation - export.py)


# ----------------------------------------------------------------------------- segmentation metadata (3D-Slicer keys)
def to_color_str_rgb_floats(color, sep: str = ' ') -> str:
    """Colour -> 'r g b' floats in [0,1] with three decimals, as the reference stamps ``Segment*_Color`` (``ts2d/core/util/
    color.py:81-85`` -> ``format_array(p=3)``): accepts '#RRGGBB' strings (the packaged label-colors.csv), integer 0..255
    triples and float 0..1 triples."""
    if isinstance(color, str):
        h = color.strip().lstrip('#')
        if len(h) != 6:
            raise ValueError(f"unsupported colour string: {color!r}")
        c = [int(h[i:i + 2], 16) for i in (0, 2, 4)]
    else:
        c = list(color)[:3]
        if len(c) != 3:
            raise ValueError(f"colour tuples need three components, found {color!r}")
        if any(not isinstance(v, (int, np.integer)) for v in c):
            c = [int(min(max(float(v), 0.0), 1.0) * 255) for v in c]          # reference tuple_to_color: floats are 0..1
        c = [min(max(int(v), 0), 255) for v in c]
    return sep.join(np.format_float_positional(v / 255.0, precision=3, unique=False) for v in c)


def set_annotation_meta(seg: Image, names: Optional[Dict[int, str]] = None, colors: Optional[Dict[str, Sequence[float]]] = None):
    """3D-Slicer ``Segment{i}_*`` keys; multilabel (multi-component) => ``Layer = i``, ``LabelValue = 1`` and channel i
    <-> label value i + 1 (reference meta.py:172-240)."""
    multilabel = seg.components > 1
    labels = list(range(seg.components)) if multilabel else [int(v) for v in np.unique(seg.array) if v != 0]
    meta = {k: v for k, v in seg.meta.items() if not k.startswith('Segment')}
    for seg_id, label in enumerate(labels):
        pat = f'Segment{seg_id}_{{}}'
        if names is None:
            continue
        meta[pat.format('ID')] = str(seg_id)
        meta[pat.format('Layer')] = str(label if multilabel else 0)
        meta[pat.format('LabelValue')] = str(1 if multilabel else label)
        name = names.get(label + 1 if multilabel else label)
        if name is not None:
            meta[pat.format('Name')] = str(name)
            meta[pat.format('NameAutoGenerated')] = '0'
            color = (colors or {}).get(name)
            if color is not None:
                meta[pat.format('Color')] = to_color_str_rgb_floats(color)
                meta[pat.format('ColorAutoGenerated')] = '0'
    seg.meta = meta


def get_annotation_labels(seg: Image) -> Dict[str, dict]:
    """name -> {value, color?} from the ``Segment*`` keys (reference meta.py:289-330); multilabel value = Layer + 1."""
    out: Dict[str, dict] = {}
    i = 0
    while f'Segment{i}_ID' in seg.meta or f'Segment{i}_Name' in seg.meta:
        name = seg.meta.get(f'Segment{i}_Name', f'Segment_{i}')
        layer = int(seg.meta.get(f'Segment{i}_Layer', 0))
        lv = int(seg.meta.get(f'Segment{i}_LabelValue', i + 1))
        info = {'value': layer + 1 if seg.components > 1 else lv}
        if f'Segment{i}_Color' in seg.meta:
            info['color'] = [float(v) for v in seg.meta[f'Segment{i}_Color'].split()]
        out[name] = info
        i += 1
    return out


def get_label_mask(seg: Image, label: int) -> Image:
    if seg.components > 1:
        assert 1 <= label <= seg.components, f'Invalid label number: {label} (label-map segmentation has {seg.components} channels)'
        arr = (seg.array[..., label - 1] > 0).astype(np.uint8)
    else:
        arr = (seg.array == label).astype(np.uint8)
    return Image(arr, seg.spacing, seg.origin, seg.direction, 1, {}, seg.space)


def combine_segmentations(segs: Sequence[Image]) -> Image:
    """Per label of every sub-model: channel > 0 mask -> one multi-component image in sub-model then label order
    (reference image.py:490-510; 117 components for the full ts2d-v2 set)."""
    names, colors, plan = {}, {}, []
    for seg in segs:
        vals = []
        for name, info in get_annotation_labels(seg).items():
            names[len(names) + 1] = name
            if info.get('color') is not None:
                colors[name] = info['color']
            vals.append(int(info['value']))
        plan.append((seg, vals))
    n = len(names)
    if n == 0:          # (the reference's sitk.Compose([]) raises on an empty list as well)
        raise ValueError('combine_segmentations: none of the segmentations carries a label (Segment*_Name metadata)')
    first = segs[0]
    if n == 1:
        seg, vals = next((s, v) for s, v in plan if v)
        out = get_label_mask(seg, vals[0])
    else:
        # The masks are written PLANE by plane ([label, ...] memory) and handed out as the interleaved [..., label] view sitk.Compose's
        # vector image has: 117 stride-117 byte scatters of a 644 x 337 case cost 27 ms of the 55 ms a case takes, the planes 2 ms
        # (nrrd.write serialises the logical order either way).
        spatial = first.array.shape[:first.array.ndim - (1 if first.components > 1 else 0)]
        planes = np.empty((n,) + tuple(spatial), dtype=np.uint8)
        o = 0
        for seg, vals in plan:
            if not vals:
                continue
            if seg.components > 1:
                assert all(1 <= v <= seg.components for v in vals), f'Invalid label number in {vals} (label-map segmentation has {seg.components} channels)'
                src = np.moveaxis(seg.array, -1, 0)
                idx = [v - 1 for v in vals]
                sel = src[idx[0]:idx[0] + len(idx)] if idx == list(range(idx[0], idx[0] + len(idx))) else src[idx]
                np.greater(sel, 0, out=planes[o:o + len(vals)])
            else:
                for i, v in enumerate(vals):
                    np.equal(seg.array, v, out=planes[o + i])
            o += len(vals)
        out = Image(np.moveaxis(planes, 0, -1), first.spacing, first.origin, first.direction, n, {}, first.space)
    set_annotation_meta(out, names=names, colors=colors)
    return out

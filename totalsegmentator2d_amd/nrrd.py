"""Pure-Python NRRD0004 reader / writer (raw + gzip), the on-disk format either side of the hot path.

The reference moves every image through ``SimpleITK`` (``sitk.ReadImage`` / ``sitk.WriteImage(img, path, True)``:
reference ``ts2d/core/inference/nnu.py:196-198,221``, ``ts2d/tool.py:263-264``); SimpleITK is not installed here,
so this module implements the subset ITK's ``NrrdImageIO`` produces: little-endian scalar or ``vector`` images,
``space directions`` / ``space origin`` geometry, free-form ``key:=value`` metadata (3D-Slicer ``Segment*`` keys
written by reference ``ts2d/core/util/meta.py:172-240``).
"""
from __future__ import annotations

import gzip
import io
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

_TYPES = {
    'signed char': 'i1', 'int8': 'i1', 'int8_t': 'i1', 'uchar': 'u1', 'unsigned char': 'u1', 'uint8': 'u1', 'uint8_t': 'u1',
    'short': 'i2', 'short int': 'i2', 'signed short': 'i2', 'signed short int': 'i2', 'int16': 'i2', 'int16_t': 'i2',
    'ushort': 'u2', 'unsigned short': 'u2', 'unsigned short int': 'u2', 'uint16': 'u2', 'uint16_t': 'u2',
    'int': 'i4', 'signed int': 'i4', 'int32': 'i4', 'int32_t': 'i4', 'uint': 'u4', 'unsigned int': 'u4', 'uint32': 'u4', 'uint32_t': 'u4',
    'longlong': 'i8', 'long long': 'i8', 'long long int': 'i8', 'signed long long': 'i8', 'int64': 'i8', 'int64_t': 'i8',
    'ulonglong': 'u8', 'unsigned long long': 'u8', 'unsigned long long int': 'u8', 'uint64': 'u8', 'uint64_t': 'u8',
    'float': 'f4', 'double': 'f8',
}
_NAMES = {'i1': 'signed char', 'u1': 'unsigned char', 'i2': 'short', 'u2': 'unsigned short', 'i4': 'int',
          'u4': 'unsigned int', 'i8': 'long long', 'u8': 'unsigned long long', 'f4': 'float', 'f8': 'double'}


@dataclass
class Image:
    """Minimal stand-in for ``sitk.Image``: ``array`` is indexed like ``sitk.GetArrayFromImage`` -
    ``[z, y, x]`` / ``[y, x]`` with a trailing component axis for vector images.

    Memory layout: ``sitk.GetArrayFromImage`` always hands out a C-contiguous ``[..., component]`` array.  The segmentations this
    package produces (``export.segmentation_to_image``, ``image.combine_segmentations``, ``image.restore_dimension``) are written
    plane by plane and ``array`` is then the interleaved VIEW of a ``[component, ...]`` buffer: same shape, same values under
    indexing, but ``array.flags.c_contiguous`` is False.  Code that reinterprets the memory (``.view(dtype)``, ``.ctypes.data``,
    ``.data``, ``.tobytes()`` expecting interleaved bytes) must go through ``contiguous()`` first; ``nrrd.write`` does."""
    array: np.ndarray
    spacing: Tuple[float, ...]
    origin: Tuple[float, ...]
    direction: Tuple[float, ...]                 # row-major dim x dim direction cosines (columns = axis directions)
    components: int = 1
    meta: Dict[str, str] = field(default_factory=dict)
    space: Optional[str] = None

    @property
    def dimension(self) -> int:
        return self.array.ndim - (1 if self.components > 1 else 0)

    @property
    def size(self) -> Tuple[int, ...]:
        """(x, y[, z]) like ``sitk.Image.GetSize``."""
        shp = self.array.shape[:self.dimension]
        return tuple(int(s) for s in reversed(shp))

    def contiguous(self) -> "Image":
        """This image with a C-contiguous ``array`` (a copy only when ``array`` is a plane-major view)."""
        if self.array.flags['C_CONTIGUOUS']:
            return self
        return Image(np.ascontiguousarray(self.array), self.spacing, self.origin, self.direction, self.components, dict(self.meta), self.space)

    def copy_geometry_from(self, other: "Image"):
        self.spacing, self.origin, self.direction, self.space = other.spacing, other.origin, other.direction, other.space


def _parse_vec(tok: str) -> Optional[List[float]]:
    tok = tok.strip()
    if tok == 'none':
        return None
    return [float(v) for v in tok.strip('()').split(',')]


def read(path: str) -> Image:
    with open(path, 'rb') as f:
        raw = f.read()
    if not raw.startswith(b'NRRD'):
        raise RuntimeError(f"{path}: not an NRRD file")
    end = raw.find(b'\n\n')
    alt = raw.find(b'\r\n\r\n')
    if end < 0 or (0 <= alt < end):
        end, skip = alt, 4
    else:
        skip = 2
    if end < 0:
        raise RuntimeError(f"{path}: NRRD header is not terminated")
    fields: Dict[str, str] = {}
    meta: Dict[str, str] = {}
    for line in raw[:end].decode('latin1').splitlines()[1:]:
        line = line.rstrip()
        if not line or line.startswith('#'):
            continue
        if ':=' in line:
            k, v = line.split(':=', 1)
            meta[k] = v.replace('\\n', '\n')
        elif ':' in line:
            k, v = line.split(':', 1)
            fields[k.strip().lower()] = v.strip()
    if 'data file' in fields or 'datafile' in fields:
        raise NotImplementedError(f"{path}: detached NRRD headers are not supported")
    dt = _TYPES.get(fields['type'].lower())
    if dt is None:
        raise RuntimeError(f"{path}: unsupported NRRD type '{fields['type']}'")
    endian = '>' if fields.get('endian', 'little').lower() == 'big' else '<'
    dtype = np.dtype(endian + dt) if dt[1] != '1' else np.dtype(dt)
    sizes = [int(s) for s in fields['sizes'].split()]
    enc = fields.get('encoding', 'raw').lower()
    data = raw[end + skip:]
    if enc in ('gzip', 'gz'):
        data = zlib.decompress(data, 16 + zlib.MAX_WBITS)
    elif enc != 'raw':
        raise NotImplementedError(f"{path}: NRRD encoding '{enc}' is not supported")
    n = int(np.prod(sizes))
    arr = np.frombuffer(data, dtype=dtype, count=n).reshape(list(reversed(sizes)))   # slowest axis first
    kinds = fields.get('kinds', '').split()
    dirs = [_parse_vec(t) for t in fields['space directions'].split()] if 'space directions' in fields else None
    comp_axis = None
    for ax in range(len(sizes)):
        if (kinds and kinds[ax] in ('vector', 'list', 'RGB-color', 'RGBA-color', 'covariant-vector', 'point')) or \
                (dirs is not None and ax < len(dirs) and dirs[ax] is None):
            comp_axis = ax
            break
    components = 1
    if comp_axis is not None:
        if comp_axis != 0:
            raise NotImplementedError(f"{path}: component axis must be the fastest axis")
        components = sizes[0]
        dom_dirs = [d for d in dirs if d is not None] if dirs else None
    else:
        dom_dirs = dirs
    dim = len(sizes) - (1 if comp_axis is not None else 0)
    arr = np.ascontiguousarray(arr.astype(dtype.newbyteorder('=')))
    if dom_dirs:
        spacing = tuple(float(np.linalg.norm(d)) for d in dom_dirs)
        cols = [np.asarray(d) / (s if s > 0 else 1.0) for d, s in zip(dom_dirs, spacing)]
        sd = len(cols[0])
        direction = tuple(float(cols[j][i]) for i in range(sd) for j in range(dim))
    else:
        sp = fields.get('spacings')
        spacing = tuple(float(s) for s in sp.split()[-dim:]) if sp else (1.0,) * dim
        direction = tuple(1.0 if i == j else 0.0 for i in range(dim) for j in range(dim))
    origin = tuple(_parse_vec(fields['space origin'])) if 'space origin' in fields else (0.0,) * dim
    space = fields.get('space')
    flips = _LPS_FLIPS.get(space.strip().lower()) if space else None
    if flips:
        # ITK's NrrdImageIO hands every anatomical space to the application as LPS: the world axes named R (vs L) and A (vs P)
        # are negated in the direction vectors and the origin.  Everything downstream (reorient_image, projection axis) reads
        # the direction matrix as LPS, so a RAS file must not be taken literally (it would come out mirrored in L/R and A/P).
        sd = len(origin)
        d = np.asarray(direction, dtype=np.float64).reshape(sd, -1)
        for ax in flips:
            if ax < sd:
                d[ax] = -d[ax]
        direction = tuple(float(v) + 0.0 for v in d.reshape(-1))            # (+ 0.0: no negative zeros)
        origin = tuple((-o if i in flips else o) + 0.0 for i, o in enumerate(origin))
        space = 'left-posterior-superior' + ('-time' if space.strip().lower().endswith('time') or space.strip().upper().endswith('T') else '')
    return Image(arr, spacing, origin, direction, components, meta, space)


# world axes to negate so that a named anatomical NRRD space becomes LPS (what ITK's reader does; other spaces pass through)
_LPS_FLIPS = {
    'right-anterior-superior': (0, 1), 'ras': (0, 1), 'right-anterior-superior-time': (0, 1), 'rast': (0, 1),
    'left-anterior-superior': (1,), 'las': (1,), 'left-anterior-superior-time': (1,), 'last': (1,),
}


def _fmt(v: float) -> str:
    return repr(float(v)) if float(v) != int(v) or abs(v) > 1e15 else str(int(v))


def write(img: Image, path: str, compress: bool = True):
    """Writes like ``sitk.WriteImage(img, path, useCompression)``: NRRD0004, component axis first."""
    arr = np.ascontiguousarray(img.array)
    dt = arr.dtype.newbyteorder('<') if arr.dtype.itemsize > 1 else arr.dtype
    code = np.dtype(dt).kind + str(np.dtype(dt).itemsize)
    if code not in _NAMES:
        raise RuntimeError(f"unsupported pixel type {arr.dtype}")
    dim = img.dimension
    sizes = ([img.components] if img.components > 1 else []) + list(img.size)
    sd = len(img.origin)
    cols = []
    for j in range(dim):
        col = [img.direction[i * dim + j] * img.spacing[j] for i in range(sd)] if len(img.direction) == sd * dim else \
              [img.spacing[j] if i == j else 0.0 for i in range(sd)]
        cols.append('(' + ','.join(_fmt(c) for c in col) + ')')
    lines = ['NRRD0004', '# Complete NRRD file format specification at:', '# http://teem.sourceforge.net/nrrd/format.html',
             f'type: {_NAMES[code]}', f'dimension: {len(sizes)}']
    lines.append(f'space: {img.space}' if img.space else f'space dimension: {sd}')
    lines.append('sizes: ' + ' '.join(str(s) for s in sizes))
    lines.append('space directions: ' + ' '.join((['none'] if img.components > 1 else []) + cols))
    lines.append('kinds: ' + ' '.join((['vector'] if img.components > 1 else []) + ['domain'] * dim))
    if np.dtype(dt).itemsize > 1:
        lines.append('endian: little')
    lines.append('encoding: ' + ('gzip' if compress else 'raw'))
    lines.append('space origin: (' + ','.join(_fmt(o) for o in img.origin) + ')')
    for k, v in img.meta.items():
        lines.append(f'{k}:=' + str(v).replace('\n', '\\n'))
    payload = arr.astype(dt, copy=False).tobytes()
    if compress:
        buf = io.BytesIO()
        with gzip.GzipFile(fileobj=buf, mode='wb', compresslevel=2, mtime=0) as g:
            g.write(payload)
        payload = buf.getvalue()
    with open(path, 'wb') as f:
        f.write(('\n'.join(lines) + '\n\n').encode('latin1'))
        f.write(payload)

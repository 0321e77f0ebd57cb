"""Randomised check of the device-side coronal projection (+ z-score) against oracle/input_oracle.py: random volume shapes, voxel types and
signed-permutation direction matrices (every DICOMOrient case), bit for bit on the planes, float64 accuracy on the statistics.
    python scripts/gpu_fuzz_projection.py SEED N"""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import input_oracle as IO
from totalsegmentator2d_amd import image, nrrd

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
perms = list(itertools.permutations(range(3)))
for t in range(n):
    shape = tuple(int(rng.integers(1, 70)) if rng.random() < 0.85 else int(rng.integers(70, 300)) for _ in range(3))
    if np.prod(shape) > 4_000_000:
        shape = (shape[0], min(shape[1], 60), shape[2])
    dt = [np.int16, np.uint8, np.float32, np.uint16, np.int32][int(rng.integers(0, 5))]
    if dt == np.float32:
        arr = (rng.normal(0, 300, shape) + rng.normal(0, 500)).astype(np.float32)
    else:
        info = np.iinfo(dt)
        lo, hi = max(info.min, -2000), min(info.max, 4000)
        arr = rng.integers(lo, hi, shape).astype(dt)
        if rng.random() < 0.3:                                  # a zero frame: the non-zero box is not the whole plane
            arr[: shape[0] // 4] = 0; arr[:, :, : shape[2] // 3] = 0
    p = perms[int(rng.integers(0, 6))]
    D = np.zeros((3, 3))
    for c in range(3):
        D[p[c], c] = 1.0 if rng.random() < 0.5 else -1.0
    vol = nrrd.Image(arr, tuple(float(v) for v in rng.uniform(0.5, 3.0, 3)), tuple(float(v) for v in rng.uniform(-50, 50, 3)),
                     tuple(float(v) for v in D.reshape(-1)), 1, {}, None)
    got = image.project_coronal_gpu(vol, zscore=True)
    ref = IO.coronal_projections_f32(vol.array, vol.direction)
    r = image.reorient_image(vol)
    ok = True
    for mode in ('max', 'mean'):
        g = got[mode]
        ok &= g.size == (r.size[0], 1, r.size[2]) and g.array.dtype == np.float32 and np.array_equal(g.array[:, 0, :], ref[mode])
    zs = got['zscore']
    for k, m in enumerate(('max', 'mean')):
        want = IO.zscore(ref[m])
        mean64, std64 = IO.zscore_stats64(ref[m])
        ok &= bool(np.abs(zs['norm'][k] - want).max() <= 4e-6 * max(1.0, float(np.abs(want).max())))
        ok &= abs(zs['stats'][2 * k] - mean64) <= 1e-9 * max(1.0, abs(mean64)) and abs(zs['stats'][2 * k + 1] - std64) <= 1e-9 * max(1.0, std64)
    print(f'{t:3d} shape={shape} {np.dtype(dt).name} perm={p} signs={[int(D[p[c], c]) for c in range(3)]}: {"ok" if ok else "MISMATCH"}', flush=True)
    assert ok
print('all ok')

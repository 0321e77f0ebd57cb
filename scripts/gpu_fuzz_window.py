"""Randomised sweep of the device-side sliding window (ts2d_engine_predict_tiled: gather, mirroring, fp16 Gaussian aggregation) against
the host numpy restatement fed with the SAME engine's per-tile logits - must agree bit for bit - and against the full torch oracle
pipeline within float16 resolution.  Random image extents / patch sizes / step sizes / mirror axes / folds / Z."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from totalsegmentator2d_amd import weights, prng
from totalsegmentator2d_amd.predictor import HIPnnUNetPredictor

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for t in range(n):
    ns = int(rng.integers(2, 4)); mult = 2 ** (ns - 1)
    feats = [32] + [int(rng.choice([32, 64])) for _ in range(ns - 1)]
    K = int(rng.integers(1, 8)); cin = int(rng.integers(1, 3))
    patch = (mult * int(rng.integers(4, 17)), 32 * int(rng.integers(1, 4)))
    Z = int(rng.choice([1, 1, 2]))
    shape = (Z, int(rng.integers(patch[0] // 2, 3 * patch[0])), int(rng.integers(patch[1] // 2, 3 * patch[1])))
    step = float(rng.choice([0.5, 0.5, 0.75, 1.0, 0.3]))
    mirror = [None, (0,), (1,), (0, 1)][int(rng.integers(0, 4))]
    folds = int(rng.choice([1, 1, 2]))
    arch = cases.unet(ns, feats, K, cin=cin)
    blobs = [weights.pack_blob(arch, weights.synthetic_state_dict(arch, 500 + 10 * t + f)) for f in range(folds)]
    data = prng.normal_f32(600 + t, 999, (cin,) + shape)
    dev = HIPnnUNetPredictor(tile_step_size=step, use_mirroring=mirror is not None)
    dev.manual_initialization(arch, blobs, patch, inference_allowed_mirroring_axes=mirror)
    try:
        engines = dev.engines
        host = HIPnnUNetPredictor(tile_step_size=step, use_mirroring=mirror is not None,
                                  network=lambda batch, fold: engines[fold].forward(np.ascontiguousarray(batch))[0])
        host.manual_initialization(arch, blobs, patch, inference_allowed_mirroring_axes=mirror)
        a = dev.predict_logits_from_preprocessed_data(data).cpu().numpy()
        b = host.predict_logits_from_preprocessed_data(data).cpu().numpy()
    finally:
        dev.close()
    ok = a.dtype == b.dtype == np.float16 and a.shape == (K,) + shape and np.array_equal(a, b)
    print(f'{t:2d} feats={feats} K={K} cin={cin} data={shape} patch={patch} step={step} mirror={mirror} folds={folds}: bit-identical={ok}', flush=True)
    assert ok
print('all bit-identical')

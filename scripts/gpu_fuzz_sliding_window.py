"""Randomised check of the device-side sliding window (ts2d_engine_predict_tiled: tile gather, mirroring, float16 Gaussian aggregation,
division, inf flag) against the host restatement (tests/host_predictor.py) fed with the SAME engine's per-tile logits: bit-identical
float16 logits on random image shapes, patch sizes, step sizes, mirror axes, fold counts and tile dtypes.
    python scripts/gpu_fuzz_sliding_window.py SEED N"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from tests.host_predictor import HostLogicPredictor
from totalsegmentator2d_amd import weights, prng
from totalsegmentator2d_amd.predictor import HIPnnUNetPredictor
from totalsegmentator2d_amd.engine import Engine
# The device path runs the network in chunks of <= 64 rows, the host restatement in one batch: bit-identity of the AGGREGATION needs a network whose
# result does not depend on the batch a tile travels in - the small-batch dispatch of round 6 ("sbk") off (with it on, the two sides may differ by one
# float16 ulp where a chunk and the whole batch fall on different sides of a fill threshold: seed 611, case 30).
Engine.default_options = {'sbk': 0}

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for t in range(n):
    ns = int(rng.integers(2, 4))
    feats = [32] + [int(rng.choice([32, 64])) for _ in range(ns - 1)]
    strides = None
    if rng.random() < 0.3:
        strides = [(1, 1)] + [(2, 2)] * (ns - 2) + [[(2, 1), (1, 2)][int(rng.integers(0, 2))]]
    arch = cases.unet(ns, feats, int(rng.integers(1, 9)), cin=int(rng.integers(1, 3)), nconv=1, strides=strides)
    dy, dx = arch.divisors
    patch = (dy * int(rng.integers(max(1, 16 // dy), 96 // dy + 1)), max(32, dx * int(rng.integers(1, 128 // dx + 1))) // 32 * 32)
    patch = (max(patch[0], dy), max(patch[1], 32))
    Z = int(rng.integers(1, 3))
    shape = (Z, int(rng.integers(5, 3 * patch[0])), int(rng.integers(5, 3 * patch[1])))
    step = float(rng.choice([0.3, 0.5, 0.75, 1.0]))
    mirror = [None, (0,), (1,), (0, 1)][int(rng.integers(0, 4))]
    folds = int(rng.integers(1, 3))
    order = ['float', 'half'][int(rng.integers(0, 2))]
    blobs = [weights.pack_blob(arch, weights.synthetic_state_dict(arch, 900 + 7 * t + f)) for f in range(folds)]
    data = prng.normal_f32(1000 + t, 999, (arch.input_channels,) + shape)
    dev = HIPnnUNetPredictor(tile_step_size=step, use_mirroring=mirror is not None, tile_dtype=order)
    dev.manual_initialization(arch, blobs, patch, inference_allowed_mirroring_axes=mirror)
    try:
        engines = dev.engines
        host = HostLogicPredictor(network=lambda batch, fold: engines[fold].forward(np.ascontiguousarray(batch))[0],
                                  tile_step_size=step, use_mirroring=mirror is not None, tile_dtype=order)
        host.manual_initialization(arch, blobs, patch, inference_allowed_mirroring_axes=mirror)
        a = dev.predict_logits_from_preprocessed_data(data).cpu().numpy()
        b = host.predict_logits_from_preprocessed_data(data).cpu().numpy()
    finally:
        dev.close()
    same = a.dtype == b.dtype == np.float16 and a.shape == b.shape and np.array_equal(a, b)
    print(f'{t:3d} stages={ns} feats={feats} strides_last={None if strides is None else strides[-1]} K={arch.num_classes} cin={arch.input_channels} image={shape} '
          f'patch={patch} step={step} mirror={mirror} folds={folds} tile={order}: {"identical" if same else "DIFFERENT"}', flush=True)
    assert same, float(np.abs(a.astype(np.float32) - b.astype(np.float32)).max())
print('all identical')

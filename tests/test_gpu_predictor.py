"""Drop-in predictor on the GPU: sliding window + mirroring + fold ensemble through the HIP engine, and BASELINE
config 1 (the reference's sample_s0616.nrrd, 2 tiles x 4 mirror passes)."""
import os

import numpy as np
import pytest

from tests import cases
from tests.conftest import golden, blob_for, GOLDEN
from totalsegmentator2d_amd import prng
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd.predictor import HIPnnUNetPredictor
from tests.host_predictor import HostLogicPredictor

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _batch_independent_network():
    """Device aggregation vs the host restatement is compared bit for bit; the device path feeds the network chunks of <= 64 rows, the host side one
    batch - so the network must not depend on the batch a tile travels in: round 6's small-batch dispatch ("sbk") off.  (The product default is covered
    with tolerances by tests/test_gpu_surface.py and tests/test_gpu_small_batch.py.)"""
    from totalsegmentator2d_amd.engine import Engine as _E
    old = dict(_E.default_options)
    _E.default_options = {**old, 'sbk': 0}
    yield
    _E.default_options = old


@pytest.mark.parametrize('order', ['float', 'half'])
@pytest.mark.parametrize('name', list(cases.SW_CASES))
def test_sliding_window_goldens(name, order):
    arch, shape, patch, step, mirror, folds, seed = cases.SW_CASES[name]
    blobs = [blob_for(arch, seed + f)[1] for f in range(folds)]
    data = prng.normal_f32(seed, 999, (arch.input_channels,) + tuple(shape))
    p = HIPnnUNetPredictor(tile_step_size=step, use_mirroring=mirror is not None, tile_dtype=order)
    p.manual_initialization(arch, blobs, patch, inference_allowed_mirroring_axes=mirror)
    try:
        out = p.predict_logits_from_preprocessed_data(data).cpu().numpy()
    finally:
        p.close()
    g = golden(name)['logits_f16' if order == 'float' else 'logits_f16_half']
    assert out.dtype == np.float16 and out.shape == g.shape
    # reference end-of-pipeline logits are float16 (about 3 significant digits): allow 2 half-ulps at |x| <= 8
    assert np.abs(out.astype(np.float32) - g.astype(np.float32)).max() <= 1.6e-2
    assert (out != g).mean() < 0.05


def test_config1_sample_s0616_canonical_net():
    """BASELINE config 1: real 2-channel coronal projection -> z-score -> [2,1,644,337] -> pad to 512 wide -> 2 tiles
    x 4 mirror passes -> fp16 aggregation; synthetic seeded weights (the Zenodo weights are not available offline)."""
    from oracle import torch_oracle as O
    arch = UNetArch.canonical()
    sd, blob = blob_for(arch, 1)
    p = HIPnnUNetPredictor()                      # reference defaults: step 0.5, mirroring on (wrapper.py:65-66)
    p.manual_initialization(arch, [blob], (512, 512))
    try:
        pre = p.configuration_manager.preprocessor_class(verbose=False)
        data, _, props = pre.run_case([os.path.join(GOLDEN, 'assets', 'sample_s0616.nrrd')], None, p.plans_manager,
                                      p.configuration_manager, p.dataset_json)
        assert data.shape == (2, 1, 644, 337)
        out = p.predict_logits_from_preprocessed_data(data).cpu().numpy()
    finally:
        p.close()
    assert out.shape == (18, 1, 644, 337) and out.dtype == np.float16
    ref = O.predict_logits(arch, [sd], data, (512, 512), 0.5, (0, 1)).numpy()
    d = np.abs(out.astype(np.float32) - ref.astype(np.float32))
    assert d.max() <= 1.6e-2 and (out != ref).mean() < 0.05
    seg = (out.astype(np.float32) > 1.5 * 2.0 ** -24)
    seg_ref = O.logits_to_mask(ref).numpy().astype(bool)
    assert (seg != seg_ref).mean() < 1e-3
    # the differing mask bits are tolerance flips: the oracle's aggregated logit lies within the logit error of the threshold
    flips = seg != seg_ref
    if flips.any():
        assert np.abs(ref.astype(np.float32)[flips] - 1.5 * 2.0 ** -24).max() <= d.max()


@pytest.mark.parametrize('order', ['float', 'half'])
@pytest.mark.parametrize('name', list(cases.SW_CASES))
def test_device_aggregation_is_bit_identical_to_host_aggregation(name, order):
    """The device-side gather / mirror-average / fp16 Gaussian aggregation (ts2d_engine_predict_tiled) against the host
    numpy implementation fed with the SAME engine's per-tile logits: must agree bit for bit (same half arithmetic, same
    tile order)."""
    arch, shape, patch, step, mirror, folds, seed = cases.SW_CASES[name]
    blobs = [blob_for(arch, seed + f)[1] for f in range(folds)]
    data = prng.normal_f32(seed, 999, (arch.input_channels,) + tuple(shape))
    dev = HIPnnUNetPredictor(tile_step_size=step, use_mirroring=mirror is not None, tile_dtype=order)
    dev.manual_initialization(arch, blobs, patch, inference_allowed_mirroring_axes=mirror)
    try:
        engines = dev.engines

        def net(batch, fold):
            return engines[fold].forward(np.ascontiguousarray(batch))[0]
        host = HostLogicPredictor(network=net, tile_step_size=step, use_mirroring=mirror is not None, tile_dtype=order)
        host.manual_initialization(arch, blobs, patch, inference_allowed_mirroring_axes=mirror)
        a = dev.predict_logits_from_preprocessed_data(data).cpu().numpy()
        b = host.predict_logits_from_preprocessed_data(data).cpu().numpy()
    finally:
        dev.close()
    assert a.dtype == b.dtype == np.float16 and np.array_equal(a, b)


def test_infinite_aggregated_logit_is_detected_on_the_device():
    """Upstream raises 'Encountered inf in predicted array' after the fp16 aggregation; the engine evaluates that predicate in
    sw_aggregate (ts2d_engine_tiled_inf_flag).  A head bias beyond the float16 range forces it."""
    from totalsegmentator2d_amd import weights
    arch, shape, patch, step, mirror, folds, seed = cases.SW_CASES['sw_2tiles_mirror']
    sd, blob = blob_for(arch, seed)
    data = prng.normal_f32(seed, 999, (arch.input_channels,) + tuple(shape))
    p = HIPnnUNetPredictor(tile_step_size=step, use_mirroring=True)
    p.manual_initialization(arch, [blob], patch, inference_allowed_mirroring_axes=mirror)
    try:
        p.predict_logits_from_preprocessed_data(data)
        assert p.engines[0].last_tiled_inf is False
    finally:
        p.close()
    sd = dict(sd)
    key = [k for k in sd if 'seg_layers' in k and k.endswith('bias')][-1]
    sd[key] = np.full_like(sd[key], 1e6)                       # > 65504: every aggregated logit overflows float16
    p = HIPnnUNetPredictor(tile_step_size=step, use_mirroring=True)
    p.manual_initialization(arch, [weights.pack_blob(arch, sd)], patch, inference_allowed_mirroring_axes=mirror)
    try:
        with pytest.raises(RuntimeError, match='Encountered inf'):
            p.predict_logits_from_preprocessed_data(data)
        assert p.engines[0].last_tiled_inf is True
    finally:
        p.close()


def test_runs_on_different_streams_share_the_workspace_in_order():
    """One handle, one activation workspace: an asynchronous torch-tensor forward (side stream) followed IMMEDIATELY by a
    sliding-window call (the engine's own stream) of a shape that fits the reserved workspace must not overwrite the first
    run's activations - the engine orders the runs with an end-of-run event (no host synchronisation in between)."""
    import torch
    from totalsegmentator2d_amd.engine import Engine
    arch = cases.unet(4, (32, 64, 128, 256), 5)
    _, blob = blob_for(arch, 41)
    x = torch.from_numpy(prng.normal_f32(41, 1000, (40, 2, 128, 128))).cuda()
    img = prng.normal_f32(42, 999, (2, 128, 128))
    with Engine(arch, blob) as e:
        ref, _ = e.forward(x)
        torch.cuda.synchronize()
        ref = ref.clone()
        t16, _ = e.predict_tiled(img, (128, 128), [(0, 0)], (0, 1), None)
        for _ in range(4):
            lg, _ = e.forward(x)                                              # asynchronous, torch side stream
            o16, _ = e.predict_tiled(img, (128, 128), [(0, 0)], (0, 1), None)   # engine stream, same workspace, no sync
            torch.cuda.synchronize()
            assert torch.equal(lg, ref)
            assert np.array_equal(o16, t16)


def test_randomised_sliding_windows_are_bit_identical_to_the_host_restatement():
    """A fixed-seed slice of scripts/gpu_fuzz_sliding_window.py: random image shapes (smaller and larger than the patch, Z = 1 / 2),
    patch sizes, step sizes, mirror axes, fold counts, tile dtypes and a last (2, 1) / (1, 2) stage - device gather / mirror average /
    float16 Gaussian aggregation against the host restatement fed with the same engine's per-tile logits, bit for bit."""
    from totalsegmentator2d_amd import weights
    rng = np.random.default_rng(5)
    for t in range(10):
        ns = int(rng.integers(2, 4))
        feats = [32] + [int(rng.choice([32, 64])) for _ in range(ns - 1)]
        strides = None
        if rng.random() < 0.3:
            strides = [(1, 1)] + [(2, 2)] * (ns - 2) + [[(2, 1), (1, 2)][int(rng.integers(0, 2))]]
        arch = cases.unet(ns, feats, int(rng.integers(1, 9)), cin=int(rng.integers(1, 3)), nconv=1, strides=strides)
        dy, dx = arch.divisors
        patch = (dy * int(rng.integers(max(1, 16 // dy), 96 // dy + 1)), 32 * int(rng.integers(1, 5)))
        shape = (int(rng.integers(1, 3)), int(rng.integers(5, 3 * patch[0])), int(rng.integers(5, 3 * patch[1])))
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
        assert a.dtype == b.dtype == np.float16 and np.array_equal(a, b), (t, feats, strides, shape, patch, step, mirror, folds, order)

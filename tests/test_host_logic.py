"""Host logic of the drop-in predictor on CPU: tiling, mirroring, fp16 Gaussian aggregation, fold ensemble.
The network is injected (torch oracle) into tests/host_predictor.py's subclass of the predictor, so no GPU is needed; the product
class has no such hook and always builds HIP engines (tests/test_gpu_predictor.py)."""
import numpy as np
import pytest

from tests import cases
from tests.conftest import golden, blob_for
from oracle import torch_oracle as O
from totalsegmentator2d_amd import prng, weights
from totalsegmentator2d_amd.predictor import HIPnnUNetPredictor
from tests.host_predictor import HostLogicPredictor


def _predictor(arch, sds, patch, step, mirror, order='float'):
    def net(batch, fold):   # row by row (B = 1) like upstream, so torch picks the same kernels as in the oracle run
        return np.concatenate([O.unet_forward(arch, sds[fold], batch[i:i + 1]).numpy() for i in range(batch.shape[0])])
    p = HostLogicPredictor(network=net, tile_step_size=step, use_mirroring=mirror is not None, tile_dtype=order)
    p.manual_initialization(arch, [weights.pack_blob(arch, sd) for sd in sds], patch,
                            inference_allowed_mirroring_axes=mirror)
    return p


@pytest.mark.parametrize('order', ['float', 'half'])
@pytest.mark.parametrize('name', list(cases.SW_CASES))
def test_predictor_host_logic_equals_the_aten_pinned_oracle_in_both_blend_orders(name, order):
    arch, shape, patch, step, mirror, folds, seed = cases.SW_CASES[name]
    sds = [blob_for(arch, seed + f)[0] for f in range(folds)]
    data = prng.normal_f32(seed, 999, (arch.input_channels,) + tuple(shape))
    p = _predictor(arch, sds, patch, step, mirror, order)
    out = p.predict_logits_from_preprocessed_data(data)
    out = out.cpu().numpy() if hasattr(out, 'cpu') else out
    ref = O.predict_logits(arch, sds, data, patch, step, mirror, tile_dtype=order).numpy()
    assert out.dtype == np.float16 and out.shape == ref.shape == golden(name)['logits_f16'].shape
    # same per-tile network outputs (same torch kernels) => the numpy fp16 emulation must equal torch's half arithmetic
    assert np.array_equal(out, ref)


def test_predictor_duck_type_surface():
    """What reference prediction_worker.py touches."""
    arch = cases.unet(2, (32, 32), 2)
    sd = weights.synthetic_state_dict(arch, 3)
    p = _predictor(arch, [sd], (32, 32), 0.5, (0, 1))
    assert 'nnUNetPredictor' in HIPnnUNetPredictor.__name__ and isinstance(p, HIPnnUNetPredictor)      # prediction_worker.py:103 (p: the tests' subclass of it)
    assert tuple(p.configuration_manager.patch_size) == (32, 32) and len(p.configuration_manager.spacing) == 2
    assert p.dataset_json['file_ending'] == '.nrrd' and len(p.dataset_json['channel_names']) == 2
    assert p.device.type == 'cuda' and p.verbose is False
    assert hasattr(p.configuration_manager.preprocessor_class(verbose=False), 'run_case')
    with pytest.raises(AssertionError):
        p.predict_sliding_window_return_logits(np.zeros((2, 32, 32), np.float32))   # must be 4-D [C,Z,H,W]
    with pytest.raises(RuntimeError):
        HIPnnUNetPredictor(device=type('D', (), {'type': 'cpu', 'index': None})())  # no CPU fallback


def test_pad_and_mirror_combos():
    from totalsegmentator2d_amd import sliding_window as sw
    x = np.arange(2 * 1 * 5 * 7, dtype=np.float32).reshape(2, 1, 5, 7)
    p, rev = sw.pad_nd_image(x, (8, 8))
    assert p.shape == (2, 1, 8, 8) and np.array_equal(p[rev], x) and p[0, 0, 0, 0] == 0
    assert rev[2] == slice(1, 6) and rev[3] == slice(0, 7)                       # below = diff // 2
    assert sw.mirror_combos((0, 1)) == [(), (2,), (3,), (2, 3)] and sw.mirror_combos(None) == [()]
    assert sw.tile_slicers((644, 512), (512, 512), 0.5, 1) == [(0, 0, 0), (0, 132, 0)]


def test_fp16_threshold_and_inf_predicates_match_the_float_definitions_exhaustively():
    """export.py thresholds fp16 logits and predictor.py tests them for inf on the BIT PATTERN; both must equal the float
    definitions (sigmoid(float32(x)) > 0.5, isinf(x)) on every one of the 65536 half-precision values."""
    import torch
    from totalsegmentator2d_amd.export import convert_predicted_logits_to_segmentation_with_correct_shape
    from tests.host_predictor import _any_inf_f16
    v = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    x = v.view(np.float16).reshape(1, 1, 256, 256)
    props = {'shape_before_cropping': (1, 256, 256), 'bbox_used_for_cropping': [[0, 1], [0, 256], [0, 256]]}
    seg = convert_predicted_logits_to_segmentation_with_correct_shape(x, props, True)
    with np.errstate(all='ignore'):
        ref = (torch.sigmoid(torch.from_numpy(x.copy()).float()) > 0.5).numpy().astype(np.uint8)
    assert np.array_equal(seg, ref)
    assert _any_inf_f16(x)
    finite = x.copy(); finite[np.isinf(finite)] = 1.0
    assert not _any_inf_f16(finite)                       # NaNs alone are not "inf"


def test_randomised_window_configurations_match_the_torch_pipeline():
    """Fixed-seed random extents / patch sizes / step sizes / mirror axes / folds / Z: the numpy fp16 aggregation of predictor.py
    must equal torch's half arithmetic in the oracle pipeline bit for bit (same per-tile network outputs)."""
    rng = np.random.default_rng(3)
    for t in range(5):
        ns = int(rng.integers(2, 4)); mult = 2 ** (ns - 1)
        feats = [32] + [int(rng.choice([32, 64])) for _ in range(ns - 1)]
        K = int(rng.integers(1, 6)); cin = int(rng.integers(1, 3))
        patch = (mult * int(rng.integers(4, 12)), mult * int(rng.integers(4, 12)))
        Z = int(rng.choice([1, 1, 2]))
        shape = (Z, int(rng.integers(patch[0] // 2, 3 * patch[0])), int(rng.integers(patch[1] // 2, 3 * patch[1])))
        step = float(rng.choice([0.5, 0.75, 1.0, 0.3])); mirror = [None, (0,), (1,), (0, 1)][int(rng.integers(0, 4))]
        folds = int(rng.choice([1, 2]))
        arch = cases.unet(ns, feats, K, cin=cin)
        sds = [weights.synthetic_state_dict(arch, 700 + 10 * t + f) for f in range(folds)]
        data = prng.normal_f32(800 + t, 999, (cin,) + shape)
        order = ('float', 'half')[t % 2]
        p = _predictor(arch, sds, patch, step, mirror, order)
        out = p.predict_logits_from_preprocessed_data(data)
        out = out.cpu().numpy() if hasattr(out, 'cpu') else out
        ref = O.predict_logits(arch, sds, data, patch, step, mirror, tile_dtype=order).numpy()
        assert out.dtype == np.float16 and np.array_equal(out, ref), (t, feats, shape, patch, step, mirror, folds, order)


def test_nonzero_mask_fills_holes_for_volumes_only():
    """upstream's non-zero mask is binary_fill_holes(OR over channels of data != 0) on [Z, H, W].  With Z = 1 nothing is
    enclosed (the 2-D path: an inner zero stays outside the mask); with Z > 1 an enclosed zero belongs to the mask."""
    from scipy.ndimage import binary_fill_holes
    from totalsegmentator2d_amd import preprocess as P
    vol = np.ones((1, 5, 7, 7), np.float32)
    vol[0, 2, 3, 3] = 0                                                   # a zero enclosed in 3-D
    _, _, m3 = P.crop_to_nonzero(vol, return_mask=True)
    assert m3[2, 3, 3] and np.array_equal(m3, binary_fill_holes(vol[0] != 0))
    flat = np.ones((1, 1, 7, 7), np.float32)
    flat[0, 0, 3, 3] = 0
    _, _, m2 = P.crop_to_nonzero(flat, return_mask=True)
    assert not m2[0, 3, 3] and np.array_equal(m2, binary_fill_holes(flat[0] != 0))      # (upstream's own call leaves it out, too)


def test_normalisation_schemes_of_the_preprocessor():
    """SURVEY row A1, the branches beside the plain z-score: masked ZScoreNormalization (statistics inside the non-zero mask,
    the outside stays 0), CTNormalization (clip to the dataset percentiles, dataset mean / std), Rescale / RGB / No normalisation."""
    from types import SimpleNamespace
    from totalsegmentator2d_amd import preprocess as P
    rng = np.random.default_rng(11)
    img = rng.normal(200, 80, (1, 40, 50)).astype(np.float32)
    img[:, :8] = 0; img[:, :, :5] = 0                                    # a zero frame: nnU-Net crops to the non-zero box first
    img[0, 20, 20] = 0                                                   # ... and a zero INSIDE the box: outside the mask
    data = img[None]                                                     # [C=1, Z=1, H, W]
    pm = SimpleNamespace(transpose_forward=[0, 1, 2], plans={'foreground_intensity_properties_per_channel': {
        '0': {'mean': 180.0, 'std': 70.0, 'percentile_00_5': 20.0, 'percentile_99_5': 400.0}}})
    cm = SimpleNamespace(spacing=(1.5, 1.5), normalization_schemes=['ZScoreNormalization'], use_mask_for_norm=[True])
    out, _, props = P.DefaultPreprocessor(False).run_case_npy(data.copy(), None, {'spacing': (999.0, 1.5, 1.5)}, pm, cm, {})
    assert props['bbox_used_for_cropping'] == [[0, 1], [8, 40], [5, 50]] and out.shape == (1, 1, 32, 45)
    crop = img[:, 8:, 5:]
    m = crop != 0
    ref = np.zeros_like(crop); ref[m] = (crop[m] - crop[m].mean()) / max(crop[m].std(), 1e-8)
    assert np.array_equal(out[0], ref) and out[0, 0, 12, 15] == 0.0        # the inner zero is outside the mask: untouched
    cm.normalization_schemes, cm.use_mask_for_norm = ['CTNormalization'], [False]
    out, _, _ = P.DefaultPreprocessor(False).run_case_npy(data.copy(), None, {'spacing': (999.0, 1.5, 1.5)}, pm, cm, {})
    assert np.allclose(out[0], (np.clip(crop, 20.0, 400.0) - 180.0) / 70.0, atol=1e-6)
    x = rng.integers(0, 256, (30, 20)).astype(np.float32)
    assert np.allclose(P.normalize_channel(x, 'RGBTo01Normalization', False, None, None), x / 255.0)
    r = P.normalize_channel(x, 'RescaleTo01Normalization', False, None, None)
    assert r.min() == 0.0 and abs(r.max() - 1.0) < 1e-6
    assert np.array_equal(P.normalize_channel(x, 'NoNormalization', False, None, None), x)
    with pytest.raises(NotImplementedError):
        P.normalize_channel(x, 'SomethingElse', False, None, None)


def test_resampling_to_the_plan_spacing_and_back():
    """an image whose spacing differs from the plan's is resampled in-plane (order 3, after normalising) to
    round(shape * spacing / target), the logits are resampled back (order 1) before the threshold.  The primitive restates
    skimage.transform.resize(mode='edge', anti_aliasing=False) = scipy.ndimage.zoom(grid_mode=True, mode='nearest') + range clip."""
    from types import SimpleNamespace
    from totalsegmentator2d_amd import preprocess as P, export as E
    # grid_mode zoom by 2, order 1: output sample k sits at input coordinate (k + 0.5) / 2 - 0.5 (edge-clamped)
    ramp = np.arange(8, dtype=np.float32)[None, :].repeat(3, 0)
    up = P.resize_like_skimage(ramp, (3, 16), 1)
    exp = np.clip((np.arange(16) + 0.5) / 2 - 0.5, 0, 7).astype(np.float32)
    assert up.shape == (3, 16) and np.allclose(up[1], exp, atol=1e-6)
    assert np.array_equal(P.resize_like_skimage(ramp, (3, 8), 3), ramp)                       # same shape: untouched
    c = np.full((10, 12), 3.25, np.float32)
    assert np.allclose(P.resize_like_skimage(c, (17, 9), 3), 3.25)                              # constants stay constant
    rng = np.random.default_rng(12)
    z = rng.normal(0, 1, (20, 24)).astype(np.float32)
    big = P.resize_like_skimage(z, (40, 48), 3)
    assert big.min() >= z.min() and big.max() <= z.max()                                       # clip=True: no spline overshoot
    # through the preprocessor and the exporter: 3.0 mm image, 1.5 mm plan
    data = rng.normal(100, 30, (2, 1, 50, 30)).astype(np.float32)
    pm = SimpleNamespace(transpose_forward=[0, 1, 2], transpose_backward=[0, 1, 2], plans={})
    cm = SimpleNamespace(spacing=(1.5, 1.5), normalization_schemes=None, use_mask_for_norm=None)
    out, _, props = P.DefaultPreprocessor(False).run_case_npy(data.copy(), None, {'spacing': (999.0, 3.0, 3.0)}, pm, cm, {})
    assert out.shape == (2, 1, 100, 60) and props['shape_after_cropping_and_before_resampling'] == (1, 50, 30)
    assert np.allclose(out[0, 0], P.resize_like_skimage(P.zscore(data[0])[0], (100, 60), 3))   # normalised FIRST, then resampled
    logits = np.where(out[:1] > 0, 2.0, -2.0).astype(np.float16)                               # [K=1, 1, 100, 60]
    seg = E.convert_predicted_logits_to_segmentation_with_correct_shape(logits, props, True, [0, 1, 2])
    assert seg.shape == (1, 1, 50, 30) and seg.dtype == np.uint8
    back = P.resample_data_to_shape(logits.astype(np.float32), (1, 50, 30), order=1)
    assert np.array_equal(seg, (back > E.SIGMOID_HALF_THRESHOLD).astype(np.uint8))
    with pytest.raises(NotImplementedError):
        P.resample_data_to_shape(np.zeros((1, 2, 8, 8), np.float32), (4, 8, 8))               # slice axis: 3-D configurations only


def test_blend_order_default_follows_the_precision_mode():
    """the fp32-parity modes blend like the reference's CPU path (the one BASELINE.json compares with), the 16-bit mode
    like its CUDA autocast path; an explicit tile_dtype wins."""
    net = lambda x: np.zeros((x.shape[0], 2) + x.shape[2:], np.float32)
    assert HostLogicPredictor(network=net).tile_dtype == 'float'
    assert HostLogicPredictor(network=net, precision='exact').tile_dtype == 'float'
    assert HostLogicPredictor(network=net, precision='f16').tile_dtype == 'half'
    assert HostLogicPredictor(network=net, precision='f16', tile_dtype='float').tile_dtype == 'float'
    with pytest.raises(ValueError):
        HostLogicPredictor(network=net, tile_dtype='double')

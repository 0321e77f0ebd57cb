"""Host logic of the drop-in predictor on CPU: tiling, mirroring, fp16 Gaussian aggregation, fold ensemble.
The network is injected (torch oracle) through the predictor's test hook, so no GPU is needed; the product path
always builds HIP engines (tests/test_gpu_predictor.py)."""
import numpy as np
import pytest

from tests import cases
from tests.conftest import golden, blob_for
from oracle import torch_oracle as O
from totalsegmentator2d_amd import prng, weights
from totalsegmentator2d_amd.predictor import HIPnnUNetPredictor


def _predictor(arch, sds, patch, step, mirror, order='float'):
    def net(batch, fold):   # row by row (B = 1) like upstream, so torch picks the same kernels as in the oracle run
        return np.concatenate([O.unet_forward(arch, sds[fold], batch[i:i + 1]).numpy() for i in range(batch.shape[0])])
    p = HIPnnUNetPredictor(tile_step_size=step, use_mirroring=mirror is not None, network=net, tile_dtype=order)
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
    assert 'nnUNetPredictor' in type(p).__name__                               # prediction_worker.py:103
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
    from totalsegmentator2d_amd.predictor import _any_inf_f16
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

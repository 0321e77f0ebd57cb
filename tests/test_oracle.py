"""The oracles against the golden fixtures (CPU).  Goldens come from oracle/torch_oracle.py run in the build
container (tests/gen_golden.py); the C restatement (oracle/unet_ref.c) is an independent implementation pinned here."""
import hashlib
import os

import numpy as np
import pytest

from tests import cases
from tests.conftest import golden, blob_for, GOLDEN
from oracle import c_oracle as C
from oracle import torch_oracle as O

FAST = ['k_min2', 'k_two3', 'net5_64', 'tiny_b37', 'wide64', 'xr_1ch', 'net5_128', 'aniso_21', 'aniso_12_11']


@pytest.mark.parametrize('name', FAST)
def test_torch_oracle_reproduces_goldens(name):
    arch, B, H, W, seed = cases.SMALL_CASES[name]
    sd, _ = blob_for(arch, seed)
    y = O.unet_forward(arch, sd, cases.make_input(arch, B, H, W, seed)).numpy()
    assert np.abs(y - golden(name)['logits']).max() <= 2e-5       # same ATen kernels; other hosts may pick other ISA paths


@pytest.mark.parametrize('name', FAST)
def test_c_oracle_matches_goldens(name):
    arch, B, H, W, seed = cases.SMALL_CASES[name]
    _, blob = blob_for(arch, seed)
    x = cases.make_input(arch, B, H, W, seed)
    g = golden(name)['logits']
    # bottlenecks of 2x2 pixels make InstanceNorm ill-conditioned (tiny_b37): fp32 implementations legitimately differ more
    tol = 2e-3 if name == 'tiny_b37' else 1e-4
    assert np.abs(C.unet_forward(arch, blob, x, acc64=False) - g).max() <= tol
    assert np.abs(C.unet_forward(arch, blob, x, acc64=True) - g).max() <= tol


def test_anisotropic_strides_follow_plain_torch_modules():
    """Per-axis strides (nnU-Net pools every axis separately; the plan is whatever the model folder holds: reference
    ts2d/core/inference/nnu.py:164-165): the oracle's functional restatement equals a network assembled from plain torch.nn modules
    the way upstream's PlainConvEncoder / UNetDecoder do it - strided first conv of a stage, ConvTranspose2d with kernel = stride =
    the stride of the stage below, cat((up, skip), 1)."""
    import torch
    from torch import nn
    arch, B, H, W, seed = cases.SMALL_CASES['aniso_21']
    sd, _ = blob_for(arch, seed)
    x = torch.from_numpy(cases.make_input(arch, B, H, W, seed))

    def block(cin, cout, stride):
        return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1, bias=True), nn.InstanceNorm2d(cout, eps=arch.norm_eps, affine=True),
                             nn.LeakyReLU(arch.leaky_slope))

    def load(mod, key):
        with torch.no_grad():
            mod[0].weight.copy_(torch.from_numpy(sd[f'{key}.conv.weight'])); mod[0].bias.copy_(torch.from_numpy(sd[f'{key}.conv.bias']))
            mod[1].weight.copy_(torch.from_numpy(sd[f'{key}.norm.weight'])); mod[1].bias.copy_(torch.from_numpy(sd[f'{key}.norm.bias']))
    with torch.no_grad():
        skips, cur, cin = [], x, arch.input_channels
        for s in range(arch.n_stages):
            for i in range(arch.n_conv_per_stage[s]):
                m = block(cin, arch.features_per_stage[s], tuple(arch.strides[s]) if i == 0 else 1)
                load(m, f'encoder.stages.{s}.0.convs.{i}')
                cur, cin = m(cur), arch.features_per_stage[s]
            skips.append(cur)
        for j in range(arch.n_stages - 1):
            lvl = arch.n_stages - 2 - j
            st = tuple(arch.strides[lvl + 1])
            up = nn.ConvTranspose2d(cin, arch.features_per_stage[lvl], st, st, bias=True)
            up.weight.copy_(torch.from_numpy(sd[f'decoder.transpconvs.{j}.weight'])); up.bias.copy_(torch.from_numpy(sd[f'decoder.transpconvs.{j}.bias']))
            cur = torch.cat((up(cur), skips[lvl]), 1)
            cin = 2 * arch.features_per_stage[lvl]
            for i in range(arch.n_conv_per_stage_decoder[j]):
                m = block(cin, arch.features_per_stage[lvl], 1)
                load(m, f'decoder.stages.{j}.convs.{i}')
                cur, cin = m(cur), arch.features_per_stage[lvl]
        k = f'decoder.seg_layers.{arch.n_stages - 2}'
        y = torch.nn.functional.conv2d(cur, torch.from_numpy(sd[f'{k}.weight']), torch.from_numpy(sd[f'{k}.bias'])).numpy()
    assert y.shape == (B, arch.num_classes, H, W)
    assert np.abs(y - golden('aniso_21')['logits']).max() <= 2e-5
    assert np.abs(y - O.unet_forward(arch, sd, x).numpy()).max() <= 2e-5


def test_c_oracle_mask_equals_torch_sigmoid_threshold():
    g = golden('net5_64')['logits']
    assert np.array_equal(C.logits_to_mask(g), O.logits_to_mask(g).numpy())
    assert np.array_equal(O.pack_mask(C.logits_to_mask(g)), golden('net5_64')['mask_packed'])


def test_threshold_predicate_is_pinned():
    """sigmoid(float(x)) > 0.5  <=>  x > 1.5 * 2^-24 on this ATen build: scan every fp32 in [2^-27, 2^-20)."""
    import torch
    lo, hi = np.float32(2.0 ** -27).view(np.uint32), np.float32(2.0 ** -20).view(np.uint32)
    x = np.arange(lo, hi, dtype=np.uint32).view(np.float32)
    ref = (torch.sigmoid(torch.from_numpy(x.copy())) > 0.5).numpy()
    assert np.array_equal(ref, x > np.float32(1.5 * 2.0 ** -24))
    edge = np.array([0.0, -0.0, 1.5 * 2.0 ** -24, np.nextafter(np.float32(1.5 * 2.0 ** -24), np.float32(1)), -1e-3, 1e-3, np.nan, np.inf, -np.inf], dtype=np.float32)
    assert np.array_equal(C.logits_to_mask(edge), (torch.sigmoid(torch.from_numpy(edge)) > 0.5).numpy().astype(np.uint8))


def test_sliding_window_steps_and_gaussian_goldens():
    from totalsegmentator2d_amd import sliding_window as sw
    g = golden('sliding_window')
    for k in g.files:
        if k.startswith('steps/'):
            img, st = k[6:].split('_')
            assert sw.compute_steps_for_sliding_window((max(int(img), 512),), (512,), float(st))[0] == g[k].tolist()
    assert sw.compute_steps_for_sliding_window((644, 512), (512, 512), 0.5) == [[0, 132], [0]]   # config 1: 2 tiles
    m = sw.compute_gaussian((512, 512))
    assert m.dtype == np.float16
    assert np.array_equal(np.diag(m), g['g512_diag']) and np.array_equal(m[0], g['g512_row0']) and np.array_equal(m[256], g['g512_centre'])
    assert np.array_equal(sw.compute_gaussian((64, 96)), g['g64x96'])
    assert np.array_equal(sw.compute_gaussian((64, 96)), O.compute_gaussian((64, 96)).numpy())     # scipy-based oracle


def test_zscore_of_reference_sample():
    from totalsegmentator2d_amd import nrrd, preprocess
    g = golden('sample_s0616_zscore')
    img = nrrd.read(os.path.join(GOLDEN, 'assets', 'sample_s0616.nrrd'))
    assert np.array_equal(np.frombuffer(hashlib.sha256(np.ascontiguousarray(img.array).tobytes()).digest(), dtype=np.uint8), g['raw_sha256'])
    data, _, props = preprocess.DefaultPreprocessor().run_case(
        [os.path.join(GOLDEN, 'assets', 'sample_s0616.nrrd')], None,
        type('P', (), {'transpose_forward': [0, 1, 2]})(), type('Cfg', (), {'spacing': [1.5, 1.5], 'normalization_schemes': None})(), {})
    assert data.shape == (2, 1, 644, 337) and data.dtype == np.float32           # [2,1,644,337]
    assert np.allclose(data[:, 0, ::16, ::16], g['samples'], atol=1e-6)
    assert abs(float(data.astype(np.float64).sum()) - float(g['sum'])) < 1e-2
    assert props['bbox_used_for_cropping'] == [[0, 1], [0, 644], [0, 337]]


def test_blend_order_is_pinned_by_plain_aten_statements():
    """upstream's ``prediction *= gaussian; predicted_logits[sl] += prediction`` with float16
    ``predicted_logits`` / ``gaussian``.  On the reference's CPU path the tile prediction is fp32 (``.to(results_device)`` moves
    device, it does not cast), so ATen multiplies in fp32 and rounds ONCE into the half buffer; under CUDA autocast the tile
    is half and product and sum each round.  The statements below are upstream's, on hand-made two-tile overlaps; the numpy
    arithmetic of predictor.py / kernels_sw.h (restated here) must reproduce both orders bit for bit."""
    import torch
    rng = np.random.default_rng(5)
    K, H, W, ph = 3, 48, 32, 32
    g16 = O.compute_gaussian((ph, W))
    tiles = [(0, torch.from_numpy(rng.normal(0, 3, (K, ph, W)).astype(np.float32))),
             (16, torch.from_numpy(rng.normal(0, 3, (K, ph, W)).astype(np.float32)))]
    outs = {}
    for order in ('float', 'half'):
        logits = torch.zeros((K, H, W), dtype=torch.half)
        n_pred = torch.zeros((H, W), dtype=torch.half)
        for y0, p32 in tiles:
            prediction = p32.clone() if order == 'float' else p32.to(torch.half)
            prediction *= g16
            logits[:, y0:y0 + ph] += prediction
            n_pred[y0:y0 + ph] += g16
        logits /= n_pred
        outs[order] = logits.numpy()
        # the arithmetic the product uses (predictor.py host path == sw_aggregate)
        acc = np.zeros((K, H, W), np.float16); n = np.zeros((H, W), np.float16); g = g16.numpy()
        for y0, p32 in tiles:
            p = p32.numpy()
            if order == 'half':
                acc[:, y0:y0 + ph] += p.astype(np.float16) * g
            else:
                acc[:, y0:y0 + ph] = (acc[:, y0:y0 + ph].astype(np.float32) + p * g.astype(np.float32)).astype(np.float16)
            n[y0:y0 + ph] += g
        assert np.array_equal(acc / n, outs[order]), order
    assert (outs['float'] != outs['half']).any()          # the two orders are observably different


@pytest.mark.parametrize('order', ['float', 'half'])
@pytest.mark.parametrize('name', list(cases.SW_CASES))
def test_torch_oracle_sliding_window_goldens(name, order):
    from tests.conftest import blob_for
    from totalsegmentator2d_amd import prng
    arch, shape, patch, step, mirror, folds, seed = cases.SW_CASES[name]
    sds = [blob_for(arch, seed + f)[0] for f in range(folds)]
    data = prng.normal_f32(seed, 999, (arch.input_channels,) + tuple(shape))
    out = O.predict_logits(arch, sds, data, patch, step, mirror, tile_dtype=order).numpy()
    g = golden(name)['logits_f16' if order == 'float' else 'logits_f16_half']
    assert out.dtype == np.float16 and out.shape == g.shape
    assert np.abs(out.astype(np.float32) - g.astype(np.float32)).max() <= 8e-3   # <= a couple of fp16 ulps at |x| ~ 4


def test_mean_projection_is_pinned_by_the_reference_assets():
    """The reference's pre-projected sample inputs ARE outputs of ``TS2D._project`` (tool.py:152-160,182-185): channel 1 (maximum
    of an integer CT) is integer-valued, channel 0 (mean) is exactly double(S) / n for an integer S - n = 269 coronal slices
    (403.5 mm / 1.5 mm, the collapsed axis of sample_s0332) resp. 261 (sample_s0616, stored before the Float32 cast).  This pins
    oracle/input_oracle.py:project (real-valued mean) - and rules out a mean truncated to the integer input type."""
    from oracle import input_oracle as IO
    from totalsegmentator2d_amd import nrrd
    a = nrrd.read(os.path.join(GOLDEN, 'assets', 'sample_s0332.nrrd'))
    assert a.array.dtype == np.float32 and round(a.spacing[1] / 1.5) == 269
    c0, c1 = a.array[..., 0].ravel(), a.array[..., 1].ravel()
    assert np.array_equal(c1, np.round(c1)) and np.mean(c0 != np.round(c0)) > 0.99
    S = np.round(c0.astype(np.float64) * 269)
    assert np.array_equal((S / 269.0).astype(np.float32), c0)                       # float32(double(S) / 269), every pixel
    b = nrrd.read(os.path.join(GOLDEN, 'assets', 'sample_s0616.nrrd'))
    assert b.array.dtype == np.float64
    d0, d1 = b.array[..., 0].ravel(), b.array[..., 1].ravel()
    assert np.array_equal(d1, np.round(d1)) and np.mean(d0 != np.round(d0)) > 0.99
    S = np.round(d0 * 261)
    assert np.array_equal(S / 261.0, d0)                                            # double(S) / 261, every pixel
    # the oracle reproduces that arithmetic on an integer volume with the same slice count: a (z, y, x) stack of 261 slices
    rng = np.random.default_rng(7)
    vol = rng.integers(-1024, 3000, (6, 261, 5)).astype(np.int16)
    m = IO.project(vol, 'mean', 1)
    assert m.dtype == np.float64 and np.array_equal(m, vol.sum(axis=1, dtype=np.int64) / 261.0)
    assert np.array_equal(IO.project_f32(vol, 'mean', 1), (vol.sum(axis=1, dtype=np.int64) / 261.0).astype(np.float32))
    assert np.array_equal(IO.project(vol, 'max', 1), vol.max(axis=1))


def test_host_projection_equals_input_oracle():
    """Product host code (image.reorient_image + image.project + Float32 cast) against the independent restatement in oracle/,
    on the reference's 3-D sample (int16, flipped x / y) and on float / uint8 volumes with permuted axes."""
    from oracle import input_oracle as IO
    from totalsegmentator2d_amd import image, nrrd
    rng = np.random.default_rng(3)
    vols = [nrrd.read(os.path.join(GOLDEN, 'assets', 'sample_s0521.nrrd')),
            nrrd.Image(rng.normal(0, 300, (40, 33, 50)).astype(np.float32), (1.0, 2.0, 3.0), (5.0, -7.0, 11.0),
                       (0.0, -1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, -1.0), 1, {}, 'left-posterior-superior'),
            nrrd.Image(rng.integers(0, 255, (20, 16, 24)).astype(np.uint8), (1.0, 1.0, 1.0), (0.0, 0.0, 0.0),
                       (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0), 1, {}, None)]
    for vol in vols:
        ref = IO.coronal_projections_f32(vol.array, vol.direction)
        r = image.reorient_image(vol)
        for mode in ('max', 'mean'):
            got = image.cast(image.project(r, mode, 'coronal'), np.float32).array[:, 0, :]
            assert got.dtype == np.float32 and np.array_equal(got, ref[mode]), mode
    x = rng.normal(-300, 250, (64, 48)).astype(np.float32)
    from totalsegmentator2d_amd import preprocess
    assert np.array_equal(preprocess.zscore(x), IO.zscore(x))


def test_preprocessor_properties_carry_the_sitk_geometry():
    """``properties['sitk_stuff']`` holds spacing / origin / direction in SimpleITK order - what upstream's SimpleITK writer reads
    when the reference's unchanged ``export_prediction_from_logits`` (prediction_worker.py:215-221) consumes these properties."""
    from totalsegmentator2d_amd import nrrd, preprocess
    fp = os.path.join(GOLDEN, 'assets', 'sample_s0616.nrrd')
    img = nrrd.read(fp)
    _, props = preprocess.read_images([fp])
    st = props['sitk_stuff']
    assert set(st) >= {'spacing', 'origin', 'direction', 'files'} and st['files'] == [fp]
    assert st['spacing'] == tuple(img.spacing) and len(st['spacing']) == 2 and len(st['origin']) == 2 and len(st['direction']) == 4
    assert props['spacing'] == (999.0, img.spacing[1], img.spacing[0])            # nnU-Net order beside it
    _, p2 = preprocess.image_to_array(img)
    assert p2['sitk_stuff']['direction'] == tuple(img.direction) and p2['sitk_stuff']['files'] == []

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

FAST = ['k_min2', 'k_two3', 'net5_64', 'tiny_b37', 'wide64', 'xr_1ch', 'net5_128']


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

"""The split mode (fp16 hi/lo products) under adversarial weights: large InstanceNorm gains, weights spanning 20 binades inside
one layer, and an un-normalised transposed-conv output driven past the fp16 range - which must surface as an error naming the
layer, never as a silent inf."""
import numpy as np
import pytest

from tests import cases
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _large_batch_kernels_at_small_b():
    """The tests of this module address the kernels a full batch runs on (persistent / composed / 512-thread) by driving them with one
    to three slices; round 6's small-batch dispatch ("sbk": split-K and the two-kernel decoder entry where those kernels would leave
    most CUs idle) would take such batches elsewhere.  It is switched off here and has its own module (tests/test_gpu_small_batch.py)."""
    from totalsegmentator2d_amd.engine import Engine as _E
    old = dict(_E.default_options)
    _E.default_options = {**old, 'sbk': 0}
    yield
    _E.default_options = old


def _case():
    arch = cases.unet(4, (32, 64, 128, 128), 5)
    sd = weights.synthetic_state_dict(arch, 61)
    x = cases.make_input(arch, 2, 64, 64, 61)
    return arch, sd, x


def _rel_err(lg, ref):
    return float(np.abs(lg - ref).max() / max(1.0, float(np.abs(ref).max())))


def test_large_instance_norm_gains():
    arch, sd, x = _case()
    sd = {k: (v * np.float32(100.0) if k.endswith('norm.weight') else v) for k, v in sd.items()}      # |gamma| ~ 100
    ref = O.unet_forward(arch, sd, x).numpy()
    with Engine(arch, weights.pack_blob(arch, sd)) as e:
        lg, _ = e.forward(x)
    assert np.isfinite(lg).all() and _rel_err(lg, ref) <= 1e-4


def test_weights_spanning_twenty_binades_in_one_layer():
    arch, sd, x = _case()
    rng = np.random.default_rng(0)
    sd = dict(sd)
    for k in list(sd):
        if k.endswith('conv.weight') and sd[k].ndim == 4 and sd[k].shape[1] >= 32:
            w = sd[k].copy()
            scale = np.exp2(-rng.integers(0, 21, size=w.shape)).astype(np.float32)      # 2^0 .. 2^-20, element by element
            sd[k] = w * scale        # the per-layer power-of-two pre-scale is set by the largest weight: small ones lose their lo part
    ref = O.unet_forward(arch, sd, x).numpy()
    with Engine(arch, weights.pack_blob(arch, sd)) as e:
        lg, _ = e.forward(x)
    assert np.isfinite(lg).all() and _rel_err(lg, ref) <= 1e-4


def _wide_case():
    """2 x 128 x 128 through a 5-stage net: every statistics-producing kernel family runs (first layer, resident 32 -> 32, the
    persistent q pipeline at levels 1-2, 512-thread stride 2, composed decoder entries incl. the upq form, generic kernels and
    split-K at the deep levels)."""
    arch = cases.unet(5, (32, 64, 128, 256, 512), 6)
    sd = weights.synthetic_state_dict(arch, 62)
    x = cases.make_input(arch, 2, 128, 128, 62)
    return arch, sd, x


@pytest.mark.parametrize('gain', [100.0, 1000.0])
def test_large_conv_biases_are_dead_under_instance_norm(gain):
    """a conv bias is removed again by the InstanceNorm behind it, so blowing the biases up by 100x / 1000x
    (|mean| / sigma of the raw conv output ~ 1 ... 30) barely moves the oracle's logits - and must not move the engine's: the
    per-tile statistics are SHIFTED sums around a pivot taken from the tile (kernels.h), not sum(v) / sum(v^2) of the biased values."""
    arch, sd, x = _wide_case()
    big = {k: (v * np.float32(gain) if k.endswith('conv.bias') else v) for k, v in sd.items()}
    ref = O.unet_forward(arch, big, x).numpy()
    base = O.unet_forward(arch, sd, x).numpy()
    assert np.abs(ref - base).max() <= 1e-3 * max(1.0, gain / 100.0)          # the oracle itself: the bias is (almost) dead
    with Engine(arch, weights.pack_blob(arch, big)) as e:
        for mode in ('split', 'exact'):
            e.set_precision(mode)
            lg, _ = e.forward(x)
            assert np.isfinite(lg).all() and _rel_err(lg, ref) <= 1e-4, (mode, gain, _rel_err(lg, ref))


def test_input_offset_puts_the_mean_far_above_sigma_at_the_first_layer():
    """A network input with an offset of 50 sigma gives the first conv's raw output |mean| / sigma of up to ~20 per channel (the zero-padded border bounds it) (the
    same happens to any layer whose input statistics drift): torch normalises with mean first, then sum((x - mean)^2).  The
    normalised activation of enc0.c0 and the logits must follow the oracle."""
    arch, sd, x = _wide_case()
    xo = (x + np.float32(50.0)).astype(np.float32)
    ref, inter = O.unet_forward(arch, sd, xo, return_intermediates=True)
    raw = O.F.conv2d(O._t(xo), O._t(sd['encoder.stages.0.0.convs.0.conv.weight']), O._t(sd['encoder.stages.0.0.convs.0.conv.bias']), padding=1).numpy()
    ratio = np.abs(raw.mean(axis=(2, 3))) / raw.std(axis=(2, 3))
    assert ratio.max() >= 15.0                                                 # the regime this test is about
    for opts in ({}, {'fuse0': 0}):              # default: enc0.c0 is a statistics-only pass, recomputed inside enc0.c1; and as its own kernel
        with Engine(arch, weights.pack_blob(arch, sd), options=opts) as e:
            for mode in ('split', 'exact'):
                e.set_precision(mode)
                lg, _ = e.forward(xo)
                first = 'enc0.c0' if e.materialised('enc0.c0') else 'enc0.c1'
                assert (first == 'enc0.c0') == (mode == 'exact' or 'fuse0' in opts)
                a0 = e.debug_tensor(first)
                assert np.abs(a0 - inter[first].numpy()).max() <= 3e-5, (mode, first)
                assert np.isfinite(lg).all() and _rel_err(lg, ref.numpy()) <= 1e-4, (mode, opts, _rel_err(lg, ref.numpy()))


def test_overflowing_transposed_conv_output_is_an_error_naming_the_layer():
    arch, sd, x = _case()
    blob_ok = weights.pack_blob(arch, sd)
    sd = dict(sd)
    key = 'decoder.transpconvs.1.weight'
    sd[key] = sd[key] * np.float32(3e5)                    # un-normalised output of dec1.up far beyond 65504
    blob = weights.pack_blob(arch, sd)
    ref = O.unet_forward(arch, sd, x).numpy()
    with Engine(arch, blob) as e:
        # default path: the transposed conv is COMPOSED into dec1.c0 (kernels_upc.h) - its un-normalised output never exists as
        # an fp16 operand, the composed weights carry their own power-of-two pre-scale: finite and right
        lg, _ = e.forward(x)                               # (level 1 of this case is 32 x 32: complete tiles, composed)
        assert not e.materialised('dec1.up')
        assert np.isfinite(lg).all() and _rel_err(lg, ref) <= 1e-4
    with Engine(arch, blob, options={'upc': 0}) as e:      # two-kernel path: the overflow is detected and named
        with pytest.raises(RuntimeError, match=r'non-finite logits: inf / NaN first appears in layer dec1\.c0'):
            e.forward(x)                                   # host-buffer forward runs ts2d_engine_check itself
        import torch
        e.forward(torch.from_numpy(x).cuda())              # asynchronous device-pointer call: no error yet ...
        with pytest.raises(RuntimeError, match='non-finite'):
            e.check()                                      # ... until the caller asks
        e.set_precision('exact')                           # the fp32 MFMA path has no fp16 range limit
        lg, _ = e.forward(x)
        assert np.isfinite(lg).all()
        assert _rel_err(lg, ref) <= 1e-4
        e.set_precision('split')
        e.load_weights(blob_ok)                            # the flag does not stick: sane weights, same engine
        good, _ = e.forward(x)
        assert np.isfinite(good).all()


def test_two_engines_on_two_devices_of_one_process():
    """Handles are independent (include/ts2d_engine.h): two engines on two devices of ONE process, interleaved calls, identical
    results.  Exercises the per-device state of `allow_max_lds` (hipFuncAttributeMaxDynamicSharedMemorySize is a per-device
    property of a kernel; ADVICE r1: the "already set" flags used to be per process).  Needs two visible GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two visible GPUs (the GPU box of this round has one)')
    from totalsegmentator2d_amd.arch import UNetArch
    arch = UNetArch.canonical(num_classes=5)
    sd = weights.synthetic_state_dict(arch, 77)
    blob = weights.pack_blob(arch, sd)
    x = cases.make_input(arch, 2, 256, 256, 77)
    with Engine(arch, blob, device=1) as e1:                   # device 1 FIRST: its kernels need the attribute before device 0 ever ran
        l1, m1 = e1.forward(x, logits=True, mask=True)
        with Engine(arch, blob, device=0) as e0:
            l0, m0 = e0.forward(x, logits=True, mask=True)
            l1b, _ = e1.forward(x, logits=True)
            for mode in ('exact', 'f16'):
                e0.set_precision(mode); e1.set_precision(mode)
                a, _ = e0.forward(x, logits=True); b, _ = e1.forward(x, logits=True)
                assert np.array_equal(a, b), mode
    assert np.array_equal(l0, l1) and np.array_equal(l1, l1b) and np.array_equal(m0, m1)


@pytest.mark.parametrize('cin', [1, 2])
def test_split_first_block_and_its_exact_fallback(cin):
    """Round 6: the first block's K = 9 C contraction runs as ONE fp16 hi / lo split product (conv3x3_first_split: the input as
    x / 4 = hi + lo, 22 bits) in the split and the 16-bit mode.  (a) Both sides of the "first_split" switch against the oracle's
    enc0.c0, as its own kernel (fuse0 = 0, STORE form) and as the statistics-only pass in front of the fused second block.
    (b) The precondition |x| < 262 016 must only cost speed: a tile whose patch holds a larger value - or one just below the
    limit - is computed with the exact fp32 MFMAs (in-kernel, per tile) and the result follows the oracle all the same; the
    reference feeds whatever float32 array its preprocessor produced (ts2d/core/inference/prediction_worker.py:206)."""
    arch = cases.unet(3, (32, 64, 64), 4, cin=cin)
    sd = weights.synthetic_state_dict(arch, 71)
    blob = weights.pack_blob(arch, sd)
    x = cases.make_input(arch, 2, 64, 96, 71)
    xs = {'plain': x}
    xb = x.copy()
    xb[0, 0, 5, 7] = 3.0e5                      # beyond the split range: tile (0, 0) of image 0 falls back
    xb[1, cin - 1, 40, 70] = -2.6e5             # just inside it (|x| / 4 = 65 000 < 65 504): stays on the split path
    xb[1, 0, 63, 95] = 7.0e8                    # last pixel of the last tile
    xs['outliers'] = xb
    for tag, xin in xs.items():
        ref, inter = O.unet_forward(arch, sd, xin, return_intermediates=True)
        ref = ref.numpy()
        want0 = inter['enc0.c0'].numpy()
        got = {}
        for fs in (1, 0):
            with Engine(arch, blob, options={'first_split': fs, 'fuse0': 0}) as e:
                e.set_profiling(True)
                lg, _ = e.forward(xin)
                assert (e.op_kernels()['enc0.c0'] == 'conv3x3_first_split') == bool(fs)
                a0 = e.debug_tensor('enc0.c0')
                assert np.isfinite(a0).all() and np.abs(a0 - want0).max() <= 3e-5, (tag, fs, float(np.abs(a0 - want0).max()))
                assert _rel_err(lg, ref) <= 1e-4, (tag, fs, _rel_err(lg, ref))
                got[fs] = a0
                if tag != 'plain':               # (the raw conv output of such an input does not fit fp16 STORAGE: the 16-bit mode is not for it)
                    continue
                e.set_precision('f16')           # 16-bit mode: the same kernel with fp16 stores; enc0.c0 against ONE block of the 16-bit oracle
                e.forward(xin)
                assert (e.op_kernels()['enc0.c0'] == 'conv3x3_first_split') == bool(fs)
                h0 = e.debug_tensor('enc0.c0')
                w16 = O.layer_forward(arch, sd, 'enc0.c0', xin, emulate='f16', storage_view=True).numpy()
                d = h0 - w16
                assert np.isfinite(h0).all() and np.abs(d).max() <= 1e-2 and np.sqrt((d ** 2).mean()) <= 1e-4, (tag, fs, float(np.abs(d).max()))
        assert np.abs(got[1] - got[0]).max() <= 2e-5
        with Engine(arch, blob) as e:            # default: statistics-only pass (split product) + the recompute inside enc0.c1
            e.set_profiling(True)
            lg, _ = e.forward(xin)
            if e.op_kernels().get('enc0.c0') == 'conv3x3_first_stats':
                a1 = e.debug_tensor('enc0.c1')
                assert np.abs(a1 - inter['enc0.c1'].numpy()).max() <= 3e-5, tag
            assert _rel_err(lg, ref) <= 1e-4, (tag, _rel_err(lg, ref))

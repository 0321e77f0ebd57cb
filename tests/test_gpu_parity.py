"""Parity tests proper (run with -m gpu on an MI355X): the HIP engine, called through the C-ABI, against the golden
fixtures and the oracles.  Tolerance: north_star states 1e-4 max-abs on fp32 logits; masks bit-exact."""
import hashlib

import numpy as np
import pytest

from tests import cases
from tests.conftest import golden, blob_for
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd.engine import Engine, unpack_mask

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
TOL = 1e-4


THR = 1.5 * 2.0 ** -24          # sigmoid(float(x)) > 0.5  <=>  x > 1.5 * 2^-24 (tests/test_oracle.py pins the predicate)


def _oracle_mask(logits):
    from oracle import c_oracle as C
    return C.logits_to_mask(logits)


def assert_flips_are_tolerance_flips(lg, ref, mask_u8, ref_mask_u8):
    """Mask bits may differ from the oracle's end to end ONLY where the oracle's logit lies within the slice's logit error of the
    threshold: a flipped bit at |oracle logit| = 0.3 is a bug, not a tolerance flip.  Returns (#flips, the
    largest |oracle logit - threshold| among them)."""
    lg, ref = np.asarray(lg), np.asarray(ref)
    diff = np.asarray(mask_u8) != np.asarray(ref_mask_u8)
    worst = 0.0
    for b in range(lg.shape[0]):
        if diff[b].any():
            err = float(np.abs(lg[b] - ref[b]).max())
            dist = float(np.abs(ref[b][diff[b]].astype(np.float64) - THR).max())
            assert dist <= err, f'slice {b}: a mask bit differs from the oracle at |oracle logit - thr| = {dist:.3e} > logit error {err:.3e}'
            worst = max(worst, dist)
    return int(diff.sum()), worst


@pytest.mark.parametrize('name', list(cases.SMALL_CASES))
def test_logits_and_masks_match_goldens(name):
    arch, B, H, W, seed = cases.SMALL_CASES[name]
    _, blob = blob_for(arch, seed)
    x = cases.make_input(arch, B, H, W, seed)
    g = golden(name)
    with Engine(arch, blob) as e:
        lg, mk = e.forward(x, logits=True, mask=(W % 32 == 0))
        if name == 'tiny_b37':
            # 2x2-pixel bottleneck: InstanceNorm over 4 values is ill-conditioned; judge against the fp64-accumulating
            # C oracle and require the GPU to be no worse than the torch-fp32 golden is
            from oracle import c_oracle as C
            truth = C.unet_forward(arch, blob, x, acc64=True)
            assert np.abs(lg - truth).max() <= max(2.0 * np.abs(g['logits'] - truth).max(), TOL)
        else:
            assert np.abs(lg - g['logits']).max() <= TOL
        if mk is not None:
            assert np.array_equal(unpack_mask(mk, W), _oracle_mask(lg))            # bit-exact on the same logits
            disagree = int((mk.view(np.uint32) != g['mask_packed']).sum())
            assert disagree <= max(2, lg.size // 20000)                              # vs oracle end-to-end: only |logit| ~ 1e-5 flips
            if name != 'tiny_b37':                                                   # ... and every one of them within the logit error of the threshold
                assert_flips_are_tolerance_flips(lg, g['logits'], unpack_mask(mk, W), unpack_mask(g['mask_packed'], W))
        inter = [k for k in g.files if k.startswith('inter/')]
        # K5 composed into K6 (kernels_upc.h); the first block recomputed inside the second (conv3x3_res32<.., FUSE>): not materialised
        composed = [k for k in inter if (k.endswith('.up') or k == 'inter/enc0.c0') and not e.materialised(k[6:])]
        for k in inter:                                                               # per-kernel parity K1..K7
            if k not in composed:
                t = e.debug_tensor(k[6:])
                assert t.shape == g[k].shape and np.abs(t - g[k]).max() <= TOL, k
    if composed:                                                                      # K5 / the first block on their own kernels
        with Engine(arch, blob, options={'upc': 0, 'fuse0': 0}) as e:
            lg0, _ = e.forward(x, logits=True)
            for k in inter:
                t = e.debug_tensor(k[6:])
                assert t.shape == g[k].shape and np.abs(t - g[k]).max() <= TOL, k
            assert np.abs(lg0 - lg).max() <= 3e-5


def test_canonical_net_512_golden():
    arch = UNetArch.canonical()
    _, blob = blob_for(arch, 1)
    x = cases.make_input(arch, 1, 512, 512, 1)
    g = golden('canonical_512')
    with Engine(arch, blob) as e:
        lg, mk = e.forward(x, logits=True, mask=True)
    assert np.abs(lg[:, :, ::8, ::8] - g['samples']).max() <= TOL
    assert abs(float(lg.astype(np.float64).sum()) - float(g['sum'])) <= 1e-5 * float(g['abs_sum'])
    assert abs(float(np.abs(lg.astype(np.float64)).sum()) - float(g['abs_sum'])) <= 1e-5 * float(g['abs_sum'])
    assert abs(float(lg.max()) - float(g['vmax'])) <= TOL and abs(float(lg.min()) - float(g['vmin'])) <= TOL
    assert np.array_equal(unpack_mask(mk, 512), _oracle_mask(lg))
    assert abs(int(unpack_mask(mk, 512).sum()) - int(g['mask_count'])) <= 64
    assert int((mk.view(np.uint32)[:, :, ::8] != g['mask_rows']).sum()) <= 8
    if hashlib.sha256(mk.view(np.uint32).tobytes()).digest() != g['mask_sha256'].tobytes():
        # expected: a handful of pixels with |logit| < 1e-4 fall on the other side; report, do not fail
        print('note: packed-mask hash differs from the oracle end-to-end hash (logits within tolerance)')


def test_full_batch_properties_config2():
    """BASELINE config 2 size (B=64, 2x512x512): size-independent properties."""
    import torch
    arch = UNetArch.canonical()
    _, blob = blob_for(arch, 1)
    gen = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(64, 2, 512, 512, device='cuda', generator=gen)
    with Engine(arch, blob) as e:
        lg, mk = e.forward(x, logits=True, mask=True)
        torch.cuda.synchronize()
        lg2, _ = e.forward(x, logits=True, mask=False)
        torch.cuda.synchronize()
        assert torch.equal(lg, lg2)                                                   # deterministic (no float atomics)
        assert bool(torch.isfinite(lg).all())
        # batch independence: slice i alone == slice i inside the batch of 64 (per-sample statistics, no cross-slice state)
        for i in (0, 37, 63):
            li, _ = e.forward(x[i:i + 1].contiguous(), logits=True)
            torch.cuda.synchronize()
            assert torch.equal(li[0], lg[i]), f'slice {i} depends on its batch'
        # mask == predicate(logits), checked on device for the whole batch
        bits = (lg > 1.5 * 2.0 ** -24).to(torch.int64).reshape(64, 18, 512, 16, 32)
        packed = (bits << torch.arange(32, device='cuda')).sum(-1)
        assert torch.equal(packed, mk.to(torch.int64) & 0xFFFFFFFF)
        # InstanceNorm property on an intermediate: mean 0 / var 1 before the affine is hard to read after LeakyReLU,
        # so check the statistics kernel instead: enc0.c1 activations must be finite and O(1)
        x2 = x[:2].contiguous()                               # (the engine holds its last input weakly: keep it alive for debug_tensor)
        li, _ = e.forward(x2, logits=True)
        t = e.debug_tensor('enc0.c1')
        assert np.isfinite(t).all() and 0.2 < float(np.abs(t).mean()) < 2.0
    # spot parity of one slice of the big batch against the torch oracle
    from oracle import torch_oracle as O
    sd, _ = blob_for(arch, 1)
    ref = O.unet_forward(arch, sd, x[37:38].cpu().numpy()).numpy()
    assert np.abs(lg[37:38].cpu().numpy() - ref).max() <= TOL
    nflip, _ = assert_flips_are_tolerance_flips(lg[37:38].cpu().numpy(), ref, unpack_mask(mk[37:38].cpu().numpy(), 512), O.logits_to_mask(ref).numpy())
    assert nflip <= 64


def test_error_paths_raise_runtime_error():
    arch = cases.unet(3, (32, 32, 64), 2)
    _, blob = blob_for(arch, 5)
    with pytest.raises(RuntimeError, match='weight blob'):
        Engine(arch, blob[:-1])
    with Engine(arch, blob) as e:
        with pytest.raises(RuntimeError, match='multiples of 4'):
            e.forward(np.zeros((1, 2, 30, 32), np.float32))
        with pytest.raises(RuntimeError, match='channels'):
            e.forward(np.zeros((1, 3, 32, 32), np.float32))
        with pytest.raises(RuntimeError, match='more than 1 spatial element'):
            e.forward(np.zeros((1, 2, 4, 4), np.float32))
        lg, _ = e.forward(np.zeros((1, 2, 32, 32), np.float32))                       # warm-up contract: zero patch works
        assert np.isfinite(lg).all()
    with Engine(arch, None) as e:                                                     # replica before the broadcast
        with pytest.raises(RuntimeError, match='weights not loaded'):
            e.forward(np.zeros((1, 2, 32, 32), np.float32))


def test_weight_broadcast_hook_single_rank():
    """Replica created without weights + device-to-device copy into its arena == engine created with weights."""
    import torch
    arch = cases.unet(3, (32, 32, 64), 2)
    _, blob = blob_for(arch, 5)
    x = cases.make_input(arch, 2, 32, 64, 5)
    with Engine(arch, blob) as src, Engine(arch, None) as dst:
        from totalsegmentator2d_amd.parallel import _DevicePtrTensor
        ps, ns = src.weight_buffer()
        pd, nd = dst.weight_buffer()
        assert ns == nd and ns > 0
        ts = torch.as_tensor(_DevicePtrTensor(ps, ns), device='cuda')
        td = torch.as_tensor(_DevicePtrTensor(pd, nd), device='cuda')
        td.copy_(ts)
        torch.cuda.synchronize()
        dst.weights_ready()
        a, _ = src.forward(x)
        b, _ = dst.forward(x)
        assert np.array_equal(a, b)
        src.load_weights(blob * 0.5)                                                  # fold switch
        c, _ = src.forward(x)
        assert not np.array_equal(a, c)


@pytest.mark.parametrize('K,seed', [(23, 2), (26, 4)])
def test_config3_other_submodel_heads(K, seed):
    """BASELINE config 3 geometry: the other ts2d-v2 sub-models differ only in the number of heads (18/23/24/26/26)."""
    from oracle import torch_oracle as O
    arch = UNetArch.canonical(num_classes=K)
    sd, blob = blob_for(arch, seed)
    x = cases.make_input(arch, 1, 512, 512, seed)
    with Engine(arch, blob) as e:
        lg, mk = e.forward(x, logits=True, mask=True)
    ref = O.unet_forward(arch, sd, x).numpy()
    assert lg.shape == (1, K, 512, 512) and np.abs(lg - ref).max() <= TOL
    assert np.array_equal(unpack_mask(mk, 512), _oracle_mask(lg))


def test_config5_tsxr_geometry_9_stages_1024():
    """BASELINE config 5 geometry (tsxr: 1-channel 1024x1024, 9 stages, K=26), fp32 path."""
    from oracle import torch_oracle as O
    arch = UNetArch.canonical(input_channels=1, num_classes=26, n_stages=9)
    sd, blob = blob_for(arch, 7)
    x = cases.make_input(arch, 1, 1024, 1024, 7)
    with Engine(arch, blob) as e:
        lg, mk = e.forward(x, logits=True, mask=True)
        for mode in ('exact',):
            e.set_precision(mode)
            lg2, _ = e.forward(x, logits=True)
    ref = O.unet_forward(arch, sd, x).numpy()
    assert np.abs(lg - ref).max() <= TOL and np.abs(lg2 - ref).max() <= TOL
    assert np.array_equal(unpack_mask(mk, 1024), _oracle_mask(lg))


# ----------------------------------------------------------------------------------------------------------------------
# "mixed fp16" mode (BASELINE configs 3 and 5: 16-bit activations/weights, fp32 accumulate, fp32 InstanceNorm statistics).
# Two references, two bounds, both stated:
#  (a) the 16-BIT oracle (oracle/torch_oracle.py: unet_forward(emulate='f16') - the mode's own arithmetic contract: fp16 weights, fp16
#      stored activations, fp32 accumulation / statistics).  What is left between it and the engine is fp32 summation order, which can
#      flip the fp16 rounding of single stored values, and the composed decoder entries (fp64-composed weights rounded ONCE to fp16
#      where the oracle rounds the transposed conv's weights, its output and the 3x3 weights separately).  Measured on the MI355X
#      (scripts/gpu_f16_oracle_err.py, numbers in DESIGN.md section 4): the bounds below are ~2x the worst case seen.
#  (b) the fp32 oracle: how far the MODE is from fp32 (canonical net 7e-2 max / 8e-3 rms, 0.25 % of the mask bits) - a property of
#      16-bit arithmetic, kept as a loose sanity bound only.
# Measured (gpurun_out r4_f16cal2, canonical nets): end to end 3.2e-2 ... 4.6e-2 max / 4.2e-3 ... 5.4e-3 rms - two CORRECT fp16 pipelines
# drift apart through 38 layers of rounding flips almost as far as either is from fp32 (6e-2 ... 9e-2 / 7e-3 ... 1e-2), so the end-to-end
# bound can only be ~1.5x tighter than (b).  The sharp check is per layer: a plain block 3e-3 ... 5e-3 max (single fp16 flips of stored
# values) / 3.5e-5 ... 4.6e-5 rms; a composed decoder entry 3.4e-3 ... 4.9e-3 max / 2.8e-4 ... 3.2e-4 rms (its weights are composed in fp64
# and rounded once, the oracle rounds the transposed conv's weights, its output and the 3x3 weights).
F16E_MAX, F16E_RMS = 0.1, 0.012                # (a) end to end, logits
F16_LAYER_MAX, F16_LAYER_RMS, F16_LAYER_RMS_COMPOSED = 1e-2, 1e-4, 7e-4    # (a) ONE block, fed with the engine's own inputs of that block


def _f16_layer_ok(name, got, want):
    d = np.asarray(got, np.float64) - np.asarray(want, np.float64)
    rms_tol = F16_LAYER_RMS_COMPOSED if (name.startswith('dec') and name.endswith('.c0')) else F16_LAYER_RMS
    return float(np.abs(d).max()) <= F16_LAYER_MAX and float(np.sqrt((d ** 2).mean())) <= rms_tol
F16_MAX, F16_RMS, F16_MASK = 0.15, 0.02, 0.01  # (b)


@pytest.mark.parametrize('name', ['k_two3', 'net5_128', 'wide64', 'xr_1ch', 'tiny_b37'])
def test_f16_mode_small_cases(name):
    from oracle import torch_oracle as O
    arch, B, H, W, seed = cases.SMALL_CASES[name]
    sd, blob = blob_for(arch, seed)
    x = cases.make_input(arch, B, H, W, seed)
    g = golden(name)['logits']
    ref16, inter16 = O.unet_forward(arch, sd, x, return_intermediates=True, emulate='f16')
    with Engine(arch, blob) as e:
        e.set_precision('f16')
        lg, mk = e.forward(x, logits=True, mask=(W % 32 == 0))
        if name != 'tiny_b37':                              # (2x2-pixel bottleneck: InstanceNorm over 4 values amplifies any rounding)
            d = lg - ref16.numpy()
            assert np.abs(d).max() <= F16E_MAX and np.sqrt((d ** 2).mean()) <= F16E_RMS, (float(np.abs(d).max()), float(np.sqrt((d ** 2).mean())))
        assert np.abs(lg - g).max() <= F16_MAX and np.sqrt(((lg - g) ** 2).mean()) <= F16_RMS
        if mk is not None:
            assert np.array_equal(unpack_mask(mk, W), _oracle_mask(lg))            # still bit-exact on its own logits
        if name == 'net5_128':                              # every block of the net, from the engine's own inputs of that block
            prog = {o['name']: o for o in arch.program()}
            for n, o in prog.items():
                if n.endswith('.up') or n == 'head':
                    continue
                if n.endswith('.c0') and n.startswith('dec'):
                    ins = (prog[n.replace('.c0', '.up')]['src'], o['skip'])
                else:
                    ins = (o['src'],)
                if n == 'enc0.c0' and not e.materialised(n):      # recomputed inside enc0.c1 (checked there, against the oracle's own enc0.c0)
                    continue
                srcs = [x if i == 'input' else (e.debug_tensor(i) if e.materialised(i) else inter16[i].numpy()) for i in ins]
                got = e.debug_tensor(n)
                want = O.layer_forward(arch, sd, n, *srcs, emulate='f16', storage_view=True).numpy()
                lvl_px = got.shape[2] * got.shape[3]
                if lvl_px >= 64:                            # (below: statistics over a handful of pixels)
                    assert _f16_layer_ok(n, got, want), (n, float(np.abs(got - want).max()), float(np.sqrt(np.mean((got - want) ** 2))))
            want = O.layer_forward(arch, sd, 'head', e.debug_tensor(prog['head']['src']), emulate='f16').numpy()
            assert np.abs(lg - want).max() <= F16_LAYER_MAX
        e.set_precision('split')                                                    # modes can be switched on a live engine
        lg2, _ = e.forward(x)
        assert np.abs(lg2 - g).max() <= (2e-3 if name == 'tiny_b37' else TOL)


@pytest.mark.parametrize('name', ['aniso_21', 'aniso_12_11'])
def test_per_axis_strides_in_every_precision_mode(name):
    """Plans whose stages pool one axis only ((2, 1) / (1, 2) / (1, 1) strides, transposed conv kernel = stride; the reference runs
    whatever plans.json names: ts2d/core/inference/nnu.py:164-165, prediction_worker.py:76-77).  Such stages run the generic
    implicit-GEMM kernel in every mode: exact / split against the fp32 goldens at 1e-4, the 16-bit mode against its own oracle per
    layer (the strided block and the anisotropic transposed conv from the engine's own inputs)."""
    import torch
    import torch.nn.functional as F
    from oracle import torch_oracle as O
    arch, B, H, W, seed = cases.SMALL_CASES[name]
    sd, blob = blob_for(arch, seed)
    x = cases.make_input(arch, B, H, W, seed)
    g = golden(name)
    prog = {o['name']: o for o in arch.program()}
    aniso = [n for n, o in prog.items() if tuple(o['stride']) not in ((1, 1), (2, 2)) or (n.endswith('.up') and tuple(o['stride']) != (2, 2))]
    assert aniso
    with Engine(arch, blob) as e:
        e.set_profiling(True)
        for mode in ('exact', 'split'):
            e.set_precision(mode)
            lg, mk = e.forward(x, logits=True, mask=True)
            assert np.abs(lg - g['logits']).max() <= TOL, mode
            assert np.array_equal(unpack_mask(mk, W), _oracle_mask(lg))
            for n in aniso:
                t = e.debug_tensor(n)
                assert t.shape == g[f'inter/{n}'].shape and np.abs(t - g[f'inter/{n}']).max() <= TOL, (mode, n)
            kern = e.op_kernels()
            assert all(kern[n] in ('conv_mfma_f32', 'convT_mfma_f32') for n in aniso), kern
        e.set_precision('f16')
        lg, _ = e.forward(x, logits=True)
        ref16, inter16 = O.unet_forward(arch, sd, x, return_intermediates=True, emulate='f16')
        d = lg - ref16.numpy()
        assert np.abs(d).max() <= F16E_MAX and np.sqrt((d ** 2).mean()) <= F16E_RMS, (float(np.abs(d).max()), float(np.sqrt((d ** 2).mean())))
        for n in aniso:
            o = prog[n]
            got = e.debug_tensor(n)
            src = e.debug_tensor(o['src'])
            if n.endswith('.up'):
                k = o['key']
                want = O._h(F.conv_transpose2d(O._h(torch.from_numpy(src)), O._h(torch.from_numpy(sd[f'{k}.weight'])), torch.from_numpy(sd[f'{k}.bias']),
                                               stride=tuple(o['stride']))).numpy()
            else:
                want = O.layer_forward(arch, sd, n, src, emulate='f16', storage_view=True).numpy()
            assert got.shape == want.shape and _f16_layer_ok(n, got, want), (n, float(np.abs(got - want).max()), float(np.sqrt(np.mean((got - want) ** 2))))


@pytest.mark.parametrize('feats,strides', [((24, 40, 72), None), ((16, 48, 80, 100), [(1, 1), (2, 2), (2, 2), (2, 1)]), ((40, 40), None)])
def test_stage_widths_that_are_not_multiples_of_32(feats, strides):
    """A plans.json may name any features_per_stage (the reference builds whatever arch_kwargs say: ts2d/core/inference/nnu.py:164-165).
    The engine runs such a stage rounded up to a multiple of 32 with zero weights / bias / gamma / beta in the added channels
    (csrc/engine.hip pad_arch / expand_blob) - exact, so the tolerance is the usual one; the weight blob and every tensor read back
    keep the caller's widths."""
    from oracle import torch_oracle as O
    arch = cases.unet(len(feats), feats, 5, cin=2, nconv=2, strides=strides)
    sd, blob = blob_for(arch, 77)
    dy, dx = arch.divisors
    B, H, W = 3, 8 * dy, 32 * dx
    x = cases.make_input(arch, B, H, W, 77)
    ref, inter = O.unet_forward(arch, sd, x, return_intermediates=True)
    ref16 = O.unet_forward(arch, sd, x, emulate='f16').numpy()
    with Engine(arch, blob) as e:
        for mode in ('exact', 'split'):
            e.set_precision(mode)
            lg, mk = e.forward(x, logits=True, mask=True)
            assert np.abs(lg - ref.numpy()).max() <= TOL, mode
            assert np.array_equal(unpack_mask(mk, W), _oracle_mask(lg))
        e.set_option('upc', 0); e.set_option('fuse0', 0)          # (every tensor materialised)
        e.forward(x, logits=True)
        for n, want in inter.items():
            got = e.debug_tensor(n)
            assert got.shape == tuple(want.shape), (n, got.shape)
            assert np.abs(got - want.numpy()).max() <= TOL, n
        e.set_option('upc', 1); e.set_option('fuse0', 1)
        e.set_precision('f16')
        lg, _ = e.forward(x, logits=True)
        d = lg - ref16
        assert np.abs(d).max() <= F16E_MAX and np.sqrt((d ** 2).mean()) <= F16E_RMS
        # a blob of the rounded-up architecture's size is refused: the caller's layout is the contract
        with pytest.raises(RuntimeError, match='weight blob has'):
            e.load_weights(np.zeros(blob.size + 1, np.float32))


@pytest.mark.parametrize('H,W,cin', [(112, 16, 1), (32, 16, 2), (16, 64, 2), (8, 128, 1)])
def test_f16_first_block_on_every_complete_tile_shape(H, W, cin):
    """The 16-bit first block stores an M tile (32 consecutive GEMM rows) as 2 KB through a per-wave LDS transpose; on 16-wide tiles
    (W = 16: the 256-pixel tile is 16 x 16) an M tile is TWO image rows, on 64- / 128-wide tiles a part of one.  Found by
    scripts/gpu_fuzz_parity.py (seed 601, case 5: 112 x 16 - the store addressed the M tile as one row of a 32-wide tile and the
    layer came back with inf): the layer is compared with the 16-bit oracle here on every tile shape the planner can choose."""
    from oracle import torch_oracle as O
    arch = cases.unet(3, (32, 64, 64), 4, cin=cin, nconv=1)
    sd, blob = blob_for(arch, 31)
    x = cases.make_input(arch, 2, H, W, 31)
    ref16 = O.unet_forward(arch, sd, x, emulate='f16')
    # both first-block kernels: conv3x3_first_split (round 6: the contraction as one fp16 hi / lo split product - the same transposed
    # 16-byte-store epilogue) and, with "first_split" = 0, the exact-fp32 conv3x3_first
    for fs, kname in ((1, 'conv3x3_first_split'), (0, 'conv3x3_first')):
        with Engine(arch, blob, options={'first_split': fs}) as e:
            e.set_precision('f16')
            e.set_profiling(True)
            lg, _ = e.forward(x, logits=True)
            assert e.op_kernels()['enc0.c0'] == kname
            got = e.debug_tensor('enc0.c0')
            want = O.layer_forward(arch, sd, 'enc0.c0', x, emulate='f16', storage_view=True).numpy()
            assert got.shape == want.shape and np.isfinite(got).all()
            assert _f16_layer_ok('enc0.c0', got, want), (fs, float(np.abs(got - want).max()), float(np.sqrt(np.mean((got - want) ** 2))))
            d = lg - ref16.numpy()
            assert np.abs(d).max() <= F16E_MAX and np.sqrt((d ** 2).mean()) <= F16E_RMS


def _level_kernels(kern, lo, hi):
    return {n: k for n, k in kern.items() if not n.endswith('.stats') and n != 'head' and lo <= int(n[3:n.index('.')]) <= hi}


@pytest.mark.parametrize('H,W,strides', [(640, 384, None), (640, 384, (2, 1)), (448, 576, 7)])
def test_canonical_widths_on_extents_other_than_512(H, W, strides):
    """The reference runs whatever patch size and pooling plans.json names (ts2d/core/inference/prediction_worker.py:76-77,
    nnu.py:164-165); nnU-Net's 2-D planner derives the patch from the median shape, so 640 x 384 / 448 x 576-like patches - levels of
    80 x 48, 40 x 24, 56 x 72, 28 x 36 pixels - are as likely as 512 x 512.  Canonical channel widths, B = 1, split mode against the
    fp32 oracle at 1e-4, 16-bit mode against its own oracle; the levels that are complete multiples of the fixed tiles must be served
    by the same kernels as on 512 x 512, the ragged ones by the one-image / composed kernels on extent-following tiles - never by the
    generic multi-image kernels."""
    from oracle import torch_oracle as O
    if strides == 7:
        arch = UNetArch.canonical(n_stages=7)
    else:
        arch = UNetArch.canonical()
        if strides is not None:
            arch.strides = tuple(arch.strides[:-1]) + (tuple(strides),)
    sd, blob = blob_for(arch, 1)
    x = cases.make_input(arch, 1, H, W, 3)
    ref = O.unet_forward(arch, sd, x).numpy()
    ref16 = O.unet_forward(arch, sd, x, emulate='f16').numpy()
    with Engine(arch, blob) as e:
        e.set_profiling(True)
        lg, mk = e.forward(x, logits=True, mask=True)
        assert np.abs(lg - ref).max() <= TOL
        assert np.array_equal(unpack_mask(mk, W), _oracle_mask(lg))
        assert_flips_are_tolerance_flips(lg, ref, unpack_mask(mk, W), O.logits_to_mask(ref).numpy())
        kern = e.op_kernels()
        lv = _level_kernels(kern, 0, 4)
        generic = {n: k for n, k in lv.items() if k in ('conv3x3_f16x3', 'conv3x3s2_f16x3', 'conv_mfma_f32', 'convT_mfma_f32')}
        assert not generic, generic                                    # levels 0-4: dedicated kernels only
        assert not any(n.endswith('.up') for n in lv), lv                 # ... every decoder entry composed with its transposed conv
        assert kern['enc0.c1'] == 'conv3x3_res32f' and kern['dec0.c0'] == 'conv3x3_up0' and kern['enc1.c1'] == 'conv3x3_f16x3_qp'
        if strides == (2, 1):
            assert kern['enc7.c0'] == 'conv_mfma_f32' and kern['dec6.up'] == 'convT_mfma_f32'
        # the same engine with the extent-following composed tiles switched off: transposed conv + conv on the ragged levels
        e.set_option('flex', 0)
        lg0, _ = e.forward(x, logits=True)
        assert np.abs(lg0 - ref).max() <= TOL and np.abs(lg0 - lg).max() <= 3e-5
        assert any(n.endswith('.up') for n in _level_kernels(e.op_kernels(), 2, 4))
        e.set_option('flex', 1)
        # ... and with the 512-thread stride-2 kernel kept off the level-dividing tiles ("flex2" = 0: the one-image stride-2 kernel serves
        # the ragged levels, as in round 4) - the other side of that switch, in both modes
        s2_on = {n: k for n, k in _level_kernels(kern, 1, 5).items() if n.endswith('.c0') and n.startswith('enc')}
        e.set_option('flex2', 0)
        lg2, _ = e.forward(x, logits=True)
        s2_off = {n: k for n, k in _level_kernels(e.op_kernels(), 1, 5).items() if n.endswith('.c0') and n.startswith('enc')}
        assert np.abs(lg2 - ref).max() <= TOL and np.abs(lg2 - lg).max() <= 3e-5
        assert s2_off != s2_on and not any(k in ('conv3x3s2_f16x3', 'conv_mfma_f32') for n, k in s2_off.items() if int(n[3]) <= 4), (s2_on, s2_off)
        e.set_precision('f16')
        lg16b, _ = e.forward(x, logits=True)
        d = lg16b - ref16
        assert np.abs(d).max() <= F16E_MAX and np.sqrt((d ** 2).mean()) <= F16E_RMS
        e.set_option('flex2', 2)
        e.set_precision('f16')
        lg16, mk16 = e.forward(x, logits=True, mask=True)
        d = lg16 - ref16
        assert np.abs(d).max() <= F16E_MAX and np.sqrt((d ** 2).mean()) <= F16E_RMS, (float(np.abs(d).max()), float(np.sqrt((d ** 2).mean())))
        assert np.array_equal(unpack_mask(mk16, W), _oracle_mask(lg16))
        lv16 = _level_kernels(e.op_kernels(), 0, 4)
        assert not {n: k for n, k in lv16.items() if k in ('conv3x3_f16x3', 'conv3x3s2_f16x3')} and not any(n.endswith('.up') for n in lv16), lv16


def test_f16_per_layer_oracle_on_the_256_and_512_channel_kernels_of_the_canonical_net():
    """the per-layer 16-bit oracle check reached conv3x3_h2 / conv3x3_upc_h2 only on a net whose 256- / 512-channel
    blocks run at 16 x 16 and 8 x 8 (other kernels).  Here: the canonical net at 512 x 512, B = 1 - enc3.c1 / enc4.c1 / dec4.c1
    (conv3x3_h2 with 4 / 8 column tiles) and dec3.c0 / dec4.c0 (conv3x3_upc_h2: KS = 4, the 8-chunk weight-DMA ring), each block
    from the engine's OWN inputs of that block against ONE block of the 16-bit oracle, bounds rms 1e-4 / 7e-4."""
    from oracle import torch_oracle as O
    arch = UNetArch.canonical()
    sd, blob = blob_for(arch, 1)
    x = cases.make_input(arch, 1, 512, 512, 1)
    prog = {o['name']: o for o in arch.program()}
    want_kernel = {'enc3.c1': 'conv3x3_h2', 'enc4.c1': 'conv3x3_h2', 'dec4.c1': 'conv3x3_h2', 'dec3.c0': 'conv3x3_upc_h2', 'dec4.c0': 'conv3x3_upc_h2',
                   'dec2.c0': 'conv3x3_upc_h2', 'enc2.c1': 'conv3x3_h2'}
    with Engine(arch, blob) as e:
        e.set_precision('f16')
        e.set_profiling(True)
        e.forward(x, logits=True)
        kern = e.op_kernels()
        for n, k in want_kernel.items():
            assert kern[n] == k, (n, kern[n])
            o = prog[n]
            ins = (prog[n.replace('.c0', '.up')]['src'], o['skip']) if n.endswith('.c0') else (o['src'],)
            srcs = [e.debug_tensor(i) for i in ins]
            got = e.debug_tensor(n)
            want = O.layer_forward(arch, sd, n, *srcs, emulate='f16', storage_view=True).numpy()
            assert got.shape == want.shape and _f16_layer_ok(n, got, want), (n, float(np.abs(got - want).max()), float(np.sqrt(np.mean((got - want) ** 2))))


@pytest.mark.parametrize('k32', [1, 0])
def test_f16_per_layer_oracle_on_the_stride_2_kernels_of_the_canonical_net(k32):
    """The 512-thread stride-2 kernel of the 16-bit mode on 32-channel chunks (round 6, "s2k32": the second part of its LDS images holds
    channels 16-31 of the chunk - half the items and barriers per tile) and on 16-channel chunks: enc1.c0 (resident weights, 64 columns),
    enc2 ... enc4.c0 (weights by DMA, 128 columns, 2 ... 8 chunks of 32) and enc5.c0 (the extent-following 16 x 16 tile), each block from
    the engine's own input against ONE block of the 16-bit oracle."""
    from oracle import torch_oracle as O
    arch = UNetArch.canonical()
    sd, blob = blob_for(arch, 1)
    x = cases.make_input(arch, 2, 512, 512, 2)
    prog = {o['name']: o for o in arch.program()}
    with Engine(arch, blob, options={'s2k32': k32}) as e:
        e.set_precision('f16')
        e.set_profiling(True)
        lg, _ = e.forward(x, logits=True)
        kern = e.op_kernels()
        for n in ('enc1.c0', 'enc2.c0', 'enc3.c0', 'enc4.c0', 'enc5.c0'):
            assert kern[n].startswith('conv3x3s2_v2<') and kern[n].endswith(',k32>') == bool(k32), (n, kern[n])
            got = e.debug_tensor(n)
            want = O.layer_forward(arch, sd, n, e.debug_tensor(prog[n]['src']), emulate='f16', storage_view=True).numpy()
            assert got.shape == want.shape and _f16_layer_ok(n, got, want), (n, float(np.abs(got - want).max()), float(np.sqrt(np.mean((got - want) ** 2))))
        d = lg[:1] - O.unet_forward(arch, sd, x[:1], emulate='f16').numpy()
        assert np.abs(d).max() <= F16E_MAX and np.sqrt((d ** 2).mean()) <= F16E_RMS


def test_config3_config5_in_f16():
    """Config 3 (a 26-head sub-model, 512x512) and config 5 (tsxr: 1-channel 1024x1024, 9 stages) in the 16-bit mode, against the
    16-bit oracle (tight) and the fp32 oracle (what the mode costs)."""
    from oracle import torch_oracle as O
    for arch, hw, seed in ((UNetArch.canonical(num_classes=26), 512, 4),
                           (UNetArch.canonical(input_channels=1, num_classes=26, n_stages=9), 1024, 7)):
        sd, blob = blob_for(arch, seed)
        x = cases.make_input(arch, 1, hw, hw, seed)
        ref = O.unet_forward(arch, sd, x).numpy()
        ref16 = O.unet_forward(arch, sd, x, emulate='f16').numpy()
        with Engine(arch, blob) as e:
            e.set_precision('f16')
            lg, mk = e.forward(x, logits=True, mask=True)
        d = lg - ref16
        assert np.abs(d).max() <= F16E_MAX and np.sqrt((d ** 2).mean()) <= F16E_RMS, (hw, float(np.abs(d).max()), float(np.sqrt((d ** 2).mean())))
        assert_flips_are_tolerance_flips(lg, ref16, unpack_mask(mk, hw), O.logits_to_mask(ref16).numpy())
        assert np.abs(lg - ref).max() <= F16_MAX and np.sqrt(((lg - ref) ** 2).mean()) <= F16_RMS
        assert (unpack_mask(mk, hw) != O.logits_to_mask(ref).numpy()).mean() <= F16_MASK


def test_one_image_kernels_are_bit_identical_to_the_generic_kernels():
    """The one-image-tile kernels (conv3x3_f16x3_one, conv3x3s2_f16x3_one, convT2x2_f16x3_one, first-layer fast epilogue) claim
    the arithmetic and summation order of the generic kernels: every convolution output must be bit-identical with them switched
    off (option "one" = 0).  The head differs by design (matrix-core head vs plain FMA chain)."""
    arch, B, H, W, seed = cases.SMALL_CASES['net5_128']
    _, blob = blob_for(arch, seed)
    x = cases.make_input(arch, B, H, W, seed)
    names = ['enc0.c0', 'enc0.c1', 'enc1.c0', 'enc2.c1', 'enc4.c1', 'dec3.c0', 'dec1.c1', 'dec0.c0', 'dec0.c1']
    with Engine(arch, blob) as e:                          # default path: includes the resident-weight 32 -> 32 kernel (with the first block fused in)
        lgr, _ = e.forward(x, logits=True)
        assert not e.materialised('enc0.c0')
        tr = {n: e.debug_tensor(n) for n in names if n != 'enc0.c0'}
    # (conv3x3_res32, conv3x3s2_v2 and conv3x3_upc sum in another order: compared by value below;
    #  conv3x3_f16x3_qp: same conv outputs, statistics summed over other tiles)
    off = {'res': 0, 's2v2': 0, 'upc': 0, 'q': 0, 'fuse0': 0}
    with Engine(arch, blob, options=off) as e:
        lg1, _ = e.forward(x, logits=True)
        t1 = {n: e.debug_tensor(n) for n in names}
    with Engine(arch, blob, options=dict(off, one=0)) as e:
        lg0, _ = e.forward(x, logits=True)
        t0 = {n: e.debug_tensor(n) for n in names}
    for n in names:
        assert np.array_equal(t0[n], t1[n]), n
    assert np.abs(lg0 - lg1).max() <= 1e-5
    # conv3x3_res32 (one 288-term accumulation per output instead of two 144-term chunks), conv3x3s2_v2 (16-channel chunks,
    # one tap per k-step) and conv3x3_upc (transposed conv composed into the block): same values to fp32 rounding
    for n in tr:
        assert np.abs(tr[n] - t1[n]).max() <= 2e-5, n
    assert np.abs(lgr - lg1).max() <= 2e-5


def test_composed_upsampling_block_matches_the_two_kernel_path_and_the_oracle():
    """kernels_upc.h / kernels_upq.h: ConvTranspose2d composed into the "up" half of the next 3x3 conv (parity-specific 2x2 weights over the
    coarse tensor, nine bias variants for the image border).  Checked against the two-kernel path (option "upc" = 0) layer by layer and
    against the torch oracle, with the transposed conv's BIAS blown up so that a wrong border variant cannot hide, on extents
    where some levels compose (complete 8 x 32 tiles) and others do not, one and several tiles per image, BN = 32 and 64."""
    from oracle import torch_oracle as O
    from totalsegmentator2d_amd import weights
    runs = ((cases.unet(4, (32, 64, 128, 128), 6, cin=2), 3, 64, 128, 31),        # levels 0-2 compose, level 2 has ONE tile per image
            (cases.unet(3, (64, 64, 128), 5, cin=1), 2, 32, 64, 32),              # BN = 64 at level 0; level 1 (16 x 32) composes
            (cases.unet(3, (32, 64, 128), 3, cin=1, nconv=1), 1, 256, 256, 33),   # many tiles per image
            (cases.unet(4, (32, 64, 128, 256), 4, cin=1), 2, 128, 256, 34))       # Cb = 256 at level 2 (32 x 64): the 512-thread conv3x3_upq, edge tiles on all sides
    for arch, B, H, W, seed in runs:
        sd = weights.synthetic_state_dict(arch, seed)
        for k in sd:
            if 'transpconvs' in k and k.endswith('bias'):
                sd[k] = (sd[k] * 40.0).astype(np.float32)
        blob = weights.pack_blob(arch, sd)
        x = cases.make_input(arch, B, H, W, seed)
        ref = O.unet_forward(arch, sd, x).numpy()
        names = [f'dec{l}.c0' for l in range(arch.n_stages - 1)]
        with Engine(arch, blob) as e:
            lg1, _ = e.forward(x, logits=True)
            t1 = {n: e.debug_tensor(n) for n in names}
            with pytest.raises(Exception, match='not materialised'):
                e.debug_tensor('dec0.up')
            e.check()
        with Engine(arch, blob, options={'upc': 0}) as e:
            lg0, _ = e.forward(x, logits=True)
            t0 = {n: e.debug_tensor(n) for n in names}
            e.debug_tensor('dec0.up')
        for n in names:
            assert np.abs(t1[n] - t0[n]).max() <= 3e-5, (n, float(np.abs(t1[n] - t0[n]).max()))
        assert np.abs(lg1 - ref).max() <= TOL and np.abs(lg0 - ref).max() <= TOL


def test_level0_composed_block_runs_the_dedicated_kernel():
    """kernels_up0.h: the 64 -> (32 | 32) -> 32 decoder entry of the canonical level 0 as a persistent kernel (resident skip weights,
    composed weights streamed in fragment order, 16x16x32 transposed product).  The kernel must be the one that serves dec0.c0 in
    the split and the 16-bit mode; its output is checked against conv3x3_upc (option "up0" = 0), the two-kernel path ("upc" = 0) and the
    oracle on an extent with border tiles on all four sides, one tile per image column, and a segment that ends inside an image."""
    from oracle import torch_oracle as O
    from totalsegmentator2d_amd import weights
    arch = cases.unet(3, (32, 64, 128), 5, cin=2)
    # 9 x 3 tiles, ONE tile (every border at once), 32 x 16 tiles, and 32 x 12 tiles in segments of 4 (found by scripts/gpu_fuzz_parity.py:
    # the activation plan took the block for un-composed when the tile count was no power of two and put its output on a live buffer)
    for B, H, W, seed in ((3, 72, 96, 41), (2, 8, 32, 42), (1, 256, 512, 43), (1, 256, 384, 44)):
        seg = {'u0seg': 4} if W == 384 else {}
        sd = weights.synthetic_state_dict(arch, seed)
        for k in sd:
            if 'transpconvs' in k and k.endswith('bias'):
                sd[k] = (sd[k] * 40.0).astype(np.float32)      # a wrong bias variant at the border cannot hide
        blob = weights.pack_blob(arch, sd)
        x = cases.make_input(arch, B, H, W, seed)
        ref = O.unet_forward(arch, sd, x).numpy()
        out = {}
        ref16, inter16 = O.unet_forward(arch, sd, x, return_intermediates=True, emulate='f16')
        for tag, opt in (('up0', {}), ('upc', {'up0': 0}), ('two', {'upc': 0})):
            with Engine(arch, blob, options=dict(seg, **opt)) as e:
                e.set_profiling(True)
                lg, _ = e.forward(x, logits=True)
                kern = e.op_kernels()['dec0.c0']
                e.set_profiling(False)
                t = e.debug_tensor('dec0.c0')
                lh = kh = th = None
                if tag != 'two':                                # the 16-bit mode of the same two kernels
                    e.set_precision('f16')
                    e.set_profiling(True)
                    lh, _ = e.forward(x, logits=True)
                    kh = e.op_kernels()['dec0.c0']
                    e.set_profiling(False)
                    # per-layer, against the 16-bit oracle fed with the ENGINE's own inputs of the block (nothing accumulates in front)
                    th = e.debug_tensor('dec0.c0')
                    o16 = O.layer_forward(arch, sd, 'dec0.c0', e.debug_tensor('dec1.c1'), e.debug_tensor('enc0.c1'), emulate='f16', storage_view=True).numpy()
                    assert _f16_layer_ok('dec0.c0', th, o16), (tag, H, W, float(np.abs(th - o16).max()), float(np.sqrt(np.mean((th - o16) ** 2))))
                out[tag] = (lg, t, kern, lh, kh)
        pow2 = (W // 32) & (W // 32 - 1) == 0 and ((W // 32) * (H // 8)) & ((W // 32) * (H // 8) - 1) == 0
        assert out['up0'][2] == 'conv3x3_up0' and out['up0'][4] == 'conv3x3_up0' and out['two'][2] != 'conv3x3_up0'
        if pow2:                                                # (conv3x3_upc wants power-of-two tile counts; otherwise the generic kernel runs)
            assert out['upc'][2] == 'conv3x3_upc<32>' and out['upc'][4] == 'conv3x3_upc_h<32>'
        assert 'up0' not in out['upc'][2] and 'up0' not in out['upc'][4]
        for tag in ('upc', 'two'):
            assert np.abs(out['up0'][1] - out[tag][1]).max() <= 3e-5, (tag, H, W)
        assert np.abs(out['up0'][0] - ref).max() <= TOL
        for tag in ('up0', 'upc'):                              # both 16-bit kernels against the 16-bit oracle, end to end
            d16 = out[tag][3] - ref16.numpy()
            assert np.abs(d16).max() <= F16E_MAX and np.sqrt(np.mean(d16 ** 2)) <= F16E_RMS, (tag, float(np.abs(d16).max()), H, W)


def test_f16_composed_block_on_16x32_tiles():
    """kernels_upc_h2.h: the 16-bit composed decoder entry with four M tiles per wave (16 x 32 output tiles, skip-half weights by
    LDS-DMA).  Must be the kernel that serves the 64-column blocks on complete 16 x 32 tiles; both it and conv3x3_upc_h
    (option "uh2" = 0: same arithmetic, other summation order) are compared with the 16-BIT oracle - end to end and per layer with
    the engine's own layer inputs - transposed-conv bias x 40 (border variants), tiles on every border, one and several tiles per
    image, KS = 2 and 4 (128 / 64 coarse channels per barrier pair)."""
    from oracle import torch_oracle as O
    from totalsegmentator2d_amd import weights
    for arch, B, H, W, seed in ((cases.unet(3, (64, 64, 128), 5, cin=1), 2, 64, 64, 51),        # levels 0 and 1 compose (KS = 4, 2 x 4 and 1 x 2 tiles)
                                (cases.unet(3, (64, 96, 160), 3, cin=2), 1, 32, 128, 52)):       # Cb = 96: KS = 2; 4 x 2 tiles, all of them border tiles
        sd = weights.synthetic_state_dict(arch, seed)
        for k in sd:
            if 'transpconvs' in k and k.endswith('bias'):
                sd[k] = (sd[k] * 40.0).astype(np.float32)
        blob = weights.pack_blob(arch, sd)
        x = cases.make_input(arch, B, H, W, seed)
        ref16 = O.unet_forward(arch, sd, x, emulate='f16').numpy()
        out = {}
        for uh2 in (1, 0):
            # ("h2": the plain C -> C blocks of the same tiling - conv3x3_h2 = the skip phase alone)
            with Engine(arch, blob, options={'uh2': uh2, 'h2': uh2}) as e:
                e.set_precision('f16')
                e.set_profiling(True)
                lg, _ = e.forward(x, logits=True)
                out[uh2] = (lg, e.op_kernels()['dec0.c0'], e.op_kernels()['dec0.c1'])
                e.set_profiling(False)
                for name, ins in (('dec0.c0', ('dec1.c1', 'enc0.c1')), ('dec1.c0', ('enc2.c1', 'enc1.c1')), ('dec0.c1', ('dec0.c0',)), ('enc1.c1', ('enc1.c0',))):
                    got = e.debug_tensor(name)
                    want = O.layer_forward(arch, sd, name, *[e.debug_tensor(i) for i in ins], emulate='f16', storage_view=True).numpy()
                    assert _f16_layer_ok(name, got, want), (uh2, name, float(np.abs(got - want).max()), float(np.sqrt(np.mean((got - want) ** 2))))
        assert out[1][1] == 'conv3x3_upc_h2' and out[0][1] == 'conv3x3_upc_h<64>'
        assert out[1][2] == 'conv3x3_h2' and out[0][2] == 'conv3x3_h32<64>'
        for uh2 in (1, 0):
            d = out[uh2][0] - ref16
            assert np.abs(d).max() <= F16E_MAX and np.sqrt(np.mean(d ** 2)) <= F16E_RMS, (uh2, float(np.abs(d).max()))


def test_randomised_shapes_against_the_torch_oracle():
    """A fixed-seed slice of scripts/gpu_fuzz_parity.py: random small architectures, channel counts, batch sizes and ragged
    extents (complete and partial pixel tiles, one- and multi-image tiles, power-of-two and other tilings), all three modes."""
    from oracle import torch_oracle as O
    from totalsegmentator2d_amd import weights, prng
    rng = np.random.default_rng(7)
    for t in range(8):
        ns = int(rng.integers(2, 5))
        feats = [32]
        for _ in range(1, ns):
            feats.append(min(feats[-1] * int(rng.choice([1, 2])), 128))
        K = int(rng.integers(1, 27)); cin = int(rng.integers(1, 4)); mult = 2 ** (ns - 1)
        H = mult * int(rng.integers(1, 96 // mult + 1)); W = mult * int(rng.integers(2, 128 // mult + 1))
        B = int(rng.integers(1, 5))
        arch = cases.unet(ns, feats, K, cin=cin, nconv=int(rng.integers(1, 3)))
        sd = weights.synthetic_state_dict(arch, 300 + t)
        x = prng.normal_f32(400 + t, 1, (B, cin, H, W))
        ref = O.unet_forward(arch, sd, x).numpy()
        ref16 = O.unet_forward(arch, sd, x, emulate='f16').numpy()
        small = (H >> (ns - 1)) * (W >> (ns - 1)) < 16      # a bottleneck of a few pixels: InstanceNorm amplifies the fp16 roundings
        with Engine(arch, weights.pack_blob(arch, sd)) as e:
            for mode, tol, want in (('split', TOL, ref), ('exact', TOL, ref), ('f16', 0.2 if small else F16E_MAX, ref if small else ref16)):
                e.set_precision(mode)
                lg, mk = e.forward(x, logits=True, mask=(W % 32 == 0))
                assert np.abs(lg - want).max() <= tol, (t, mode, feats, K, cin, B, H, W, float(np.abs(lg - want).max()))
                if mk is not None:
                    assert np.array_equal(unpack_mask(mk, W), _oracle_mask(lg)), (t, mode)


def test_randomised_strides_and_extents_against_the_torch_oracle():
    """A fixed-seed slice of `scripts/gpu_fuzz_parity.py <seed> <n> geo` (round 5): random per-axis strides in the last stages, extents
    that are multiples of the strides only - extent-following tiles, composed blocks on them, whole small images per tile at any
    extent, the generic kernel for anisotropic stages - all three modes against the oracles."""
    from oracle import torch_oracle as O
    from totalsegmentator2d_amd import weights, prng
    rng = np.random.default_rng(11)
    for t in range(8):
        ns = int(rng.integers(3, 7))
        feats = [32]
        for i in range(1, ns):
            feats.append(min(feats[-1] * 2, 256))
        strides = [(1, 1)]
        for i in range(1, ns):
            strides.append((2, 2) if (i < ns - 2 or rng.random() < 0.6) else [(2, 1), (1, 2), (2, 2)][int(rng.integers(0, 3))])
        arch = cases.unet(ns, feats, int(rng.integers(1, 27)), cin=int(rng.integers(1, 3)), nconv=int(rng.integers(1, 3)), strides=strides)
        dy, dx = arch.divisors
        H = dy * int(rng.integers(1, max(2, 288 // dy) + 1)); W = dx * int(rng.integers(1, max(2, 288 // dx) + 1))
        while (H // dy) * (W // dx) < 4:
            W += dx
        B = int(rng.integers(1, 4))
        sd = weights.synthetic_state_dict(arch, 700 + t)
        x = prng.normal_f32(800 + t, 1, (B, arch.input_channels, H, W))
        ref = O.unet_forward(arch, sd, x).numpy()
        ref16 = O.unet_forward(arch, sd, x, emulate='f16').numpy()
        with Engine(arch, weights.pack_blob(arch, sd)) as e:
            for mode, tol, want in (('split', TOL, ref), ('exact', TOL, ref), ('f16', 0.2, ref16)):
                e.set_precision(mode)
                lg, mk = e.forward(x, logits=True, mask=(W % 32 == 0))
                assert np.abs(lg - want).max() <= tol, (t, mode, feats, strides, B, H, W, float(np.abs(lg - want).max()))
                if mk is not None:
                    assert np.array_equal(unpack_mask(mk, W), _oracle_mask(lg)), (t, mode)


def test_short_wide_images_get_one_image_per_tile():
    """Regression (found by scripts/gpu_fuzz_parity.py): at a level where the image is shorter than 8 rows but wider than 32
    columns (8x120 input, 4 stages -> 4x60, 2x30, 1x15) a tile of TH x 32 < 256 pixels must still hold ONE image."""
    from oracle import torch_oracle as O
    from totalsegmentator2d_amd import weights, prng
    arch = cases.unet(4, (64, 64, 128, 256), 19, cin=2, nconv=2)
    sd = weights.synthetic_state_dict(arch, 119)
    for B, H, W in ((1, 8, 120), (3, 16, 200)):
        x = prng.normal_f32(219, 1, (B, 2, H, W))
        ref = O.unet_forward(arch, sd, x).numpy()
        with Engine(arch, weights.pack_blob(arch, sd)) as e:
            for mode, tol in (('split', TOL), ('exact', TOL), ('f16', 0.2)):
                e.set_precision(mode)
                lg, _ = e.forward(x, logits=True)
                assert np.isfinite(lg).all() and np.abs(lg - ref).max() <= tol, (mode, B, H, W)


def test_torch_tensors_are_ordered_with_the_callers_stream():
    """Engine.forward on torch CUDA tensors must be ordered with the caller's torch work without an explicit synchronise:
    the outputs are consumed by torch ops on the default stream right after the call (found by a soak script that cloned the
    logits without synchronising)."""
    import torch
    arch = UNetArch.canonical()
    _, blob = blob_for(arch, 1)
    x = torch.randn(24, 2, 512, 512, device='cuda')
    with Engine(arch, blob) as e:
        a, _ = e.forward(x, logits=True)
        torch.cuda.synchronize()
        ref = a.clone()
        for _ in range(3):
            x2 = x * 1.0                               # producer on the default stream, consumed by the engine
            b, _ = e.forward(x2, logits=True)
            got = b.clone()                            # consumer on the default stream, no synchronise in between
            torch.cuda.synchronize()
            assert torch.equal(got, ref)


def test_config3_full_model_set_batch128_f16():
    """BASELINE config 3: the five ts2d-v2 sub-models (K = 18/23/24/26/26) on ONE batch of 128 2x512x512 slices in the 16-bit
    mode -> 117-channel packed mask in sorted-id / label order.  (The config says bf16; the 16-bit mode of this engine is fp16 -
    DESIGN.md section 4 records the measured bf16 error and why fp16 is the better 16-bit format for this net.)  Checks: channel
    order of the merged mask, batch independence at B = 128 (slice i alone == slice i in the batch), and each sub-model's
    slice-0 logits against the fp32 oracle within the stated 16-bit tolerance."""
    import torch
    from oracle import torch_oracle as O
    from totalsegmentator2d_amd.submodels import SubModelSet, TS2D_V2_HEADS
    from totalsegmentator2d_amd import parallel
    B = 128
    x = parallel.synth_slices(0, 0, 0, B, (2, 512, 512))
    seeds = {mid: i + 1 for i, mid in enumerate(sorted(TS2D_V2_HEADS))}
    merged_rows, lo = {}, 0
    for mid in sorted(TS2D_V2_HEADS):                      # one engine alive at a time (~7 GB of workspace each at B = 128 in this mode)
        arch = UNetArch.canonical(num_classes=TS2D_V2_HEADS[mid])
        sd, blob = blob_for(arch, seeds[mid])
        with SubModelSet([(mid, arch, blob)], precision='f16') as ms:
            m = ms.forward_masks(x)[0]
            torch.cuda.synchronize()
            assert m.shape == (B, TS2D_V2_HEADS[mid], 512, 16)
            lg0, mk0 = ms.engines[0].forward(x[:1].contiguous(), logits=True, mask=True)
            lg77, mk77 = ms.engines[0].forward(x[77:78].contiguous(), logits=True, mask=True)
            torch.cuda.synchronize()
            assert torch.equal(mk0[0], m[0]) and torch.equal(mk77[0], m[77])            # batch independence at B = 128
            ref = O.unet_forward(arch, sd, x[:1].cpu().numpy()).numpy()
            d16 = lg0.cpu().numpy() - O.unet_forward(arch, sd, x[:1].cpu().numpy(), emulate='f16').numpy()
            assert np.abs(d16).max() <= F16E_MAX and np.sqrt((d16 ** 2).mean()) <= F16E_RMS, (mid, float(np.abs(d16).max()))
            d = lg0.cpu().numpy() - ref
            assert np.abs(d).max() <= F16_MAX and np.sqrt((d ** 2).mean()) <= F16_RMS
            assert (unpack_mask(mk0.cpu().numpy(), 512) != O.logits_to_mask(ref).numpy()).mean() <= F16_MASK
            merged_rows[mid] = (lo, lo + TS2D_V2_HEADS[mid], m[:2].clone())
            lo += TS2D_V2_HEADS[mid]
    assert lo == 117
    merged = SubModelSet.merge([merged_rows[mid][2] for mid in sorted(merged_rows)])
    assert merged.shape == (2, 117, 512, 16)
    for mid, (a, b, m) in merged_rows.items():
        assert torch.equal(merged[:, a:b], m)
    assert [merged_rows[m][0] for m in sorted(merged_rows)] == [0, 18, 41, 65, 91]       # cardiac, muscles, organs, ribs, vertebrae


def test_activation_buffers_are_shared_by_liveness():
    """The activations of a forward share ONE arena by liveness (include/ts2d_engine.h, ABI 5): same logits bit for bit as with
    one buffer per tensor, a third of the activation memory; an intermediate tensor whose bytes were recycled is refused by the
    C-ABI until ts2d_engine_set_keep_activations(1) (the Python accessor switches it on and re-runs)."""
    import ctypes
    from totalsegmentator2d_amd import _lib
    from totalsegmentator2d_amd.arch import UNetArch
    arch = UNetArch.canonical(num_classes=5)
    sd, blob = blob_for(arch, 91)
    x = cases.make_input(arch, 2, 256, 256, 91)
    with Engine(arch, blob) as e:
        a, ma = e.forward(x, logits=True, mask=True)
        shared = e.device_bytes() - e.weight_buffer()[1]          # workspace only (device_bytes is the allocation's high-water mark)
        out = np.empty(1 << 24, np.float32); dims = (ctypes.c_int32 * 4)()
        rc = e.lib.ts2d_engine_debug_tensor(e._h, b'enc1.c1', out.ctypes.data, out.size, ctypes.byref(dims))
        assert rc != 0 and 'overwritten' in _lib.last_error()
        e.keep_activations(True)
        b, mb = e.forward(x, logits=True, mask=True)
        private = e.device_bytes() - e.weight_buffer()[1]
        assert np.array_equal(a, b) and np.array_equal(ma, mb)
        assert e.debug_tensor('enc1.c1').shape == (2, 64, 128, 128)
        assert shared < 0.62 * private, (shared, private)         # (the workspace also holds the staged input / logits / masks)
        for mode in ('exact', 'f16'):                             # the plan is remade per mode (the composed decoder entries differ)
            e.set_precision(mode)
            e.keep_activations(False)
            a, ma = e.forward(x, logits=True, mask=True)
            e.keep_activations(True)
            b, mb = e.forward(x, logits=True, mask=True)
            assert np.array_equal(a, b) and np.array_equal(ma, mb), mode

"""Round 6: the small-batch dispatch ("sbk").  ``TS2D.predict`` runs tiles x mirrors = 8 slices per sub-model and the
reference one slice per ``network()`` call (ts2d/core/inference/prediction_worker.py:209); at the <= 32 x 32 levels the persistent /
composed kernels of full batches then launch a handful of workgroups on 256 CUs.  Below one workgroup per CU the engine splits K on
the one-image kernels (deterministic two-phase reduction) and runs a composed decoder entry as transposed conv + conv.  Checked here:
parity with the oracles in every mode at B = 1 / 3 / 8, that the path really is taken, the bound across regimes (a slice alone vs in a
full batch), bit-reproducibility at equal B, and a small batch inside a workspace that was planned for a large one."""
import numpy as np
import pytest

from tests import cases
from tests.test_gpu_parity import TOL, F16E_MAX, F16E_RMS, _f16_layer_ok, _oracle_mask, assert_flips_are_tolerance_flips, blob_for
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd.engine import Engine, unpack_mask

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B', [1, 3, 8])
def test_canonical_net_small_batches_follow_the_oracles(B):
    from oracle import torch_oracle as O
    arch = UNetArch.canonical()
    sd, blob = blob_for(arch, 1)
    x = cases.make_input(arch, B, 512, 512, 7)
    ref = O.unet_forward(arch, sd, x[:1]).numpy()
    ref16 = O.unet_forward(arch, sd, x[:1], emulate='f16').numpy()
    with Engine(arch, blob) as e, Engine(arch, blob, options={'sbk': 0}) as e0:
        e.set_profiling(True); e0.set_profiling(True)
        lg, mk = e.forward(x, logits=True, mask=True)
        lg0, _ = e0.forward(x, logits=True)
        k1, k0 = e.op_kernels(), e0.op_kernels()
        # the path is taken: the deep composed entries run as two kernels, the 16 x 16 / 32 x 32 blocks on the one-image kernels (split-K)
        assert 'dec5.up' in k1 and 'dec5.up' not in k0, (sorted(k1), sorted(k0))
        assert k1['enc5.c1'] == 'conv3x3_f16x3_one<64>' and k1['enc5.c0'] == 'conv3x3s2_f16x3_one'
        if B == 1:
            assert 'dec4.up' in k1 and k1['enc4.c1'] == 'conv3x3_f16x3_one<64>' and k0['enc4.c1'] == 'conv3x3_f16x3_qp'
        assert k1['enc1.c1'] == k0['enc1.c1'] == 'conv3x3_f16x3_qp' and k1['dec0.c0'] == 'conv3x3_up0'      # the big levels are untouched
        assert np.abs(lg[:1] - ref).max() <= TOL
        assert np.abs(lg - lg0).max() <= 3e-5                                    # two valid summation orders of the same arithmetic
        assert np.array_equal(unpack_mask(mk, 512), _oracle_mask(lg))
        assert_flips_are_tolerance_flips(lg[:1], ref, unpack_mask(mk[:1], 512), O.logits_to_mask(ref).numpy())
        lg2, mk2 = e.forward(x, logits=True, mask=True)
        assert np.array_equal(lg, lg2) and np.array_equal(mk, mk2)              # same B: bit for bit
        # 16-bit mode: end to end against its own oracle, and the split-K blocks per layer from the engine's own inputs
        e.set_precision('f16')
        lh, mh = e.forward(x, logits=True, mask=True)
        kh = e.op_kernels()
        assert 'dec5.up' in kh and kh['enc5.c1'] == 'conv3x3_h32<64>'
        d = lh[:1] - ref16
        assert np.abs(d).max() <= F16E_MAX and np.sqrt((d ** 2).mean()) <= F16E_RMS, (float(np.abs(d).max()), float(np.sqrt((d ** 2).mean())))
        assert np.array_equal(unpack_mask(mh, 512), _oracle_mask(lh))
        if B == 1:
            for name, src in (('enc5.c1', 'enc5.c0'), ('enc4.c1', 'enc4.c0'), ('enc5.c0', 'enc4.c1')):
                got = e.debug_tensor(name)
                want = O.layer_forward(arch, sd, name, e.debug_tensor(src), emulate='f16', storage_view=True).numpy()
                assert _f16_layer_ok(name, got, want), (name, float(np.abs(got - want).max()), float(np.sqrt(np.mean((got - want) ** 2))))


def test_a_slice_alone_and_in_a_full_batch():
    """Across regimes (B = 1 splits K and un-composes, B = 64 does not) a slice's logits agree to 3e-5 - the bound the suite uses for two
    valid evaluation orders (tests/test_gpu_parity.py) - and its mask bits differ only at the threshold; with "sbk" = 0 the slice is
    bit-identical (that property is tests/test_gpu_parity.py::test_full_batch_properties_config2's)."""
    import torch
    arch = UNetArch.canonical()
    _, blob = blob_for(arch, 1)
    gen = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(64, 2, 512, 512, device='cuda', generator=gen)
    with Engine(arch, blob) as e:
        lg, mk = e.forward(x, logits=True, mask=True)
        torch.cuda.synchronize()
        big = lg[[0, 37, 63]].cpu().numpy(); bigm = mk[[0, 37, 63]].cpu().numpy()
        for j, i in enumerate((0, 37, 63)):
            li, mi = e.forward(x[i:i + 1].contiguous(), logits=True, mask=True)      # (the workspace stays the one planned for B = 64)
            torch.cuda.synchronize()
            li = li.cpu().numpy(); mi = mi.cpu().numpy()
            assert np.abs(li[0] - big[j]).max() <= 3e-5, i
            flips = unpack_mask(mi, 512)[0] != unpack_mask(bigm[j:j + 1], 512)[0]
            assert flips.sum() <= 64 and (np.abs(big[j][flips]) <= 3e-5).all()
        lg3, _ = e.forward(x, logits=True)
        torch.cuda.synchronize()
        assert torch.equal(lg3, lg)                                                   # and back: the full batch is unchanged


@pytest.mark.parametrize('H,W', [(640, 384), (448, 576)])
def test_small_batches_on_other_extents(H, W):
    """Extent-following tiles (round 5) under the small-batch dispatch: B = 2 on 640 x 384 / 448 x 576 canonical-width nets."""
    from oracle import torch_oracle as O
    arch = UNetArch.canonical(n_stages=7) if H == 448 else UNetArch.canonical()
    sd, blob = blob_for(arch, 1)
    x = cases.make_input(arch, 2, H, W, 5)
    ref = O.unet_forward(arch, sd, x).numpy()
    with Engine(arch, blob) as e:
        e.set_profiling(True)
        lg, mk = e.forward(x, logits=True, mask=True)
        assert any(n.endswith('.up') for n in e.op_kernels()), sorted(e.op_kernels())
        assert np.abs(lg - ref).max() <= TOL
        assert np.array_equal(unpack_mask(mk, W), _oracle_mask(lg))


def test_small_batch_inside_a_workspace_planned_for_a_large_one():
    """A workspace laid out for B = 16 (decoder entries composed: their upsampled tensors have no buffer in the plan) serves B = 1, where
    those entries run as two kernels: the upsampled tensor goes to the workspace's scratch region.  Same bits as an engine that only
    ever saw B = 1; the large batch afterwards is unchanged."""
    arch = cases.unet(5, (32, 64, 128, 256, 512), 6)
    sd, blob = blob_for(arch, 62)
    x = cases.make_input(arch, 16, 128, 128, 62)
    with Engine(arch, blob) as e, Engine(arch, blob) as e1:
        e.set_profiling(True)
        big, _ = e.forward(x, logits=True)
        kb = e.op_kernels()
        one, _ = e.forward(x[3:4].copy(), logits=True)
        k1 = e.op_kernels()
        assert [n for n in k1 if n.endswith('.up')] != [n for n in kb if n.endswith('.up')], (sorted(kb), sorted(k1))
        fresh, _ = e1.forward(x[3:4].copy(), logits=True)
        assert np.array_equal(one, fresh)
        assert np.abs(one[0] - big[3]).max() <= 3e-5
        again, _ = e.forward(x, logits=True)
        assert np.array_equal(again, big)


@pytest.mark.parametrize('mode', ['split', 'f16'])
def test_small_batches_in_one_shared_workspace(mode):
    """The sub-models of a set share ONE caller-provided workspace (SubModelSet; BASELINE config 3).  Reserved for a batch of 12 (whose plan
    composes every decoder entry), then driven with 2 slices - split-K partials and the scratch for an un-composed upsampled tensor live
    inside the shared memory: bit-identical to engines with memory of their own, at both batch sizes, in either order."""
    import torch
    from totalsegmentator2d_amd.submodels import SubModelSet
    models = []
    for mid, K, seed in (('a_small', 3, 31), ('b_wide', 26, 32)):
        arch = cases.unet(5, (32, 64, 128, 256, 512), K)
        models.append((mid, arch, blob_for(arch, seed)[1]))
    xb = torch.from_numpy(cases.make_input(models[0][1], 12, 128, 160, 5)).cuda()
    xs = xb[4:6].contiguous()
    want = {}
    for mid, arch, blob in models:
        with Engine(arch, blob) as e:
            e.set_precision(mode)
            e.set_profiling(True)
            for tag, x in (('small', xs), ('big', xb)):
                _, m = e.forward(x, logits=False, mask=True)
                torch.cuda.synchronize()
                want[mid, tag] = m.cpu().numpy().copy()
                if tag == 'small':
                    assert any(n.endswith('.up') and int(n[3]) < 3 for n in e.op_kernels()), sorted(e.op_kernels())      # the path under test
    with SubModelSet(models, precision=mode) as ms:
        ms.reserve(12, 128, 160)
        for tag, x in (('small', xs), ('big', xb), ('small', xs)):
            got = ms.forward_masks(x)
            torch.cuda.synchronize()
            for (mid, _, _), g in zip(models, got):
                assert np.array_equal(g.cpu().numpy(), want[mid, tag]), (mid, tag)

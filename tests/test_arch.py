"""Host logic: architecture descriptor, program order, accounting (BASELINE.md section 2 figures)."""
import numpy as np
import pytest

from tests import cases
from totalsegmentator2d_amd import prng, weights
from totalsegmentator2d_amd.arch import UNetArch, OP_CONV3X3, OP_CONVT2X2, OP_HEAD1X1


def test_canonical_accounting_matches_baseline_md():
    a = UNetArch.canonical()
    w = a.work(512, 512)
    assert a.n_params() == 46321394
    assert abs(w['flops'] / 1e9 - 119.60) < 0.01           # 119.60 GFLOP / slice
    assert abs(w['act_bytes'] / 1e6 - 743.4) < 0.1          # 743.4 MB layer-wise activation bytes
    assert abs(w['io_bytes'] / 1e6 - 20.97) < 0.01          # 20.97 MB compulsory I/O
    assert abs(w['weight_bytes'] / 1e6 - 185.3) < 0.1
    w26 = UNetArch.canonical(num_classes=26).work(512, 512)
    assert abs(w26['flops'] / 1e9 - 119.74) < 0.01 and abs(w26['act_bytes'] / 1e6 - 751.8) < 0.1
    xr = UNetArch.canonical(input_channels=1, num_classes=26, n_stages=9).work(1024, 1024)
    assert abs(xr['flops'] / 1e9 - 479.44) < 0.05


def test_program_order_and_concat():
    a = UNetArch.canonical()
    ops = a.program()
    assert [o['op'] for o in ops].count(OP_CONV3X3) == 30 and [o['op'] for o in ops].count(OP_CONVT2X2) == 7
    assert ops[-1]['op'] == OP_HEAD1X1 and ops[-1]['key'] == 'decoder.seg_layers.6'
    d = next(o for o in ops if o['name'] == 'dec0.c0')
    assert d['src'] == 'dec0.up' and d['skip'] == 'enc0.c1' and d['cin'] == 32 and d['cin_skip'] == 32   # cat((up, skip), 1)
    assert next(o for o in ops if o['name'] == 'enc3.c0')['stride'] == (2, 2)
    keys = [k for k, _ in a.param_specs()]
    assert keys[0] == 'encoder.stages.0.0.convs.0.conv.weight' and 'decoder.transpconvs.0.weight' in keys
    assert dict(a.param_specs())['decoder.transpconvs.0.weight'] == (512, 512, 2, 2)


def test_per_axis_strides():
    """A plan whose last stage pools one axis only (nnU-Net's planner treats the axes separately; e.g. a 640 x 320 patch)."""
    st = ((1, 1),) + ((2, 2),) * 6 + ((2, 1),)
    a = UNetArch(strides=st)
    a.validate()
    assert a.divisors == (128, 64) and a.divisor == 128
    assert a.extent(7, 640, 320) == (5, 5) and a.extent(6, 640, 320) == (10, 5) and a.extent(0, 640, 320) == (640, 320)
    ops = {o['name']: o for o in a.program()}
    assert ops['enc7.c0']['stride'] == (2, 1) and ops['enc7.c1']['stride'] == (1, 1) and ops['dec6.up']['stride'] == (2, 1)
    assert ops['dec5.up']['stride'] == (2, 2)
    assert dict(a.param_specs())['decoder.transpconvs.0.weight'] == (512, 512, 2, 1)      # kernel = stride
    iso = UNetArch.canonical()
    # the (2, 1) stage keeps twice the columns at level 7: two 3x3 blocks and a decoder entry cost more, the transposed conv has 2 taps
    assert a.work(640, 320)['macs'] > iso.work(640, 320)['macs'] * 0.99
    assert a.n_params() == iso.n_params() - 512 * 512 * 2
    with pytest.raises(NotImplementedError):
        UNetArch(strides=((1, 1),) + ((2, 2),) * 6 + ((3, 1),)).validate()
    with pytest.raises(NotImplementedError):
        UNetArch(strides=((2, 2),) * 8).validate()


def test_validate_rejects_unsupported():
    with pytest.raises(NotImplementedError):
        UNetArch(kernel_sizes=((5, 5),) * 8).validate()
    UNetArch(features_per_stage=(30, 64, 100, 256, 512, 512, 512, 512)).validate()        # any width (the engine rounds a stage up, zero weights)
    with pytest.raises(NotImplementedError):
        UNetArch(features_per_stage=(96, 128, 128, 256, 512, 512, 512, 512)).validate()    # the head kernel reads at most 64 channels
    with pytest.raises(ValueError):
        UNetArch(features_per_stage=(32, 0, 128, 256, 512, 512, 512, 512)).validate()
    with pytest.raises(ValueError):
        UNetArch(n_conv_per_stage=(2,) * 7).validate()


def test_from_plans():
    plans = {'configurations': {'2d': {'patch_size': [512, 512], 'spacing': [1.5, 1.5], 'architecture': {
        'network_class_name': 'dynamic_network_architectures.architectures.unet.PlainConvUNet',
        'arch_kwargs': {'n_stages': 3, 'features_per_stage': [32, 64, 128], 'conv_op': 'torch.nn.modules.conv.Conv2d',
                        'kernel_sizes': [[3, 3]] * 3, 'strides': [[1, 1], [2, 2], [2, 1]], 'n_conv_per_stage': [2, 2, 2],
                        'n_conv_per_stage_decoder': [2, 2], 'conv_bias': True, 'norm_op': 'torch.nn.modules.instancenorm.InstanceNorm2d',
                        'norm_op_kwargs': {'eps': 1e-05, 'affine': True}, 'dropout_op': None, 'dropout_op_kwargs': None,
                        'nonlin': 'torch.nn.LeakyReLU', 'nonlin_kwargs': {'inplace': True}}}}}}
    a = UNetArch.from_plans(plans, '2d', 2, 18)
    assert a.n_stages == 3 and tuple(a.features_per_stage) == (32, 64, 128) and a.num_classes == 18
    assert tuple(a.strides[2]) == (2, 1) and a.divisors == (4, 2)
    plans['configurations']['2d']['architecture']['network_class_name'] = 'x.ResidualEncoderUNet'
    with pytest.raises(NotImplementedError):
        UNetArch.from_plans(plans, '2d', 2, 18)


def test_prng_is_deterministic_and_normalish():
    a = prng.normal(3, 5, 100000)
    b = prng.normal(3, 5, 1000, offset=500)
    assert np.array_equal(a[500:1500], b)                     # counter-based: value i depends on (seed, stream, i) only
    assert abs(a.mean()) < 0.01 and abs(a.std() - 1) < 0.01
    # pinned values (any platform must reproduce them bit for bit)
    assert prng.hash_u64(1, 2, 2).tolist() == prng.hash_u64(1, 2, 2).tolist()
    v = prng.normal_f32(0, 0, (4,))
    assert v.dtype == np.float32 and np.array_equal(v, prng.normal_f32(0, 0, (4,)))


def test_blob_roundtrip():
    arch = cases.unet(2, (32, 64), 3, nconv=1)
    sd = weights.synthetic_state_dict(arch, 5)
    blob = weights.pack_blob(arch, sd)
    assert blob.size == arch.n_params()
    sd2 = weights.unpack_blob(arch, blob)
    assert all(np.array_equal(sd[k], sd2[k]) for k in sd)
    with pytest.raises(KeyError):
        weights.pack_blob(arch, {k: v for k, v in list(sd.items())[1:]})

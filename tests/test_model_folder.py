"""``initialize_from_trained_model_folder`` on a synthetic nnU-Net model directory written the way upstream
writes it (dataset.json, plans.json, fold_N/checkpoint_final.pth with 'network_weights', alias keys of the deep-supervised
training net, a 'module.' prefix, the mirroring axes and the configuration name) - the published Zenodo models are not available
offline, so the directory is generated here."""
import json
import os

import numpy as np
import pytest

from tests import cases
from totalsegmentator2d_amd import prng, weights
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd.predictor import HIPnnUNetPredictor
from tests.host_predictor import HostLogicPredictor

torch = pytest.importorskip('torch')


def _write_model_folder(root, arch, seeds, mirror, patch, module_prefix=False):
    labels = {'background': 0, **{f'organ_{i + 1}': i + 1 for i in range(arch.num_classes)}}
    with open(os.path.join(root, 'dataset.json'), 'w') as f:
        json.dump({'channel_names': {str(i): n for i, n in enumerate(['mean', 'max'][:arch.input_channels])}, 'labels': labels,
                   'file_ending': '.nrrd', 'multilabel': True, 'numTraining': 1}, f)
    n = arch.n_stages
    plans = {'plans_name': 'nnUNetPlans', 'transpose_forward': [0, 1, 2], 'transpose_backward': [0, 1, 2],
             'configurations': {'2d': {'patch_size': list(patch), 'spacing': [1.5, 1.5], 'normalization_schemes': ['ZScoreNormalization'] * 2,
                                       'use_mask_for_norm': [False, False],
                                       'architecture': {'network_class_name': 'dynamic_network_architectures.architectures.unet.PlainConvUNet',
                                                        'arch_kwargs': {'n_stages': n, 'features_per_stage': list(arch.features_per_stage),
                                                                        'conv_op': 'torch.nn.modules.conv.Conv2d', 'kernel_sizes': [[3, 3]] * n,
                                                                        'strides': [[1, 1]] + [[2, 2]] * (n - 1),
                                                                        'n_conv_per_stage': list(arch.n_conv_per_stage),
                                                                        'n_conv_per_stage_decoder': list(arch.n_conv_per_stage_decoder),
                                                                        'conv_bias': True, 'norm_op': 'torch.nn.modules.instancenorm.InstanceNorm2d',
                                                                        'norm_op_kwargs': {'eps': 1e-05, 'affine': True}, 'dropout_op': None,
                                                                        'dropout_op_kwargs': None, 'nonlin': 'torch.nn.LeakyReLU',
                                                                        'nonlin_kwargs': {'inplace': True}},
                                                        '_kw_requires_import': ['conv_op', 'norm_op', 'dropout_op', 'nonlin']}}}}
    with open(os.path.join(root, 'plans.json'), 'w') as f:
        json.dump(plans, f)
    blobs = []
    for fold, seed in enumerate(seeds):
        sd = weights.synthetic_state_dict(arch, seed)
        blobs.append(weights.pack_blob(arch, sd))
        full = {}
        for k, v in sd.items():
            t = torch.from_numpy(np.array(v))
            full[k] = t
            # aliases the training-time module tree registers for the same tensors
            if '.convs.' in k and k.startswith('encoder.'):
                full['decoder.' + k] = t                                                     # decoder.encoder.stages...
                full[k.replace('.conv.', '.all_modules.0.').replace('.norm.', '.all_modules.1.')] = t
            if '.convs.' in k and k.startswith('decoder.stages'):
                full[k.replace('.conv.', '.all_modules.0.').replace('.norm.', '.all_modules.1.')] = t
        for j in range(n - 2):                                                              # deep-supervision heads (unused at inference)
            f_j = arch.features_per_stage[n - 2 - j]
            full[f'decoder.seg_layers.{j}.weight'] = torch.zeros(arch.num_classes, f_j, 1, 1)
            full[f'decoder.seg_layers.{j}.bias'] = torch.zeros(arch.num_classes)
        if module_prefix and fold == 0:
            full = {'module.' + k: v for k, v in full.items()}
        os.makedirs(os.path.join(root, f'fold_{fold}'), exist_ok=True)
        torch.save({'network_weights': full, 'inference_allowed_mirroring_axes': tuple(mirror), 'trainer_name': 'nnUNetTrainer',
                    'init_args': {'configuration': '2d', 'fold': fold}}, os.path.join(root, f'fold_{fold}', 'checkpoint_final.pth'))
    return blobs, labels


def test_model_folder_is_read_like_upstream(tmp_path):
    arch, shape, patch, step, mirror, folds, seed = cases.SW_CASES['sw_folds_nomirror']
    blobs, labels = _write_model_folder(str(tmp_path), arch, [seed, seed + 1], (0, 1), patch, module_prefix=True)
    p = HostLogicPredictor(network=None, tile_step_size=step, use_mirroring=False)
    p.initialize_from_trained_model_folder(str(tmp_path), None, 'checkpoint_final.pth')        # folds auto-discovered
    assert p.arch == arch
    assert len(p.list_of_parameters) == 2
    for got, want in zip(p.list_of_parameters, blobs):
        assert np.array_equal(got, want)
    assert tuple(p.allowed_mirroring_axes) == (0, 1)
    assert list(p.configuration_manager.patch_size) == list(patch)
    assert p.dataset_json['labels'] == labels and p.dataset_json['multilabel'] is True
    assert p.plans_manager.transpose_forward == [0, 1, 2]
    # a single explicit fold, and a missing key is an error that names the key
    q = HostLogicPredictor(network=None)
    q.initialize_from_trained_model_folder(str(tmp_path), (1,), 'checkpoint_final.pth')
    assert len(q.list_of_parameters) == 1 and np.array_equal(q.list_of_parameters[0], blobs[1])
    ck = torch.load(os.path.join(str(tmp_path), 'fold_1', 'checkpoint_final.pth'), map_location='cpu', weights_only=False)
    del ck['network_weights']['decoder.transpconvs.0.weight']
    torch.save(ck, os.path.join(str(tmp_path), 'fold_1', 'checkpoint_final.pth'))
    with pytest.raises(KeyError, match='decoder.transpconvs.0.weight'):
        HostLogicPredictor(network=None).initialize_from_trained_model_folder(str(tmp_path), (1,), 'checkpoint_final.pth')


@pytest.mark.gpu
def test_model_folder_prediction_matches_manual_initialization(tmp_path):
    arch, shape, patch, step, mirror, folds, seed = cases.SW_CASES['sw_folds_nomirror']
    blobs, _ = _write_model_folder(str(tmp_path), arch, [seed, seed + 1], (0, 1), patch)
    data = prng.normal_f32(seed, 999, (arch.input_channels,) + tuple(shape))
    a = HIPnnUNetPredictor(tile_step_size=step, use_mirroring=True)
    a.initialize_from_trained_model_folder(str(tmp_path), (0, 1), 'checkpoint_final.pth')
    b = HIPnnUNetPredictor(tile_step_size=step, use_mirroring=True)
    b.manual_initialization(arch, blobs, patch, inference_allowed_mirroring_axes=(0, 1))
    try:
        ya = a.predict_logits_from_preprocessed_data(data).cpu().numpy()
        yb = b.predict_logits_from_preprocessed_data(data).cpu().numpy()
    finally:
        a.close(); b.close()
    assert ya.dtype == np.float16 and np.array_equal(ya, yb)
    # ... and the folder-loaded predictor against the ORACLE (not only against a second HIP predictor): two folds, four mirror passes,
    # float16 aggregation - the tolerance of the sliding-window goldens (2 half-ulps at |x| <= 8)
    from oracle import torch_oracle as O
    sds = [weights.synthetic_state_dict(arch, s) for s in (seed, seed + 1)]
    ref = O.predict_logits(arch, sds, data, patch, step, (0, 1)).numpy()
    assert ref.shape == ya.shape
    assert np.abs(ya.astype(np.float32) - ref.astype(np.float32)).max() <= 1.6e-2 and (ya != ref).mean() < 0.05

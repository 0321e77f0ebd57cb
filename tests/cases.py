"""Shared definitions of the parity cases (architectures, shapes, seeds) used by gen_golden.py and the tests."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from totalsegmentator2d_amd import prng                      # noqa: E402
from totalsegmentator2d_amd.arch import UNetArch             # noqa: E402


def unet(n_stages, feats, K, cin=2, nconv=2, nconv_dec=None, strides=None):
    nconv_dec = nconv if nconv_dec is None else nconv_dec
    strides = ((1, 1),) + ((2, 2),) * (n_stages - 1) if strides is None else tuple(tuple(s) for s in strides)
    return UNetArch(input_channels=cin, num_classes=K, n_stages=n_stages, features_per_stage=tuple(feats),
                    kernel_sizes=((3, 3),) * n_stages, strides=strides,
                    n_conv_per_stage=(nconv,) * n_stages, n_conv_per_stage_decoder=(nconv_dec,) * (n_stages - 1))


def make_input(arch, B, H, W, seed):
    """i.i.d. ~N(0,1) (post-z-score statistics), PRNG stream 1000 of `seed`."""
    return prng.normal_f32(seed, 1000, (B, arch.input_channels, H, W))


# name -> (arch, B, H, W, seed)
SMALL_CASES = {
    # minimal program: K1 (2->32), K3 (32->64 s2), K5 (convT 64->32), K6 (cat + 64->32), K7 (head): one conv per stage
    'k_min2': (unet(2, (32, 64), 3, nconv=1), 2, 32, 64, 11),
    # K2 stride-1 32->32 / 64->64 blocks, two convs per stage, odd batch
    'k_two3': (unet(3, (32, 64, 128), 5), 3, 32, 48, 12),
    # reduced-width whole net on 2x64x64 and 2x128x128
    'net5_64': (unet(5, (32, 32, 64, 64, 64), 18), 2, 64, 64, 13),
    'net5_128': (unet(5, (32, 64, 128, 256, 512), 18), 2, 128, 128, 14),
    # tiny bottleneck (2x2 / 4x4 images: several images per pixel tile), batch not a multiple of the image group
    'tiny_b37': (unet(4, (32, 64, 64, 96), 4), 37, 16, 32, 15),
    # single input channel (tsxr-like), 3 convs per encoder stage, 1 per decoder stage, non power-of-two extent
    'xr_1ch': (unet(3, (32, 64, 96), 26, cin=1, nconv=3, nconv_dec=1), 1, 96, 160, 16),
    # features[0] = 64 (head_1x1<64>, BN=64 everywhere), 3 input channels
    'wide64': (unet(3, (64, 64, 128), 7, cin=3), 1, 32, 96, 17),
    # per-axis strides (nnU-Net pools each axis separately: a plan may end in (2, 1) / (1, 2) stages); the transposed conv's kernel =
    # stride = the stride of the stage below.  Extents 32x64 -> 16x32 -> 8x16 -> 4x16 / 16x64 -> 16x32 -> 8x16 -> 8x16
    'aniso_21': (unet(4, (32, 64, 128, 128), 5, strides=((1, 1), (2, 2), (2, 2), (2, 1))), 1, 32, 64, 18),
    'aniso_12_11': (unet(4, (32, 64, 64, 96), 4, strides=((1, 1), (1, 2), (2, 2), (1, 1))), 1, 16, 64, 19),
}
KEEP_INTERMEDIATES = {'k_min2': True, 'k_two3': True, 'aniso_21': True, 'aniso_12_11': True}

# name -> (arch, data shape [Z,H,W], patch, step, mirror axes, folds, seed)
SW_CASES = {
    'sw_2tiles_mirror': (unet(3, (32, 32, 64), 4), (1, 80, 52), (64, 64), 0.5, (0, 1), 1, 21),
    'sw_folds_nomirror': (unet(3, (32, 32, 64), 3), (1, 100, 130), (64, 64), 0.5, None, 2, 22),
    'sw_z2_step1': (unet(2, (32, 32), 2), (2, 64, 96), (32, 32), 1.0, (1,), 1, 23),
}

"""Weight blob for the engine: synthetic (seeded) weights and real nnU-Net checkpoints.

Blob = the fp32 tensors of :meth:`UNetArch.param_specs` (PyTorch layouts) concatenated in program order.  The
C-ABI ``ts2d_engine_create`` (include/ts2d_engine.h) repacks that blob into its device layouts; Python never
needs to know them.

Reference behaviour mirrored here: ``nnUNetPredictor.initialize_from_trained_model_folder``
(called at ``ts2d/core/inference/nnu.py:165``) ``torch.load``s ``fold_N/checkpoint_final.pth`` and keeps
``checkpoint['network_weights']``; state-dict keys are ``encoder.stages.{s}.0.convs.{i}.{conv,norm}.{weight,bias}``,
``decoder.transpconvs.{j}.{weight,bias}``, ``decoder.stages.{j}.convs.{i}.{conv,norm}.{weight,bias}``,
``decoder.seg_layers.{j}.{weight,bias}`` plus alias duplicates (``all_modules.*``, ``decoder.encoder.*``) that are
ignored here.  PyTorch is used ONLY for unpickling the checkpoint (north_star: "PyTorch-ROCm only for weight
loading/checkpoint compat").
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

from .arch import UNetArch
from . import prng


def synthetic_state_dict(arch: UNetArch, seed: int = 1) -> Dict[str, np.ndarray]:
    """Deterministic weights: conv/convT He-normal with a=0.01 (nnU-Net ``InitWeights_He``),
    bias ~ N(0, 0.01^2), InstanceNorm gamma = 1 + N(0, 0.1^2), beta = N(0, 0.1^2) so that the affine and bias paths
    are exercised.  Tensor t of the blob uses PRNG stream t."""
    sd: Dict[str, np.ndarray] = {}
    gain = math.sqrt(2.0 / (1.0 + 0.01 ** 2))
    for t, (key, shp) in enumerate(arch.param_specs()):
        if key.endswith('norm.weight'):
            v = prng.normal_f32(seed, t, shp, mean=1.0, std=0.1)
        elif key.endswith('norm.bias'):
            v = prng.normal_f32(seed, t, shp, mean=0.0, std=0.1)
        elif key.endswith('bias'):
            v = prng.normal_f32(seed, t, shp, mean=0.0, std=0.01)
        else:
            # torch fan_in = size(1) * receptive field (also for ConvTranspose2d's [Cin,Cout,kh,kw])
            fan_in = shp[1] * shp[2] * shp[3]
            v = prng.normal_f32(seed, t, shp, mean=0.0, std=gain / math.sqrt(fan_in))
        sd[key] = v
    return sd


def pack_blob(arch: UNetArch, state_dict) -> np.ndarray:
    """state-dict (numpy arrays or torch tensors) -> contiguous fp32 blob in ``param_specs`` order."""
    parts = []
    for key, shp in arch.param_specs():
        if key not in state_dict:
            raise KeyError(f"checkpoint is missing '{key}'")
        v = state_dict[key]
        if hasattr(v, 'detach'):
            v = v.detach().cpu().numpy()
        v = np.asarray(v, dtype=np.float32)
        if tuple(v.shape) != tuple(shp):
            raise ValueError(f"'{key}': expected shape {tuple(shp)}, found {tuple(v.shape)}")
        parts.append(v.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts))


def unpack_blob(arch: UNetArch, blob: np.ndarray) -> Dict[str, np.ndarray]:
    sd, o = {}, 0
    for key, shp in arch.param_specs():
        n = int(np.prod(shp))
        sd[key] = blob[o:o + n].reshape(shp)
        o += n
    if o != blob.size:
        raise ValueError(f"blob has {blob.size} values, architecture needs {o}")
    return sd


def load_checkpoint(path: str):
    """Unpickle an nnU-Net ``checkpoint_*.pth`` -> (network_weights, inference_allowed_mirroring_axes, init_args)."""
    import torch
    ck = torch.load(path, map_location='cpu', weights_only=False)
    sd = ck['network_weights']
    sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
    return sd, ck.get('inference_allowed_mirroring_axes'), ck.get('init_args', {})

"""Exact vs split-fp16 mode: parity against the oracles and per-op timing at B=64."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights, prng
from totalsegmentator2d_amd.engine import Engine, unpack_mask
from oracle import torch_oracle as O, c_oracle as C
from tests import cases
import torch

for name in ('k_two3', 'net5_128', 'tiny_b37', 'wide64'):
    arch, B, H, W, seed = cases.SMALL_CASES[name]
    sd = weights.synthetic_state_dict(arch, seed); blob = weights.pack_blob(arch, sd)
    x = cases.make_input(arch, B, H, W, seed)
    g = np.load(f'tests/golden/{name}.npz')['logits']
    with Engine(arch, blob) as e:
        for mode in ('exact', 'split'):
            e.set_precision(mode)
            lg, _ = e.forward(x)
            print(f'{name:10s} {mode:6s} max|gpu-golden| = {np.abs(lg - g).max():.3e}', flush=True)

a = UNetArch.canonical()
blob = weights.pack_blob(a, weights.synthetic_state_dict(a, 1)); sd = weights.unpack_blob(a, blob)
x = prng.normal_f32(0, 0, (2, 2, 512, 512))
yt = O.unet_forward(a, sd, x).numpy()
truth = C.unet_forward(a, blob, x[:1], acc64=True)
with Engine(a, blob) as e:
    for mode in ('exact', 'split'):
        e.set_precision(mode)
        lg, mk = e.forward(x, logits=True, mask=True)
        print(f'canonical {mode}: max|gpu-torch| = {np.abs(lg - yt).max():.3e}  max|gpu-truth| = {np.abs(lg[:1] - truth).max():.3e}  (torch-truth {np.abs(yt[:1]-truth).max():.3e})', flush=True)
    xd = torch.randn(64, 2, 512, 512, device='cuda')
    for mode in ('exact', 'split'):
        e.set_precision(mode)
        e.forward(xd); torch.cuda.synchronize()
        t = time.time(); n = 5
        for _ in range(n): e.forward(xd)
        torch.cuda.synchronize(); dt = (time.time() - t) / n
        print(f'{mode}: B=64 {dt*1e3:.1f} ms/forward = {64/dt:.1f} slices/s', flush=True)
    e.set_profiling(True); e.forward(xd); torch.cuda.synchronize()
    for k, v in e.op_times().items():
        if not k.endswith('.stats') or v > 0.05: print(f'  {k:14s} {v:8.3f} ms')

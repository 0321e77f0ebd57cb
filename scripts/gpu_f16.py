"""fp16-storage mode: error vs the oracle and speed at B=64 / B=128."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights, prng
from totalsegmentator2d_amd.engine import Engine, unpack_mask
from oracle import torch_oracle as O
from tests import cases
import torch
for name in ('k_two3', 'net5_128', 'tiny_b37', 'wide64', 'xr_1ch'):
    arch, B, H, W, seed = cases.SMALL_CASES[name]
    sd = weights.synthetic_state_dict(arch, seed); blob = weights.pack_blob(arch, sd)
    x = cases.make_input(arch, B, H, W, seed)
    g = np.load(f'tests/golden/{name}.npz')['logits']
    with Engine(arch, blob) as e:
        e.set_precision('f16')
        lg, _ = e.forward(x)
        print(f'{name:10s} f16 max|gpu-golden| = {np.abs(lg - g).max():.3e}  rms {np.sqrt(((lg-g)**2).mean()):.3e}', flush=True)
a = UNetArch.canonical()
blob = weights.pack_blob(a, weights.synthetic_state_dict(a, 1)); sd = weights.unpack_blob(a, blob)
x = prng.normal_f32(0, 0, (2, 2, 512, 512))
yt = O.unet_forward(a, sd, x).numpy()
with Engine(a, blob) as e:
    e.set_precision('f16')
    lg, mk = e.forward(x, logits=True, mask=True)
    m_ref = O.logits_to_mask(yt).numpy()
    print(f'canonical f16: max|gpu-torch| = {np.abs(lg - yt).max():.3e} rms {np.sqrt(((lg-yt)**2).mean()):.3e}  mask disagreement {(unpack_mask(mk, 512) != m_ref).mean():.2e}', flush=True)
    for B in (64, 128):
        xd = torch.randn(B, 2, 512, 512, device='cuda')
        for mode in ('split', 'f16'):
            e.set_precision(mode)
            e.forward(xd, logits=False, mask=True); torch.cuda.synchronize()
            t = time.time(); n = 5
            for _ in range(n): e.forward(xd, logits=False, mask=True)
            torch.cuda.synchronize(); dt = (time.time() - t) / n
            print(f'{mode}: B={B} masks-only {dt*1e3:.1f} ms/forward = {B/dt:.1f} slices/s', flush=True)
    e.set_precision('f16'); e.set_profiling(True); e.forward(xd); torch.cuda.synchronize()
    ot = e.op_times()
    print(' '.join(f'{k}={v:.2f}' for k, v in ot.items() if not k.endswith('.stats')))

"""WS kernel check: parity vs goldens/oracle and timing, with per-op times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights, prng
from totalsegmentator2d_amd.engine import Engine
from oracle import torch_oracle as O
from tests import cases
import torch
for name in ('k_min2', 'k_two3', 'net5_128', 'net5_64', 'xr_1ch', 'wide64'):
    arch, B, H, W, seed = cases.SMALL_CASES[name]
    sd = weights.synthetic_state_dict(arch, seed); blob = weights.pack_blob(arch, sd)
    x = cases.make_input(arch, B, H, W, seed)
    g = np.load(f'tests/golden/{name}.npz')['logits']
    with Engine(arch, blob) as e:
        lg, _ = e.forward(x)
        print(f'{name:10s} max|gpu-golden| = {np.abs(lg - g).max():.3e}', flush=True)
a = UNetArch.canonical()
blob = weights.pack_blob(a, weights.synthetic_state_dict(a, 1)); sd = weights.unpack_blob(a, blob)
x = prng.normal_f32(0, 0, (2, 2, 512, 512))
yt = O.unet_forward(a, sd, x).numpy()
with Engine(a, blob) as e:
    lg, mk = e.forward(x, logits=True, mask=True)
    print(f'canonical: max|gpu-torch| = {np.abs(lg - yt).max():.3e}', flush=True)
    xd = torch.randn(64, 2, 512, 512, device='cuda')
    e.forward(xd); torch.cuda.synchronize()
    t = time.time(); n = 5
    for _ in range(n): e.forward(xd)
    torch.cuda.synchronize(); dt = (time.time() - t) / n
    print(f'B=64 {dt*1e3:.1f} ms/forward = {64/dt:.1f} slices/s', flush=True)
    e.set_profiling(True); e.forward(xd); torch.cuda.synchronize()
    ot = e.op_times()
    print(' '.join(f'{k}={v:.2f}' for k, v in ot.items() if not k.endswith('.stats')))
    print('stats total', sum(v for k, v in ot.items() if k.endswith('.stats')))

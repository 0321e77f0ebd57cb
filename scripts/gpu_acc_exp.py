import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights, prng
from totalsegmentator2d_amd.engine import Engine
from oracle import torch_oracle as O, c_oracle as C
import torch
a = UNetArch.canonical()
blob = weights.pack_blob(a, weights.synthetic_state_dict(a, 1)); sd = weights.unpack_blob(a, blob)
x = prng.normal_f32(0, 0, (4, 2, 512, 512))
yt = O.unet_forward(a, sd, x).numpy()
truth = C.unet_forward(a, blob, x[:2], acc64=True)
with Engine(a, blob) as e:
    lg, _ = e.forward(x)
    print(f'no-chunk-acc split: max|gpu-torch| = {np.abs(lg - yt).max():.3e}  max|gpu-truth| = {np.abs(lg[:2] - truth).max():.3e}  (torch-truth {np.abs(yt[:2]-truth).max():.3e})', flush=True)
    xd = torch.randn(64, 2, 512, 512, device='cuda')
    e.forward(xd); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5): e.forward(xd)
    torch.cuda.synchronize(); print('ms', (time.time() - t) / 5 * 1e3)

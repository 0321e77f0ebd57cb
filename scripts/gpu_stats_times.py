import os, sys
sys.path.insert(0, '/root/repo')
import torch
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine
a = UNetArch.canonical()
blob = weights.pack_blob(a, weights.synthetic_state_dict(a, 1))
with Engine(a, blob) as e:
    xd = torch.randn(64, 2, 512, 512, device='cuda')
    for _ in range(2): e.forward(xd)
    torch.cuda.synchronize(); e.set_profiling(True)
    tot = {}
    for _ in range(3):
        e.forward(xd); torch.cuda.synchronize()
        for k, v in e.op_times().items(): tot[k] = tot.get(k, 0.0) + v / 3
    print(' '.join(f'{k}={v*1000:.0f}us' for k, v in tot.items() if k.endswith('.stats')))

import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd.engine import Engine
a = UNetArch.canonical()
blob = (np.random.default_rng(0).standard_normal(a.n_params()) * 0.02).astype(np.float32)
with Engine(a, blob) as e:
    xd = torch.randn(64, 2, 512, 512, device='cuda')
    e.forward(xd); torch.cuda.synchronize()
    e.set_profiling(True); e.forward(xd); torch.cuda.synchronize(); e.forward(xd); torch.cuda.synchronize()
    ot = e.op_times()
    keys = ['enc0.c1', 'enc1.c1', 'enc3.c1', 'dec4.c0', 'dec2.c0', 'dec1.c0', 'dec0.c0', 'dec0.c1']
    print(os.environ.get('TS2D_ABL', '0'), ' '.join(f'{k}={ot[k]:.2f}' for k in keys), 'total', round(sum(ot.values()), 1))

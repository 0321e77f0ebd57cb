"""Debug helper: per-layer error of the engine vs the torch oracle (which layer / channels / pixels are off).
    python scripts/gpu_debug_layer.py MODE H W B FEATS(comma) [opt=val,...]      e.g.  split 56 128 4 32,32,32,64 flex=1"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine
from oracle import torch_oracle as O

mode = sys.argv[1] if len(sys.argv) > 1 else 'split'
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 64)
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
feats = tuple(int(v) for v in sys.argv[5].split(',')) if len(sys.argv) > 5 else (32, 64)
opts = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in sys.argv[6].split(',')} if len(sys.argv) > 6 else {}
arch = cases.unet(len(feats), feats, 3)
sd = weights.synthetic_state_dict(arch, 5); blob = weights.pack_blob(arch, sd)
x = cases.make_input(arch, B, H, W, 5)
y, inter = O.unet_forward(arch, sd, x, return_intermediates=True)
with Engine(arch, blob, options=opts) as e:
    e.set_precision(mode)
    e.set_profiling(True)
    lg, _ = e.forward(x)
    kern = e.op_kernels()
    for name in inter:
        if not e.materialised(name):
            print(f'{name:8s} (not materialised)')
            continue
        t = e.debug_tensor(name)
        r = inter[name].numpy()
        d = np.abs(np.nan_to_num(t, nan=1e9, posinf=1e9, neginf=1e9) - r)
        print(f'{name:8s} {kern.get(name, "?"):24s} {t.shape} max err {d.max():.3e}  mean {d.mean():.3e}', end='')
        if d.max() > 1e-3:
            bad = np.argwhere(d > 1e-3)
            print(f'  bad {len(bad)}/{d.size}; images {np.unique(bad[:, 0])} channels {len(np.unique(bad[:, 1]))}')
            print('   rows with errors', np.unique(bad[:, 2])[:40], ' cols', np.unique(bad[:, 3])[:70])
        else:
            print()
    print('logits', np.abs(lg - y.numpy()).max())

"""Debug helper: per-layer error of the engine vs the torch oracle on a small net (which channels / pixels are off)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine
from oracle import torch_oracle as O

mode = sys.argv[1] if len(sys.argv) > 1 else 'split'
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16, 64)
arch = cases.unet(2, (32, 64), 3)
sd = weights.synthetic_state_dict(arch, 5); blob = weights.pack_blob(arch, sd)
x = cases.make_input(arch, 2, H, W, 5)
y, inter = O.unet_forward(arch, sd, x, return_intermediates=True)
with Engine(arch, blob) as e:
    e.set_precision(mode)
    import torch
    lgt, _ = e.forward(torch.from_numpy(x).cuda()); torch.cuda.synchronize(); lg = lgt.cpu().numpy()      # (device path: no automatic finite check)
    for name in ('enc0.c0', 'enc0.c1', 'enc1.c0', 'enc1.c1', 'dec0.c0', 'dec0.c1'):
        t = e.debug_tensor(name)
        r = inter[name].numpy()
        d = np.abs(np.nan_to_num(t, nan=1e9, posinf=1e9, neginf=1e9) - r)
        print(f'{name:8s} max err {d.max():.3e}  mean {d.mean():.3e}', end='')
        if d.max() > 1e-3:
            bad = np.argwhere(d > 1e-3)
            print(f'  bad {len(bad)}/{d.size}; per-channel bad counts {np.bincount(bad[:,1], minlength=t.shape[1])[:32]}')
            print('   rows with errors', np.unique(bad[:, 2])[:40], ' cols', np.unique(bad[:, 3])[:70])
            print('   sample', t[0, :4, 0, :4], '\n   ref', r[0, :4, 0, :4])
        else:
            print()
    print('logits', np.abs(lg - y.numpy()).max())

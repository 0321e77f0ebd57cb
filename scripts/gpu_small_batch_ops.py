"""Per-op kernel times of small batches (what TS2D.predict runs: tiles x mirrors = 8 slices per sub-model).  argv: [B ...] (default 1 8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine
from totalsegmentator2d_amd import _lib as _L
if os.environ.get('TS2D_AB_LIB'):
    _L.LIB_PATH = os.path.abspath(os.environ['TS2D_AB_LIB'])
Bs = [int(v) for v in sys.argv[1:]] or [1, 8]
a = UNetArch.canonical(num_classes=18)
opts = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in os.environ.get('SB_OPTS', '').split(',') if kv}
e = Engine(a, weights.pack_blob(a, weights.synthetic_state_dict(a, 1)), options=opts)
print('options', opts, flush=True)
for mode in ('split', 'f16'):
    e.set_precision(mode)
    for B in Bs:
        x = torch.randn(B, 2, 512, 512, device='cuda')
        for _ in range(3): e.forward(x, logits=True, mask=False)
        torch.cuda.synchronize()
        e.set_profiling(True)
        tot = {}
        for _ in range(5):
            e.forward(x, logits=True, mask=False); torch.cuda.synchronize()
            for k, v in e.op_times().items(): tot[k] = tot.get(k, 0.0) + v / 5
        e.set_profiling(False)
        kern = e.op_kernels()
        st = sum(v for k, v in tot.items() if k.endswith('.stats'))
        print(f'[{mode} B={B}] total {sum(tot.values()):.3f} ms (stats launches {st:.3f}): ' +
              ' '.join(f'{k}={v:.3f}' for k, v in tot.items() if not k.endswith('.stats')), flush=True)
        print('    kernels: ' + ' '.join(f'{k}:{kern[k]}' for k in tot if not k.endswith('.stats')), flush=True)
e.close()

"""Per-op times of one B=64 canonical forward (no parity check).  argv: [mode [ops [option=value,...]]]; TS2D_DBG selects ablations / stamps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine
from totalsegmentator2d_amd import _lib as _L
if os.environ.get('TS2D_AB_LIB'):      # script-only hook: time another build of the library (A/B across processes on one box)
    _L.LIB_PATH = os.path.abspath(os.environ['TS2D_AB_LIB'])
a = UNetArch.canonical()
blob = weights.pack_blob(a, weights.synthetic_state_dict(a, 1))
mode = sys.argv[1] if len(sys.argv) > 1 else 'split'
sel = sys.argv[2].split(',') if len(sys.argv) > 2 and sys.argv[2] != '-' else None
opts = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in sys.argv[3].split(',')} if len(sys.argv) > 3 else {}
with Engine(a, blob, options=opts) as e:
    e.set_precision(mode)
    xd = torch.randn(64, 2, 512, 512, device='cuda')
    for _ in range(2): e.forward(xd)
    torch.cuda.synchronize()
    e.set_profiling(True)
    tot = {}
    for _ in range(3):
        e.forward(xd); torch.cuda.synchronize()
        for k, v in e.op_times().items(): tot[k] = tot.get(k, 0.0) + v / 3
    print(f'[{mode} DBG={os.environ.get("TS2D_DBG", 0)} {opts}] total {sum(tot.values()):.2f} ms: ' +
          ' '.join(f'{k}={v:.2f}' for k, v in tot.items() if (sel is None and not k.endswith('.stats')) or (sel and k in sel)))

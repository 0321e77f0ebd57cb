"""Latency of small batches (device-resident input, logits out): wall time vs the sum of the kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine
from totalsegmentator2d_amd import _lib as _L
if os.environ.get('TS2D_AB_LIB'):      # script-only hook: time another build of the library (A/B across processes on one box)
    _L.LIB_PATH = os.path.abspath(os.environ['TS2D_AB_LIB'])

a = UNetArch.canonical(num_classes=18)
opts = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in os.environ.get('SB_OPTS', '').split(',') if kv}
e = Engine(a, weights.pack_blob(a, weights.synthetic_state_dict(a, 1)), options=opts)
print('options', opts, flush=True)
for mode in ('split', 'f16'):
    e.set_precision(mode)
    for B in (1, 2, 4, 8, 16):
        x = torch.randn(B, 2, 512, 512, device='cuda')
        e.reserve(B, 512, 512)
        def fwd():
            e.forward(x, logits=True, mask=False); torch.cuda.synchronize()
        fwd(); fwd(); t = time.time(); n = 50
        for _ in range(n): fwd()
        wall = (time.time() - t) / n * 1e3
        e.set_profiling(True); e.forward(x, logits=True, mask=False); torch.cuda.synchronize()
        ot = e.op_times(); e.set_profiling(False)
        top = sorted(ot.items(), key=lambda kv: -kv[1])[:4]
        print(f'{mode} B={B:2d}: wall {wall:6.2f} ms, kernels {sum(ot.values()):6.2f} ms over {len(ot)} launches; top {[(k, round(v, 3)) for k, v in top]}', flush=True)
e.close()

"""Per-slice time of the HBM-heavy layers against the batch size: does a producer's output that still sits in the 256-MB Infinity
Cache make its consumer faster (sub-batching the shallow levels)?    python scripts/gpu_batch_sweep.py [split|f16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine
a = UNetArch.canonical()
blob = weights.pack_blob(a, weights.synthetic_state_dict(a, 1))
mode = sys.argv[1] if len(sys.argv) > 1 else 'split'
ops = ['enc0.c0', 'enc0.c1', 'enc1.c0', 'enc1.c1', 'enc2.c0', 'dec2.c0', 'dec1.c0', 'dec1.c1', 'dec0.c0', 'dec0.c1', 'head']
with Engine(a, blob) as e:
    e.set_precision(mode)
    for B in (1, 2, 4, 8, 16, 32, 64):
        xd = torch.randn(B, 2, 512, 512, device='cuda')
        for _ in range(3): e.forward(xd)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        n = max(3, 128 // B)
        for _ in range(n): e.forward(xd)
        torch.cuda.synchronize()
        whole = (time.perf_counter() - t0) / n * 1e3
        e.set_profiling(True)
        tot = {}
        for _ in range(5):
            e.forward(xd); torch.cuda.synchronize()
            for k, v in e.op_times().items(): tot[k] = tot.get(k, 0.0) + v / 5
        e.set_profiling(False)
        print(f'[{mode}] B={B:2d} whole {whole:7.3f} ms = {whole / B * 1e3:7.1f} us/slice | us/slice: ' + ' '.join(f'{k}={tot[k] / B * 1e3:.1f}' for k in ops), flush=True)

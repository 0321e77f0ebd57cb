"""Throughput of BASELINE configs 3 and 5 on one MI355X (synthetic weights/inputs, inputs resident in HBM):
config 3 = full ts2d-v2 (5 sub-models, K = 18/23/24/26/26 -> 117 packed mask channels), batch 128, 16-bit mode;
config 5 = tsxr geometry (1-channel 1024x1024, 9 stages, K = 26), 16-bit mode (one GPU's share of the 8-GPU config)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n


dev = torch.device('cuda', 0)
# ---- config 3
B = 128
x = torch.randn(B, 2, 512, 512, device=dev)
engines = []
for i, K in enumerate((18, 23, 24, 26, 26)):
    a = UNetArch.canonical(num_classes=K)
    e = Engine(a, weights.pack_blob(a, weights.synthetic_state_dict(a, i + 1)))
    e.reserve(B, 512, 512)
    engines.append((e, torch.empty(B, K, 512, 512 // 32, dtype=torch.int32, device=dev)))
for mode in ('f16', 'split'):
    for e, _ in engines: e.set_precision(mode)
    def run():
        for e, m in engines: e.forward(x, logits=False, mask=True, out_mask=m)
    dt = timed(run)
    print(f'config 3 ({mode}): 5 sub-models, B={B}, 117 mask channels: {dt*1e3:.1f} ms per batch = {B/dt:.1f} slices/s '
          f'({5*B/dt:.0f} sub-model forwards/s)', flush=True)
for e, _ in engines: e.close()
del engines
# ---- config 5
a = UNetArch.canonical(input_channels=1, num_classes=26, n_stages=9)
e = Engine(a, weights.pack_blob(a, weights.synthetic_state_dict(a, 7)))
for B in (16, 32):
    x = torch.randn(B, 1, 1024, 1024, device=dev)
    m = torch.empty(B, 26, 1024, 1024 // 32, dtype=torch.int32, device=dev)
    e.reserve(B, 1024, 1024)
    for mode in ('f16', 'split'):
        e.set_precision(mode)
        dt = timed(lambda: e.forward(x, logits=False, mask=True, out_mask=m))
        w = a.work(1024, 1024)
        print(f'config 5 ({mode}): 1x1024x1024, 9 stages, K=26, B={B}: {dt*1e3:.1f} ms = {B/dt:.1f} images/s = {B/dt*w["flops"]/1e12:.0f} TFLOP/s', flush=True)
e.close()

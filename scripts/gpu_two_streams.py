"""Experiment: does running two half-batches concurrently on two HIP streams (HBM-bound layers of one overlapping the
MFMA-bound layers of the other) beat one full batch?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine

a = UNetArch.canonical(num_classes=18)
blob = weights.pack_blob(a, weights.synthetic_state_dict(a, 1))
dev = torch.device('cuda', 0)
B = 64
x = torch.randn(B, 2, 512, 512, device=dev)
lg = torch.empty(B, 18, 512, 512, device=dev); mk = torch.empty(B, 18, 512, 16, dtype=torch.int32, device=dev)
e0 = Engine(a, blob); e0.reserve(B, 512, 512)
def one():
    e0.forward(x, logits=True, mask=True, out_logits=lg, out_mask=mk)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
print(f'one engine, B=64: {timed(one):.2f} ms', flush=True)
for parts in (2, 4):
    es = [Engine(a, blob) for _ in range(parts)]
    ss = [torch.cuda.Stream(device=dev) for _ in range(parts)]
    nb = B // parts
    for e in es: e.reserve(nb, 512, 512)
    xs = [x[i * nb:(i + 1) * nb].contiguous() for i in range(parts)]
    ls = [lg[i * nb:(i + 1) * nb] for i in range(parts)]; ms = [mk[i * nb:(i + 1) * nb] for i in range(parts)]
    def many():
        for e, s, xi, li, mi in zip(es, ss, xs, ls, ms):
            e.forward(xi, logits=True, mask=True, out_logits=li, out_mask=mi, stream=s.cuda_stream)
    # staggered start: the second stream begins half a network later, so that different layer types overlap
    print(f'{parts} engines x B={nb} on {parts} streams: {timed(many):.2f} ms', flush=True)
    for e in es: e.close()

# staggered steady state: stream B starts `delay` ms after stream A, then both run back to back
es = [Engine(a, blob) for _ in range(2)]
ss = [torch.cuda.Stream(device=dev) for _ in range(2)]
nb = B // 2
for e in es: e.reserve(nb, 512, 512)
xs = [x[i * nb:(i + 1) * nb].contiguous() for i in range(2)]
ls = [lg[i * nb:(i + 1) * nb] for i in range(2)]; ms = [mk[i * nb:(i + 1) * nb] for i in range(2)]
def fwd(i):
    es[i].forward(xs[i], logits=True, mask=True, out_logits=ls[i], out_mask=ms[i], stream=ss[i].cuda_stream)
for delay in (0.0, 3.0, 7.0, 10.0):
    fwd(0); fwd(1); torch.cuda.synchronize()
    n = 12
    t = time.time()
    fwd(0)
    if delay: time.sleep(delay * 1e-3)
    for _ in range(n - 1):
        fwd(1); fwd(0)
    fwd(1)
    torch.cuda.synchronize()
    dt = (time.time() - t) / n * 1e3
    print(f'staggered by {delay:.0f} ms: {dt:.2f} ms per 64 slices', flush=True)

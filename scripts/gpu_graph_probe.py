"""Does a hipGraph help the small-batch forward?  Eager launches vs torch.cuda.CUDAGraph replay of the same forward (B = 1, 2, 8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine

a = UNetArch.canonical(num_classes=18)
e = Engine(a, weights.pack_blob(a, weights.synthetic_state_dict(a, 1)))
for mode in ('split', 'f16'):
    e.set_precision(mode)
    for B in (1, 2, 8):
        x = torch.randn(B, 2, 512, 512, device='cuda')
        lg = torch.empty(B, 18, 512, 512, device='cuda')
        e.reserve(B, 512, 512)
        for _ in range(3): e.forward(x, out_logits=lg)
        torch.cuda.synchronize()
        ref = lg.clone()
        n = 50
        t = time.time()
        for _ in range(n): e.forward(x, out_logits=lg)
        torch.cuda.synchronize()
        eager = (time.time() - t) / n * 1e3
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.stream(s):
                for _ in range(2): e.forward(x, out_logits=lg)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                e.forward(x, out_logits=lg)
            torch.cuda.synchronize()
            lg.zero_()
            g.replay(); torch.cuda.synchronize()
            same = bool(torch.equal(lg, ref))
            t = time.time()
            for _ in range(n): g.replay()
            torch.cuda.synchronize()
            graph = (time.time() - t) / n * 1e3
            print(f'{mode} B={B}: eager {eager:.3f} ms, graph replay {graph:.3f} ms, identical {same}', flush=True)
        except Exception as ex:
            print(f'{mode} B={B}: eager {eager:.3f} ms, graph capture failed: {ex}', flush=True)
            torch.cuda.synchronize()
e.close()

"""Interleaved A/B of kernel variants in ONE process on ONE device (guide rule 24): engines created with different dispatch options
(ts2d_engine_set_option), B=64 canonical forwards alternated for several rounds, per-op HIP-event times.

    python scripts/gpu_ab.py - fuse0=0 [--ops enc0.c0,enc0.c1] [--rounds 5] [--mode split]
Each positional argument is one variant: comma-separated option=value pairs ('-' = defaults).  Prints per variant the whole-forward
time (median / min over rounds, un-profiled) and the per-op medians of the selected ops, plus max|logits A - logits B| vs variant 0."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine
from totalsegmentator2d_amd import _lib as _L
if os.environ.get('TS2D_AB_LIB'):      # script-only hook: time another build of the library (A/B across processes on one box)
    _L.LIB_PATH = os.path.abspath(os.environ['TS2D_AB_LIB'])

ap = argparse.ArgumentParser()
ap.add_argument('variants', nargs='+')
ap.add_argument('--ops', default=None)
ap.add_argument('--rounds', type=int, default=5)
ap.add_argument('--mode', default='split')
ap.add_argument('--batch', type=int, default=64)
args = ap.parse_args()
a = UNetArch.canonical()
blob = weights.pack_blob(a, weights.synthetic_state_dict(a, 1))
engines = []
for v in args.variants:
    opts = {} if v == '-' else {kv.split('=')[0]: int(kv.split('=')[1]) for kv in v.split(',')}
    dbg = opts.pop('dbg', None)                  # pseudo-option: TS2D_DBG (diagnostic switches inside kernels), read at engine creation
    if dbg is not None:
        os.environ['TS2D_DBG'] = str(dbg)
    e = Engine(a, blob, options=opts)
    os.environ.pop('TS2D_DBG', None)
    e.set_precision(args.mode)
    engines.append(e)
xd = torch.randn(args.batch, 2, 512, 512, device='cuda')
lg = torch.empty(args.batch, 18, 512, 512, device='cuda')
outs = []
for e in engines:
    e.forward(xd, out_logits=lg); torch.cuda.synchronize()
    outs.append(lg[:2].cpu().numpy().copy())
whole = [[] for _ in engines]; per = [{} for _ in engines]
for r in range(args.rounds):
    for i, e in enumerate(engines):
        e.set_profiling(False)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream()
        t0.record(st)
        for _ in range(3): e.forward(xd, out_logits=lg)
        t1.record(st); torch.cuda.synchronize()
        whole[i].append(t0.elapsed_time(t1) / 3)
        e.set_profiling(True)
        e.forward(xd, out_logits=lg); torch.cuda.synchronize()
        for k, v in e.op_times().items(): per[i].setdefault(k, []).append(v)
sel = args.ops.split(',') if args.ops else None
for i, v in enumerate(args.variants):
    ks = [k for k in per[i] if (sel is None and not k.endswith('.stats')) or (sel and k in sel)]
    kern = engines[i].op_kernels()
    print(f'[{v}] forward median {np.median(whole[i]):.3f} ms  min {min(whole[i]):.3f} ms  ({args.batch / np.median(whole[i]) * 1e3:.0f} slices/s)  '
          f'max|logits - variant0| {np.abs(outs[i] - outs[0]).max():.2e}')
    print('    ' + ' '.join(f'{k}={np.median(per[i][k]):.3f}' for k in ks))
    if sel:
        print('    kernels: ' + ' '.join(f'{k}:{kern.get(k, "?")}' for k in ks))
for e in engines: e.close()

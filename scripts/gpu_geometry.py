"""Throughput and kernel dispatch on extents other than 512 x 512: the reference runs whatever patch size and
pooling plans.json names (ts2d/core/inference/prediction_worker.py:76-77, nnu.py:164-165).

    python scripts/gpu_geometry.py 512x512 640x384 448x576:7 640x320:8:21 [--modes split,f16] [--batch 64] [--check] [--ops]
A geometry is HxW[:n_stages[:SS]] - SS = the last stage's stride, e.g. 21 = (2, 1); default: the canonical 8 stages of (2, 2).
Prints slices/s, Mpixel/s and the kernel that served every op (--ops: also its HIP-event time); --check compares a B=1 forward
with the torch oracle."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine


def parse_geometry(g, K=18, cin=2):
    parts = g.split(':')
    H, W = (int(v) for v in parts[0].split('x'))
    n = int(parts[1]) if len(parts) > 1 else 8
    a = UNetArch.canonical(input_channels=cin, num_classes=K, n_stages=n)
    if len(parts) > 2:
        a.strides = tuple(a.strides[:-1]) + ((int(parts[2][0]), int(parts[2][1])),)
    return a, H, W


def measure(a, blob, H, W, B, mode, rounds=3, reps=3, opts=None):
    """-> dict(ms, slices_per_s, mpixel_per_s, kernels {op: kernel}, op_ms {op: ms})."""
    with Engine(a, blob, options=opts or {}) as e:
        e.set_precision(mode)
        x = torch.randn(B, a.input_channels, H, W, device='cuda')
        lg = torch.empty(B, a.num_classes, H, W, device='cuda')
        for _ in range(2):
            e.forward(x, out_logits=lg)
        torch.cuda.synchronize()
        ts = []
        for _ in range(rounds):
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            st = torch.cuda.current_stream()
            t0.record(st)
            for _ in range(reps):
                e.forward(x, out_logits=lg)
            t1.record(st); torch.cuda.synchronize()
            ts.append(t0.elapsed_time(t1) / reps)
        e.set_profiling(True)
        e.forward(x, out_logits=lg); torch.cuda.synchronize()
        op_ms, kern = e.op_times(), e.op_kernels()
    ms = float(np.median(ts))
    return dict(ms=ms, slices_per_s=B / ms * 1e3, mpixel_per_s=B * H * W / ms * 1e-3, kernels=kern, op_ms=op_ms)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('geometries', nargs='+')
    ap.add_argument('--modes', default='split,f16')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--check', action='store_true')
    ap.add_argument('--ops', action='store_true')
    ap.add_argument('--opts', default='')
    args = ap.parse_args()
    opts = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in args.opts.split(',')} if args.opts else {}
    blobs = {}
    for g in args.geometries:
        a, H, W = parse_geometry(g)
        key = repr(a)
        if key not in blobs:
            sd = weights.synthetic_state_dict(a, 1)
            blobs[key] = (sd, weights.pack_blob(a, sd))
        sd, blob = blobs[key]
        for mode in args.modes.split(','):
            r = measure(a, blob, H, W, args.batch, mode, opts=opts)
            gf = a.work(H, W)['flops'] * args.batch / r['ms'] * 1e-9
            print(f"[{g} {mode} B={args.batch}] {r['ms']:.2f} ms  {r['slices_per_s']:.0f} slices/s  {r['mpixel_per_s']:.1f} Mpixel/s  {gf:.0f} TFLOP/s", flush=True)
            ks = [k for k in r['kernels'] if not k.endswith('.stats')]
            if args.ops:
                print('    ' + ' '.join(f"{k}:{r['kernels'][k]}={r['op_ms'][k]:.3f}" for k in ks), flush=True)
            else:
                print('    ' + ' '.join(f"{k}:{r['kernels'][k]}" for k in ks), flush=True)
            if args.check:
                from oracle import torch_oracle as O
                x1 = np.random.RandomState(3).randn(1, a.input_channels, H, W).astype(np.float32)
                with Engine(a, blob, options=opts) as e:
                    e.set_precision(mode)
                    lg, _ = e.forward(x1)
                ref = O.unet_forward(a, sd, x1, emulate='f16' if mode == 'f16' else None).numpy()
                d = lg - ref
                print(f"    check vs oracle: max {np.abs(d).max():.3e} rms {np.sqrt((d ** 2).mean()):.3e}", flush=True)

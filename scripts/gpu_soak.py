"""Soak: 60 s of forwards in random precision modes and batch sizes, interleaved with device-side tiled predictions, consuming the
outputs with torch ops WITHOUT explicit synchronisation; slice 0 of every (mode, batch size) must come out bit-identical every time (round 6: with
the small-batch dispatch on, batches of different size may take different kernels at the deep levels - the same batch size never does)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.engine import Engine
from totalsegmentator2d_amd import sliding_window as SW

a = UNetArch.canonical(num_classes=18)
e = Engine(a, weights.pack_blob(a, weights.synthetic_state_dict(a, 1)))
rng = np.random.default_rng(0)
x = torch.randn(96, 2, 512, 512, device='cuda')
g = SW.compute_gaussian((512, 512)).astype(np.float16)
img = rng.standard_normal((2, 700, 600)).astype(np.float32)
ref, tref = {}, None
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[1] if len(sys.argv) > 1 else 60):
    mode = ('split', 'f16', 'exact')[int(rng.integers(0, 3))]
    B = int(rng.choice([1, 2, 5, 8, 17, 33, 64, 96]))
    e.set_precision(mode)
    lg, mk = e.forward(x[:B], logits=True, mask=True)
    cur = (lg[0].clone(), mk[0].clone())                    # consumed on torch's stream, no synchronise
    if (mode, B) in ref:
        assert torch.equal(ref[mode, B][0], cur[0]) and torch.equal(ref[mode, B][1], cur[1]), (mode, B, n)
    else:
        ref[mode, B] = cur
        if mode != 'f16' and (mode, 64) in ref and B != 64:      # across batch sizes: summation order only
            assert float((ref[mode, 64][0] - cur[0]).abs().max()) <= 3e-5, (mode, B, n)
    if n % 7 == 0:
        e.set_precision('split')
        o, s = e.predict_tiled(img, (512, 512), [(0, 0), (188, 0), (0, 88), (188, 88)], (0, 1), g, True, True)
        if tref is None: tref = (o.copy(), s.copy())
        assert np.array_equal(o, tref[0]) and np.array_equal(s, tref[1]), ('tiled', n)
    n += 1
torch.cuda.synchronize()
print(f'soak ok: {n} forwards; device bytes {e.lib.ts2d_engine_device_bytes(e._h) / 1e9:.1f} GB')
e.close()

"""Re-run one case of scripts/gpu_fuzz_parity.py (default mode) by its parameters and report per mode / per layer where the engine leaves the oracle.
    python scripts/gpu_repro_case.py T FEATS(comma) K CIN H W B NCONV [opt=val,...]     (T = the case index: weight seed 100 + T, input seed 200 + T)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from totalsegmentator2d_amd import weights, prng
from totalsegmentator2d_amd.engine import Engine
from oracle import torch_oracle as O

t = int(sys.argv[1]); feats = [int(v) for v in sys.argv[2].split(',')]
K, cin, H, W, B, nconv = (int(v) for v in sys.argv[3:9])
opts = {kv.split('=')[0]: int(kv.split('=')[1]) for kv in sys.argv[9].split(',')} if len(sys.argv) > 9 else {}
arch = cases.unet(len(feats), feats, K, cin=cin, nconv=nconv)
sd = weights.synthetic_state_dict(arch, 100 + t); blob = weights.pack_blob(arch, sd)
x = prng.normal_f32(200 + t, 1, (B, cin, H, W))
ref, inter = O.unet_forward(arch, sd, x, return_intermediates=True)
for mode in ('split', 'exact', 'f16'):
    with Engine(arch, blob, options=opts) as e:
        e.set_precision(mode)
        e.keep_activations(True)
        e.set_profiling(True)
        try:
            lg, _ = e.forward(x)
            print(mode, 'logits err', float(np.abs(lg - ref.numpy()).max()))
        except RuntimeError as ex:
            print(mode, 'FAILED:', str(ex)[:160])
        kern = e.op_kernels()
        for name in inter:
            if not e.materialised(name):
                print(f'   {name:8s} (not materialised) {kern.get(name, "")}')
                continue
            try:
                got = e.debug_tensor(name)
            except RuntimeError as ex:
                print(f'   {name:8s} unreadable: {str(ex)[:100]}'); continue
            want = inter[name].numpy()
            bad = ~np.isfinite(got)
            d = np.abs(np.where(bad, 0, got) - want)
            print(f'   {name:8s} {kern.get(name, "?"):26s} {got.shape} max err {d.max():.3e} nonfinite {int(bad.sum())} max|want| {np.abs(want).max():.3e}')

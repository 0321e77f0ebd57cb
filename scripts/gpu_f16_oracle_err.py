"""Calibration of the 16-bit mode's test bounds: engine (precision 'f16') against the 16-BIT oracle (oracle/torch_oracle.py,
emulate='f16') and against the fp32 oracle - end to end on every case the f16 tests use, and per layer (each block fed with the
ENGINE's own inputs of that block) on a canonical slice.  Prints one line per case; the bounds in tests/test_gpu_parity.py
(F16E_*, F16_LAYER_*) are ~2x the worst figures printed here."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import cases
from oracle import torch_oracle as O
from totalsegmentator2d_amd import weights
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd.engine import Engine, unpack_mask


def stats(a, b):
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return float(np.abs(d).max()), float(np.sqrt((d ** 2).mean()))


runs = [(n,) + cases.SMALL_CASES[n] for n in ('k_two3', 'net5_128', 'wide64', 'xr_1ch', 'tiny_b37')]
runs += [('canon_k26', UNetArch.canonical(num_classes=26), 1, 512, 512, 4),
         ('canon_k18', UNetArch.canonical(num_classes=18), 1, 512, 512, 1),
         ('tsxr_1024', UNetArch.canonical(input_channels=1, num_classes=26, n_stages=9), 1, 1024, 1024, 7)]
worst_e2e, worst_layer = (0.0, 0.0), (0.0, 0.0)
for name, arch, B, H, W, seed in runs:
    sd = weights.synthetic_state_dict(arch, seed)
    blob = weights.pack_blob(arch, sd)
    x = cases.make_input(arch, B, H, W, seed)
    ref32 = O.unet_forward(arch, sd, x).numpy()
    ref16, inter16 = O.unet_forward(arch, sd, x, emulate='f16', return_intermediates=True)
    ref16 = ref16.numpy()
    with Engine(arch, blob) as e:
        e.set_precision('f16')
        lg, mk = e.forward(x, logits=True, mask=(W % 32 == 0))
        m16, r16 = stats(lg, ref16); m32, r32 = stats(lg, ref32)
        o16, q16 = stats(ref16, ref32)
        flips = None
        if mk is not None:
            flips = float((unpack_mask(mk, W) != O.logits_to_mask(ref16).numpy()).mean())
        print(f'[{name}] engine vs f16-oracle max {m16:.3e} rms {r16:.3e} | engine vs fp32-oracle max {m32:.3e} rms {r32:.3e} | '
              f'f16-oracle vs fp32-oracle max {o16:.3e} rms {q16:.3e} | mask bits differing vs f16-oracle {flips}', flush=True)
        if name != 'tiny_b37':
            worst_e2e = (max(worst_e2e[0], m16), max(worst_e2e[1], r16))
        if name in ('net5_128', 'canon_k18'):
            prog = {o['name']: o for o in arch.program()}
            for n, o in prog.items():
                if n.endswith('.up'):
                    continue
                if n == 'enc0.c0' and not e.materialised(n):
                    continue
                if n == 'head':
                    got, ins = lg, (o['src'],)
                else:
                    got = e.debug_tensor(n)
                    ins = (prog[n.replace('.c0', '.up')]['src'], o['skip']) if (n.startswith('dec') and n.endswith('.c0')) else (o['src'],)
                srcs = [x if i == 'input' else (e.debug_tensor(i) if e.materialised(i) else inter16[i].numpy()) for i in ins]
                want = O.layer_forward(arch, sd, n, *srcs, emulate='f16', storage_view=True).numpy()
                m, r = stats(got, want)
                kern = ''
                print(f'    layer {n:8s} {got.shape[2]:4d}x{got.shape[3]:<4d} max {m:.3e} rms {r:.3e}', flush=True)
                if got.shape[2] * got.shape[3] >= 64:
                    worst_layer = (max(worst_layer[0], m), max(worst_layer[1], r))
print(f'WORST end-to-end (without tiny_b37): max {worst_e2e[0]:.3e} rms {worst_e2e[1]:.3e};  WORST per layer (>= 64 px): max {worst_layer[0]:.3e} rms {worst_layer[1]:.3e}')

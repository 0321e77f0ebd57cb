"""Accuracy experiment: emulate a split-fp16 (hi+lo, 3 products) convolution on CPU and compare canonical-net logits
with the fp64-accumulating truth and the plain torch fp32 oracle."""
import sys, time, numpy as np, torch, torch.nn.functional as F
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import prng, weights
from oracle import torch_oracle as O

def split(t, dtype):
    hi = t.to(dtype).float()
    lo = (t - hi).to(dtype).float()
    return hi, lo

def pow2_scale(w, target=8192.0):
    m = float(w.abs().max())
    return 2.0 ** np.floor(np.log2(target / m)) if m > 0 else 1.0

def make_conv(dtype, nterms):
    def conv(x, w, b, stride=1, padding=0, transposed=False):
        s = pow2_scale(w) if dtype == torch.float16 else 1.0
        wh, wl = split(w * s, dtype)
        xh, xl = split(x, dtype)
        f = (lambda a, bb: F.conv_transpose2d(a, bb, None, stride=stride)) if transposed else (lambda a, bb: F.conv2d(a, bb, None, stride=stride, padding=padding))
        y = f(xh, wh)
        if nterms >= 3:
            y = y + f(xh, wl) + f(xl, wh)
        if nterms >= 4:
            y = y + f(xl, wl)
        return y / s + b.view(1, -1, 1, 1)
    return conv

def forward(arch, sd, x, conv):
    sd = {k: torch.from_numpy(v) for k, v in sd.items()}
    x = torch.from_numpy(x)
    skips = []
    def block(x, k, stride, exact=False):
        if exact:
            y = F.conv2d(x, sd[f'{k}.conv.weight'], sd[f'{k}.conv.bias'], stride=stride, padding=1)
        else:
            y = conv(x, sd[f'{k}.conv.weight'], sd[f'{k}.conv.bias'], stride=stride, padding=1)
        y = F.instance_norm(y, None, None, sd[f'{k}.norm.weight'], sd[f'{k}.norm.bias'], True, 0.1, arch.norm_eps)
        return F.leaky_relu(y, arch.leaky_slope)
    with torch.no_grad():
        for s in range(arch.n_stages):
            for i in range(2):
                x = block(x, f'encoder.stages.{s}.0.convs.{i}', 2 if (i == 0 and s > 0) else 1, exact=(s == 0 and i == 0))
            skips.append(x)
        for j in range(arch.n_stages - 1):
            lvl = arch.n_stages - 2 - j
            k = f'decoder.transpconvs.{j}'
            x = conv(x, sd[f'{k}.weight'], sd[f'{k}.bias'], stride=2, transposed=True)
            x = torch.cat((x, skips[lvl]), 1)
            for i in range(2):
                x = block(x, f'decoder.stages.{j}.convs.{i}', 1)
        k = f'decoder.seg_layers.{arch.n_stages - 2}'
        return F.conv2d(x, sd[f'{k}.weight'], sd[f'{k}.bias']).numpy()

a = UNetArch.canonical(); sd = weights.synthetic_state_dict(a, 1); blob = weights.pack_blob(a, sd)
x = prng.normal_f32(0, 0, (1, 2, 512, 512))
from oracle import c_oracle as C
truth = C.unet_forward(a, blob, x, acc64=True); yt = O.unet_forward(a, sd, x).numpy()
print('torch fp32 vs truth', np.abs(yt - truth).max(), flush=True)
for name, dt, nt in (('f16x3', torch.float16, 3), ('f16x4', torch.float16, 4), ('bf16x3', torch.bfloat16, 3), ('f16x1', torch.float16, 1)):
    t = time.time(); y = forward(a, sd, x, make_conv(dt, nt))
    d = y - truth
    print(f'{name}: vs truth max {np.abs(d).max():.3e} rms {np.sqrt((d**2).mean()):.3e} | vs torch max {np.abs(y - yt).max():.3e}  ({time.time()-t:.1f}s)', flush=True)

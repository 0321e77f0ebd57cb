"""CPU experiment: would Winograd F(2x2,3x3) for the stride-1 3x3 layers (2.25x fewer MFMAs on 83 % of the
FLOPs) stay inside the fp32 parity budget?  Emulation: fp64-pre-transformed weights U = G g G^T, fp32 input transform V = B^T d B
and output transform Y = A^T M A, the 16 per-position GEMMs as the engine would issue them (fp16 hi/lo split of U and V, three
products, fp32 accumulation).  Compared on the canonical net with the fp64-accumulating truth and the torch fp32 oracle, next to
the direct split convolution.  Also prints the single-product 16-bit formats (fp16 vs bf16 storage) for the config-3 note."""
import os, sys, time
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import prng, weights
from oracle import torch_oracle as O
from oracle import c_oracle as C

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def split(t, dtype=torch.float16):
    hi = t.to(dtype).float()
    return hi, (t - hi).to(dtype).float()


def pow2_scale(w, target=8192.0):
    m = float(w.abs().max())
    return 2.0 ** np.floor(np.log2(target / m)) if m > 0 else 1.0


def direct_split(x, w, b, stride=1, padding=1, transposed=False, dtype=torch.float16, nterms=3):
    s = pow2_scale(w) if dtype == torch.float16 else 1.0
    wh, wl = split(w * s, dtype); xh, xl = split(x, dtype)
    f = (lambda a, bb: F.conv_transpose2d(a, bb, None, stride=stride)) if transposed else (lambda a, bb: F.conv2d(a, bb, None, stride=stride, padding=padding))
    y = f(xh, wh)
    if nterms >= 3:
        y = y + f(xh, wl) + f(xl, wh)
    return y / s + b.view(1, -1, 1, 1)


def winograd_split(x, w, b):
    """3x3 stride-1 pad-1 conv as F(2x2,3x3); x [1,C,H,W] (H, W even), w [K,C,3,3]."""
    _, Cc, H, W = x.shape
    K = w.shape[0]
    U = torch.einsum('ij,kcjl,ml->imkc', G, w.double(), G)                  # [4,4,K,C] in fp64
    s = pow2_scale(U.float())
    Uh, Ul = split((U * s).float())
    d = F.unfold(x, kernel_size=4, stride=2, padding=1).view(Cc, 4, 4, -1)   # [C,4,4,T]
    V = torch.einsum('ij,cjlt,ml->imct', BT, d, BT)                          # fp32 input transform [4,4,C,T]
    Vh, Vl = split(V)
    M = torch.einsum('imkc,imct->imkt', Uh, Vh) + torch.einsum('imkc,imct->imkt', Uh, Vl) + torch.einsum('imkc,imct->imkt', Ul, Vh)
    Y = torch.einsum('pi,imkt,qm->kpqt', AT, M / s, AT)                      # [K,2,2,T]
    th, tw = H // 2, W // 2
    y = Y.view(K, 2, 2, th, tw).permute(0, 3, 1, 4, 2).reshape(1, K, H, W)
    return y + b.view(1, -1, 1, 1)


def forward(arch, sd, x, mode):
    sd = {k: torch.from_numpy(v) for k, v in sd.items()}
    x = torch.from_numpy(x)
    skips = []

    def conv(x, w, b, stride=1, transposed=False):
        if mode == 'winograd' and stride == 1 and not transposed and x.shape[-1] >= 4:
            return winograd_split(x, w, b)
        if mode in ('f16x1', 'bf16x1'):
            return direct_split(x, w, b, stride, 1, transposed, torch.float16 if mode == 'f16x1' else torch.bfloat16, 1)
        return direct_split(x, w, b, stride, 1, transposed)

    def block(x, k, stride, exact=False):
        y = F.conv2d(x, sd[f'{k}.conv.weight'], sd[f'{k}.conv.bias'], stride=stride, padding=1) if exact else \
            conv(x, sd[f'{k}.conv.weight'], sd[f'{k}.conv.bias'], stride)
        y = F.instance_norm(y, None, None, sd[f'{k}.norm.weight'], sd[f'{k}.norm.bias'], True, 0.1, arch.norm_eps)
        y = F.leaky_relu(y, arch.leaky_slope)
        return y.to(torch.float16).float() if mode == 'f16x1' else (y.to(torch.bfloat16).float() if mode == 'bf16x1' else y)
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


if __name__ == '__main__':
    torch.set_num_threads(os.cpu_count() or 1)
    a = UNetArch.canonical(); sd = weights.synthetic_state_dict(a, 1); blob = weights.pack_blob(a, sd)
    x = prng.normal_f32(0, 0, (1, 2, 512, 512))
    truth = C.unet_forward(a, blob, x, acc64=True); yt = O.unet_forward(a, sd, x).numpy()
    print(f'torch fp32 (ATen) vs fp64 truth: max {np.abs(yt - truth).max():.3e}', flush=True)
    for mode in ('direct f16x3', 'winograd', 'f16x1', 'bf16x1'):
        t = time.time(); y = forward(a, sd, x, mode.split()[-1] if mode != 'direct f16x3' else 'direct')
        d = y - truth
        print(f'{mode:14s}: vs truth max {np.abs(d).max():.3e} rms {np.sqrt((d ** 2).mean()):.3e} | vs torch max {np.abs(y - yt).max():.3e} '
              f'rms {np.sqrt(((y - yt) ** 2).mean()):.3e}  ({time.time() - t:.0f} s)', flush=True)

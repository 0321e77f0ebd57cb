"""First GPU contact: parity of the HIP engine vs the oracles on small nets and the canonical net, and rough timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from totalsegmentator2d_amd.arch import UNetArch
from totalsegmentator2d_amd import weights, prng
from totalsegmentator2d_amd.engine import Engine, unpack_mask
from oracle import torch_oracle as O, c_oracle as C

def small(n_stages=4, feats=(32, 64, 64, 96), K=5, C_in=2):
    return UNetArch(input_channels=C_in, num_classes=K, n_stages=n_stages, features_per_stage=feats,
                    kernel_sizes=((3, 3),) * n_stages, strides=((1, 1),) + ((2, 2),) * (n_stages - 1),
                    n_conv_per_stage=(2,) * n_stages, n_conv_per_stage_decoder=(2,) * (n_stages - 1))

def check(arch, B, H, W, seed=1, tag=''):
    sd = weights.synthetic_state_dict(arch, seed)
    blob = weights.pack_blob(arch, sd)
    x = prng.normal_f32(0, 7, (B, arch.input_channels, H, W))
    with Engine(arch, blob) as e:
        lg, mk = e.forward(x, logits=True, mask=(W % 32 == 0))
    yt = O.unet_forward(arch, sd, x).numpy()
    err = np.abs(lg - yt).max()
    ok_mask = None
    if mk is not None:
        ref = O.logits_to_mask(lg).numpy()
        ok_mask = bool((unpack_mask(mk, W) == ref).all())
    print(f"[{tag}] B={B} {H}x{W} max|gpu-torch|={err:.3e} max|logit|={np.abs(yt).max():.2f} mask_exact={ok_mask}", flush=True)
    return err

check(small(), 2, 64, 96, tag='small4')
check(small(), 3, 16, 16, tag='small4-tiny')
check(small(), 37, 8, 16, tag='small4-tiny-b37')
check(small(3, (32, 32, 64), 3, 1), 1, 32, 64, tag='small3-1ch')
check(small(5, (32, 64, 128, 256, 512), 18), 2, 128, 128, tag='mid5')
a = UNetArch.canonical()
t = time.time(); blob = np.load('/tmp/blob_canon_s1.npy') if os.path.exists('/tmp/blob_canon_s1.npy') else weights.pack_blob(a, weights.synthetic_state_dict(a, 1)); print('weights', time.time() - t, flush=True)
sd = weights.unpack_blob(a, blob)
x = prng.normal_f32(0, 0, (2, 2, 512, 512))
with Engine(a, blob) as e:
    lg, mk = e.forward(x, logits=True, mask=True)
    yt = O.unet_forward(a, sd, x).numpy()
    print('canonical B=2 max|gpu-torch| =', np.abs(lg - yt).max(), 'mask exact', bool((unpack_mask(mk, 512) == O.logits_to_mask(lg).numpy()).all()),
          'mask vs torch-e2e mismatch', int((unpack_mask(mk, 512) != O.logits_to_mask(yt).numpy()).sum()), flush=True)
    import torch
    for B in (8, 64):
        xd = torch.randn(B, 2, 512, 512, device='cuda')
        e.forward(xd); torch.cuda.synchronize()
        t = time.time(); n = 3
        for _ in range(n): e.forward(xd)
        torch.cuda.synchronize(); dt = (time.time() - t) / n
        print(f'B={B}: {dt*1e3:.1f} ms/forward = {B/dt:.1f} slices/s = {B/dt*119.6/1e3:.1f} TFLOP/s', flush=True)
    e.set_profiling(True); e.forward(xd); torch.cuda.synchronize()
    for k, v in e.op_times().items(): print(f'  {k:14s} {v:8.3f} ms')

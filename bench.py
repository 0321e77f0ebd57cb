#!/usr/bin/env python3
"""bench.py - BASELINE.json metric: 2-ch 512x512 slices/sec on N MI355X (config 2: synthetic batch=64, one ts2d-v2
sub-model (K=18), fp32).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one forward pass of the hot path (the whole PlainConvUNet, input NCHW -> fp32 logits NCHW + packed masks)
over one batch of 64 synthetic slices per GPU, inputs already resident in HBM.  Slices shard over ranks with no
data-path collective (weak scaling); weights are broadcast once from rank 0 with RCCL before the timed region.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_F16_MFMA_TFLOPS = 2500.0     # dense fp16/bf16 MFMA peak (same guide); the split path issues 3 fp16 MFMA products per MAC
MEASURED_F16_PIPE_TFLOPS = 32768 * 1024 / 20.8e-9 / 1e12      # 32x32x16 MFMA = 32768 FLOP, 1024 SIMDs, 20.8 ns each (probe): ~1613
PEAK_HBM_GBS = 8000.0


def host_cores() -> int:
    """Threads for the CPU baseline: the process's CPU share (affinity mask, cgroup quota), capped at 16 - the
    per-GPU host share of the pool's boxes (a 256-thread pool on a 16-core share ran 30x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get('TS2D_CPU_THREADS', 16))))


def cpu_baseline(arch, sd, budget_s: float = 20.0, check=None):
    """The oracle (torch-CPU restatement = the ATen kernels the reference CPU path runs), B = 1 per call like the
    reference, all host cores, on a bounded sample of the same workload.  `check` = (x, gpu logits)
    of one slice: the oracle's output on it gives the live logit max-abs-err of the metric."""
    import torch
    from oracle import torch_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    x = torch.randn(1, arch.input_channels, 512, 512)
    err = None
    if check is not None:
        ref = O.unet_forward(arch, sd, check[0]).numpy()         # doubles as the warm-up
        err = float(np.abs(ref - check[1]).max())
    else:
        O.unet_forward(arch, sd, x)                  # warm-up
    t0 = time.time()
    n = 0
    while n < 3 or (time.time() - t0 < budget_s and n < 64):
        O.unet_forward(arch, sd, x)
        n += 1
    dt = time.time() - t0
    return {'value': round(n / dt, 3), 'unit': 'slices/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} single-slice (B=1, no mirroring) 2x512x512 forwards of the same network, torch-CPU oracle, '
                      f'{torch.get_num_threads()} threads'}, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=64, help='slices per GPU per step (config 2: 64)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-profile', action='store_true', help='do not bracket kernels with HIP events')
    ap.add_argument('--precision', choices=('split', 'exact', 'f16'), default='split',
                    help="split: fp16 hi/lo x3 MFMA with fp32 accumulation (fp32-equivalent accuracy, default); exact: fp32 MFMA; "
                         "f16: fp16 storage + one fp16 MFMA product (BASELINE configs 3/5, outside the fp32 parity tolerance)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from totalsegmentator2d_amd import parallel, weights
    from totalsegmentator2d_amd.arch import UNetArch, OP_CONV3X3
    from totalsegmentator2d_amd.engine import Engine

    rank, local_rank, world = parallel.env_rank_world()
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    multi = world > 1 or os.environ.get('TS2D_FORCE_DIST') == '1'      # the flag rehearses the RCCL path on one GPU
    if multi:
        parallel.init_process_group('nccl')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    arch = UNetArch.canonical(input_channels=2, num_classes=18)            # one ts2d-v2 sub-model (cardiac, K = 18)
    B, H, W = args.batch, 512, 512
    sd = None
    if rank == 0:
        sd = weights.synthetic_state_dict(arch, seed=1)                    # He-normal, BASELINE.md section 4
        engine = Engine(arch, weights.pack_blob(arch, sd), device=local_rank)
    else:
        engine = Engine(arch, None, device=local_rank)
    if os.environ.get('TS2D_FORCE_DIST') == '1' and world == 1:
        # rehearsal: a second, empty replica on the same GPU receives the arena by device copy after the (self) broadcast
        replica = Engine(arch, None, device=local_rank)
    bcast_ms = None
    if multi:
        dist.barrier()
        t0 = time.time()
        parallel.broadcast_engine_weights(engine, src=0)                   # one RCCL broadcast over xGMI
        bcast_ms = (time.time() - t0) * 1e3
        if os.environ.get('TS2D_FORCE_DIST') == '1' and world == 1:
            ps, ns = engine.weight_buffer(); pd, nd = replica.weight_buffer()
            torch.as_tensor(parallel._DevicePtrTensor(pd, nd), device=dev).copy_(torch.as_tensor(parallel._DevicePtrTensor(ps, ns), device=dev))
            torch.cuda.synchronize(dev); replica.weights_ready()
            xa = torch.randn(2, 2, 512, 512, device=dev)
            la, _ = engine.forward(xa); lb, _ = replica.forward(xa); torch.cuda.synchronize(dev)
            assert torch.equal(la, lb), 'replica filled through the broadcast hook differs'
            replica.close()
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    x = torch.randn(B, 2, H, W, device=dev, generator=gen)                 # synthetic N(0,1) = post-z-score statistics
    logits = torch.empty(B, arch.num_classes, H, W, device=dev)
    mask = torch.empty(B, arch.num_classes, H, W // 32, dtype=torch.int32, device=dev)
    engine.reserve(B, H, W)
    engine.set_precision(args.precision)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        engine.forward(x, logits=True, mask=True, out_logits=logits, out_mask=mask, stream=stream)

    for _ in range(args.warmup):
        step()
    profile = not args.no_profile and rank == 0
    if profile:
        engine.set_profiling(True)
    op_ms = {}
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        if profile:                                                        # HIP events on the launch stream, read per step
            for k, v in engine.op_times().items():
                op_ms[k] = op_ms.get(k, 0.0) + v
    torch.cuda.synchronize(dev)
    if multi:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0)

    if rank == 0:
        work = arch.work(H, W)
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        out = {
            'metric': '2-ch 512x512 slices/sec', 'value': round(value, 2), 'unit': 'slices/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: synthetic batch=64 2x512x512, one ts2d-v2 sub-model '
                                   '(canonical 8-stage PlainConvUNet, K=18), fp32, logits + packed masks out',
                       'batch_per_gpu': B, 'global_batch': world * B, 'H': H, 'W': W, 'K': arch.num_classes,
                       'parallelism': f'slice-dp{world}', 'gflop_per_slice': round(work['flops'] / 1e9, 2),
                       'act_mb_per_slice': round(work['act_bytes'] / 1e6, 1)},
            'tflops': round(value * work['flops'] / 1e12, 2),
        }
        if bcast_ms is not None:
            out['weight_broadcast_ms'] = round(bcast_ms, 2)
        split = args.precision == 'split'
        out['dtype'] = {'split': 'f32 storage/accumulate, products as 3x fp16-split MFMA (f16x3)', 'exact': 'f32',
                        'f16': 'f16 storage + f16 MFMA, f32 accumulate/statistics (NOT within the fp32 parity tolerance)'}[args.precision]
        out['precision_mode'] = args.precision
        if profile and op_ms:
            # dominant kernel = the stride-1 3x3 implicit-GEMM conv (conv3x3_f16x3_one + conv3x3_f16x3 on the 8x8/4x4 levels /
            # conv3x3_h32 / conv_mfma_f32<9,1,..>): 22 launches/step
            prog = arch.program()
            per = {o['name']: 2.0 * m['macs'] for o, m in zip(prog, work['per_layer'])
                   if o['op'] == OP_CONV3X3 and o['stride'] == 1 and o['src'] != 'input'}
            conv_ms = sum(v for k, v in op_ms.items() if k in per) / args.steps
            conv_flops = sum(per.values()) * B
            achieved = conv_flops / (conv_ms * 1e-3) / 1e12
            peak = {'split': PEAK_F16_MFMA_TFLOPS / 3.0, 'exact': PEAK_FP32_MFMA_TFLOPS, 'f16': PEAK_F16_MFMA_TFLOPS}[args.precision]
            traffic = None
            pmc = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(f'conv3x3_s1_{args.precision}_hbm_bytes_per_launch_avg')
                except Exception:
                    traffic = None
            all3 = {o['name'] for o in prog if o['op'] == OP_CONV3X3}
            out['roofline'] = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                               'frac': round(achieved / peak, 4), 'traffic': traffic,
                               'kernel': {'split': 'conv3x3_f16x3_one (18) + conv3x3_f16x3 (4)', 'f16': 'conv3x3_h32 (18) + conv3x3_f16x3<f16> (4)',
                                          'exact': 'conv_mfma_f32<9,1,16,*>'}[args.precision] + f' ({len(per)} launches/step)',
                               'peak_note': ('dense fp16 MFMA 2500 TFLOP/s / 3 products per MAC' if split else 'fp32 MFMA 32x32x2'),
                               # profiles/r01_mfma_coissue_probe.txt: a bare v_mfma_f32_32x32x16_f16 stream on every SIMD runs at
                               # 20.5-21 ns per MFMA (= 32 cycles at the ~1.52 GHz this chip sustains under dense matrix load)
                               'frac_of_measured_matrix_pipe_rate': (round(achieved / (MEASURED_F16_PIPE_TFLOPS / (3.0 if split else 1.0)), 4)
                                                                     if args.precision in ('split', 'f16') else None),
                               'algorithmic_flop_per_launch_avg': round(conv_flops / len(per)),
                               'kernel_ms_per_launch_avg': round(conv_ms / len(per), 4),
                               'kernel_ms_per_step': round(conv_ms, 3), 'kernel_share_of_step': round(conv_ms / ms_per_step, 4),
                               'all_conv3x3_ms_per_step': round(sum(v for k, v in op_ms.items() if k in all3) / args.steps, 3),
                               'whole_step_hbm_frac_layerwise': round(value / world * work['act_bytes'] / 1e9 / PEAK_HBM_GBS, 4),
                               'whole_step_fp32_mfma_equiv_frac': round(value / world * work['flops'] / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)}
            top = sorted(op_ms.items(), key=lambda kv: -kv[1])[:8]
            out['top_ops_ms'] = {k: round(v / args.steps, 3) for k, v in top}
        if world == 1:
            # the other arithmetic mode on the same workload (outside the timed region above; same step definition)
            other = 'exact' if split else 'split'
            out['f16_mode'] = None
            engine.set_profiling(False)
            engine.set_precision(other)
            step(); torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(max(2, args.steps // 2)):
                step()
            torch.cuda.synchronize(dev)
            out['other_mode'] = {'precision_mode': other, 'value': round(B * max(2, args.steps // 2) / (time.perf_counter() - t1), 2), 'unit': 'slices/s'}
            if split:                                                   # third mode, for the record (configs 3/5 arithmetic)
                engine.set_precision('f16')
                step(); torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(max(2, args.steps // 2)):
                    step()
                torch.cuda.synchronize(dev)
                out['f16_mode'] = {'precision_mode': 'f16', 'value': round(B * max(2, args.steps // 2) / (time.perf_counter() - t1), 2), 'unit': 'slices/s',
                                   'note': 'fp16 storage + single fp16 MFMA product; logit rms error ~8e-3, outside the 1e-4 parity tolerance'}
            engine.set_precision(args.precision)
        if world == 1 and not args.no_cpu_baseline:
            step(); torch.cuda.synchronize(dev)
            chk = (x[:1].cpu().numpy(), logits[:1].cpu().numpy())
            out['cpu_baseline'], err = cpu_baseline(arch, sd, check=chk)
            out['logit_max_abs_err_vs_oracle'] = {'value': err, 'tol': 1e-4, 'slices_checked': 1,
                                                  'note': 'live check in the cpu_baseline leg; full parity: pytest -m gpu'}
        print(json.dumps(out), flush=True)
    engine.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

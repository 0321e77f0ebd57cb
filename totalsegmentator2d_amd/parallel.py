"""Slice-batch data parallelism across the GPUs of one node.

The reference has no distributed code at all (its only parallelism is one worker process per sub-model, reference
``ts2d/core/inference/predictor.py:79-86``).  The path shards embarrassingly over slices - InstanceNorm statistics
are per sample - so the design is: one process per GPU (``torch.distributed``, backend ``nccl`` = RCCL on ROCm,
``gloo`` in CPU tests), ONE broadcast of the packed weight arena from rank 0 over xGMI at load time, then per-rank
independent inference on a contiguous block of slices.  No all-reduce; the only other traffic is a gather of scalars.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`: sizes differ by at most one, earlier ranks take the remainder."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_rank_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init_process_group(backend: Optional[str] = None):
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    rank, local_rank, world = env_rank_world()
    if backend == 'nccl':
        # RCCL prints a version banner on STDOUT when the communicator is created; bench.py's contract is ONE JSON line on
        # stdout, so fd 1 points at stderr while RCCL initialises (init + first collective), then is restored.
        import sys
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)


class _DevicePtrTensor:
    """Wraps raw device memory for torch (``__cuda_array_interface__``) so RCCL can broadcast INTO the engine's weight
    arena without a staging copy."""
    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {'shape': (nbytes // 4,), 'typestr': '<f4', 'data': (ptr, False), 'version': 3,
                                         'strides': None}


def broadcast_engine_weights(engine, src: int = 0):
    """RCCL broadcast of the packed weight arena (185 MB fp32 for the canonical net) from rank `src`; afterwards every
    replica is marked ready (C-ABI: ts2d_engine_weight_buffer / ts2d_engine_weights_ready)."""
    import torch
    import torch.distributed as dist
    ptr, nbytes = engine.weight_buffer()
    t = torch.as_tensor(_DevicePtrTensor(ptr, nbytes), device=torch.device('cuda', engine.device))
    dist.broadcast(t, src=src)
    torch.cuda.synchronize(engine.device)
    engine.weights_ready()
    return nbytes


def broadcast_blob(blob: Optional[np.ndarray], n_floats: int, src: int = 0) -> np.ndarray:
    """Host-side variant (gloo / CPU tests): broadcast the PyTorch-layout blob itself."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.float32)) if dist.get_rank() == src else torch.empty(n_floats, dtype=torch.float32)
    dist.broadcast(t, src=src)
    return t.numpy()


def max_over_ranks(value: float) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ----------------------------------------------------------------------------- BASELINE config 4: sharded slice stream
STREAM_ID = 1000          # prng stream of the synthetic slices (tests/cases.py::make_input uses the same stream)


def synth_slices(device: int, seed: int, first_slice: int, n_slices: int, shape_chw, out=None, stream: int = 0):
    """Slices [first_slice, first_slice + n_slices) of the synthetic stream `seed`, generated ON THE DEVICE (C-ABI
    ts2d_synth_slices) into a torch CUDA tensor [n_slices, C, H, W]; bit-identical to
    ``prng.normal_f32(seed, STREAM_ID, (n_slices, C, H, W), offset=first_slice * C * H * W)`` on the host."""
    import ctypes
    import torch
    from . import _lib, prng
    C, H, W = (int(v) for v in shape_chw)
    per = C * H * W
    if out is None:
        out = torch.empty((n_slices, C, H, W), dtype=torch.float32, device=torch.device('cuda', device))
    lib = _lib.load()
    _lib.check(lib.ts2d_synth_slices(int(device), prng.key(seed, STREAM_ID), first_slice * per, n_slices * per,
                                     out.data_ptr(), ctypes.c_void_p(stream)), 'ts2d_synth_slices')
    return out


def run_slice_stream(engine, seed: int, total_slices: int, rank: int, world: int, shape_chw=(2, 512, 512), batch: int = 64,
                     keep_masks: bool = True):
    """Config 4 on one rank: the contiguous block ``shard_range(total_slices, rank, world)`` of the stream, generated on the
    device BEFORE the timed region (inputs resident in HBM), pushed through the engine in batches of `batch`, packed masks kept
    for the whole block.  Returns (lo, hi, masks int32 [hi-lo, K, H, W/32] or None, seconds of the timed region).  No
    data-path collective: the caller reduces the time with max_over_ranks."""
    import time
    import torch
    lo, hi = shard_range(total_slices, rank, world)
    n = hi - lo
    C, H, W = shape_chw
    K = engine.arch.num_classes
    dev = torch.device('cuda', engine.device)
    x = synth_slices(engine.device, seed, lo, n, shape_chw) if n else torch.empty((0, C, H, W), device=dev)
    masks = torch.empty((n, K, H, W // 32), dtype=torch.int32, device=dev) if keep_masks else None
    scratch = None if keep_masks else torch.empty((min(batch, max(n, 1)), K, H, W // 32), dtype=torch.int32, device=dev)
    engine.reserve(min(batch, max(n, 1)), H, W)
    stream = torch.cuda.current_stream(dev).cuda_stream
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for a in range(0, n, batch):
        b = min(a + batch, n)
        engine.forward(x[a:b], logits=False, mask=True, out_mask=(masks[a:b] if keep_masks else scratch[:b - a]), stream=stream)
    torch.cuda.synchronize(dev)
    return lo, hi, masks, time.perf_counter() - t0

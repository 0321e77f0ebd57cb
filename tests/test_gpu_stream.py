"""BASELINE config 4 (slice-batch sharding of a synthetic stream): the device-side generator and the rank-block runner."""
import numpy as np
import pytest

from tests import cases
from tests.conftest import blob_for
from totalsegmentator2d_amd import parallel, prng
from totalsegmentator2d_amd.engine import Engine

pytestmark = pytest.mark.gpu


def test_device_generator_is_bit_identical_to_the_host_prng():
    import torch
    shape = (2, 64, 96)
    per = int(np.prod(shape))
    for seed, first, n in ((0, 0, 3), (7, 5, 2), (0, 9999, 1)):
        dev = parallel.synth_slices(0, seed, first, n, shape)
        torch.cuda.synchronize()
        host = prng.normal_f32(seed, parallel.STREAM_ID, (n,) + shape, offset=first * per)
        assert np.array_equal(dev.cpu().numpy(), host)
    # a value depends on (seed, slice index) only: overlapping blocks agree
    a = parallel.synth_slices(0, 3, 10, 4, shape).cpu().numpy()
    b = parallel.synth_slices(0, 3, 12, 4, shape).cpu().numpy()
    assert np.array_equal(a[2:], b[:2]) and abs(float(a.mean())) < 0.02 and abs(float(a.std()) - 1.0) < 0.02


def test_rank_blocks_equal_the_single_rank_result():
    """Sharding is by contiguous blocks with no data-path collective: the masks each rank of a world of 3 produces for its
    block are bit-identical to the single-rank run over the same slice indices (per-sample statistics, no cross-slice state)."""
    import torch
    arch = cases.unet(4, (32, 64, 64, 128), 6)
    _, blob = blob_for(arch, 51)
    shape, total = (2, 64, 96), 23
    # ("sbk" = 0: the kernels of full batches at any batch size - with the small-batch dispatch of round 6 a slice is bit-identical
    #  only between batches that take the same path; its cross-regime bound is tests/test_gpu_small_batch.py's)
    with Engine(arch, blob, options={'sbk': 0}) as e:
        lo, hi, whole, _ = parallel.run_slice_stream(e, 11, total, 0, 1, shape, batch=8)
        assert (lo, hi) == (0, total) and whole.shape == (total, 6, 64, 3)
        got = []
        for r in range(3):
            lo, hi, m, dt = parallel.run_slice_stream(e, 11, total, r, 3, shape, batch=5)
            assert dt > 0 and torch.equal(m, whole[lo:hi])
            got.append((lo, hi))
        assert got == [parallel.shard_range(total, r, 3) for r in range(3)] and got[-1][1] == total
        # the masks are the predicate of the engine's own logits on the generated slices
        x = parallel.synth_slices(0, 11, 4, 2, shape)
        lg, mk = e.forward(x, logits=True, mask=True)
        torch.cuda.synchronize()
        assert torch.equal(mk, whole[4:6])
        assert (lg > 1.5 * 2.0 ** -24).any() and (lg <= 1.5 * 2.0 ** -24).any()
    with Engine(arch, blob) as e:                  # product default: blocks of another batch size may differ in bits that sit at the threshold only
        _, _, whole1, _ = parallel.run_slice_stream(e, 11, total, 0, 1, shape, batch=8)
        for r in range(3):
            lo, hi, m, _ = parallel.run_slice_stream(e, 11, total, r, 3, shape, batch=5)
            diff = (m ^ whole1[lo:hi]).to(torch.int64) & 0xFFFFFFFF
            nbits = sum(int(((diff >> b) & 1).sum()) for b in range(32))
            assert nbits <= max(4, m.numel() * 32 // 20000), nbits
        assert int(((whole1 ^ whole).to(torch.int64) & 0xFFFFFFFF != 0).sum()) <= max(4, whole.numel() // 20000)


def test_bench_config4_mode_with_the_rccl_path_rehearsed_on_one_gpu():
    """`bench.py --workload config4` under TS2D_FORCE_DIST=1: the process group (RCCL, world of 1), the weight broadcast into the
    engine arena and the sharded stream run end to end; ONE JSON line with the config-4 workload and the broadcast time."""
    import json, os, subprocess, sys
    from tests.conftest import ROOT
    env = dict(os.environ, TS2D_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'config4', '--stream', '192', '--batch', '32'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['scaling'] == 'strong' and 'configs[3]' in out['config']['workload'] and out['config']['stream_slices'] == 192
    assert out['value'] > 0 and out['weight_broadcast_ms'] is not None and out['n_gpus'] == 1

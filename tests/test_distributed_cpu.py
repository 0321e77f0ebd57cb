"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding + weight-broadcast logic of parallel.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import ROOT
from totalsegmentator2d_amd import parallel

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["TS2D_ROOT"])
import torch.distributed as dist
from tests import cases
from totalsegmentator2d_amd import parallel, weights
parallel.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
arch = cases.unet(2, (32, 32), 2)
blob = weights.pack_blob(arch, weights.synthetic_state_dict(arch, 9)) if rank == 0 else None
got = parallel.broadcast_blob(blob, arch.n_params(), src=0)                     # weight broadcast, root 0
ref = weights.pack_blob(arch, weights.synthetic_state_dict(arch, 9))
assert np.array_equal(got, ref), "broadcast blob differs"
lo, hi = parallel.shard_range(10, rank, world)                                   # contiguous slice blocks
mx = parallel.max_over_ranks(float(rank + 1))                                    # timing reduction used by bench.py
assert mx == float(world)
open(os.path.join(os.environ["TS2D_OUT"], f"rank{rank}.txt"), "w").write(f"{lo} {hi}")
dist.barrier(); dist.destroy_process_group()
'''


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 10000):
        for world in (1, 2, 3, 4, 8):
            blocks = [parallel.shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_range(4, 2, 2)


def test_two_rank_gloo_broadcast_and_sharding(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    env = dict(os.environ, TS2D_ROOT=ROOT, TS2D_OUT=str(tmp_path), MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                        '--master-addr', '127.0.0.1', '--master-port', '29533', str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (tmp_path / 'rank0.txt').read_text() == '0 5' and (tmp_path / 'rank1.txt').read_text() == '5 10'


@pytest.mark.parametrize('n', [2, 3])
def test_bench_self_launches_n_ranks(n):
    """`python bench.py --gpus N` with no launcher around it (the driver's command): the script starts torch.distributed.run
    itself as a child, rank 0 prints ONE JSON line, the exit code is relayed.  TS2D_BENCH_DRYRUN=gloo keeps it on the CPU: process
    group, weight-blob broadcast, config-4 slice blocks (remainder ranks hold one slice more), max-over-ranks time."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(TS2D_BENCH_DRYRUN='gloo', OMP_NUM_THREADS='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '3', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == n and out['steps'] == 3 and out['warmup'] == 1 and out['dry_run'] == 'gloo'
    blocks = [hi - lo for lo, hi in (parallel.shard_range(10000, k, n) for k in range(n))]
    assert out['stream_blocks'] == blocks and out['stream_steps'] == (max(blocks) + 63) // 64
    assert out['elapsed_max_s'] >= 0.01 * n                      # the slowest rank's time is the one reported
    # a failing child's exit code comes back (unknown flag -> argparse exits 2 in every rank)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--no-such-flag'],
                         env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0

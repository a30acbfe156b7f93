"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- batch split by rank, barrier, max-over-ranks timing --
with the CPU oracle standing in for the GPU step.  The union of the two ranks' outputs must equal a single-process
run bit for bit (instances are independent: no collective touches sample data)."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from oalsfxpp_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TOTAL, FRAMES, BUFFERS = 7, 256, 3


def run_instances(first, last):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from harness import OracleApi, preset_effect
    from oalsfxpp_amd import desc
    from oracle import oracle as orc
    out = []
    for g in range(first, last):
        api = OracleApi(desc.FMT_STEREO, 48000, 1)
        api.set_effect(0, preset_effect(g * 13 % 113))
        api.apply_changes()
        out.append(np.concatenate([api.mix(orc.synth(g, k, FRAMES * 2).reshape(FRAMES, 2)).reshape(-1) for k in range(BUFFERS)]))
    return np.stack(out) if out else np.zeros((0, BUFFERS * FRAMES * 2), dtype=np.float32)


def worker(rank, world, port, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    first, last = sharding.shard_range(N_TOTAL, rank, world)
    sharding.barrier()
    out = run_instances(first, last)
    elapsed = sharding.max_over_ranks(1.0 + rank)          # a stand-in duration: the slower rank must win
    frames = sharding.sum_over_ranks(float((last - first) * FRAMES * BUFFERS))
    np.save(os.path.join(result_dir, f"out{rank}.npy"), out)
    if rank == 0:
        np.save(os.path.join(result_dir, "agg.npy"), np.array([elapsed, frames]))
    sharding.barrier()
    dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_ranges_partition_the_batch():
    for n in (1, 7, 4096, 262144):
        for world in (1, 2, 3, 8):
            ranges = [sharding.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_reproduce_the_single_process_result(tmp_path):
    world = 2
    mp.spawn(worker, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"out{r}.npy") for r in range(world)])
    want = run_instances(0, N_TOTAL)
    assert got.shape == want.shape and got.tobytes() == want.tobytes()
    elapsed, frames = np.load(tmp_path / "agg.npy")
    assert elapsed == 2.0 and frames == N_TOTAL * FRAMES * BUFFERS

"""world_size-2 gloo test of the N>1 path: signals are sharded over ranks, each rank produces its
shard's result block (here: the oracle stands in for the device as the block producer -- the test
exercises the partition + single gather, not the kernels), rank 0 gathers with one collective and
must hold exactly the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _signals(nsig, n=6000, sr=8000.0):
    t = np.arange(n) / sr
    return np.stack([0.3 * np.sin(2 * np.pi * (200.0 + 37.0 * b) * t) +
                     0.1 * np.sin(2 * np.pi * (900.0 + 11.0 * b) * t) for b in range(nsig)])


def _worker(rank, world, port, nsig, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pvoracle
        from pypevoc_amd.batch import gather_results, shard_range
        x = _signals(nsig)
        a, b = shard_range(nsig, rank, world)
        blocks = []
        for i in range(a, b):
            o = pvoracle.analyze(x[i], 8000.0, 512, 128, 4)
            blocks.append(np.stack([o[k] for k in ("f", "mag", "ph", "realph", "binno")]))
        F = pvoracle.nframes(x.shape[1], 512, 128)
        local = torch.from_numpy(np.stack(blocks)) if blocks else torch.zeros((0, 5, F, 4), dtype=torch.float64)
        full = gather_results(local, nsig, dst=0)
        if rank == 0:
            np.save(os.path.join(outdir, "gathered.npy"), full.numpy())
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nsig", [5, 4])
def test_sharded_analysis_gather_world2(tmp_path, nsig):
    from oracle import pvoracle
    pvoracle.build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, nsig, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npy"))
    x = _signals(nsig)
    exp = []
    for i in range(nsig):
        o = pvoracle.analyze(x[i], 8000.0, 512, 128, 4)
        exp.append(np.stack([o[k] for k in ("f", "mag", "ph", "realph", "binno")]))
    exp = np.stack(exp)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp)


def _pipe_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pypevoc_amd.batch import PipelinedGather
        n, steps = 1000, 7
        pg = PipelinedGather(n, torch.float64, torch.device("cpu"), dst=0)
        seen = []
        for i in range(steps):
            buf = pg.buffer(i)
            if rank == 0 and i >= 2:
                # the gather of step i-2 has completed before its buffers are reused
                seen.append(torch.stack(pg.result(i - 2)).clone())
            buf.copy_(torch.arange(n, dtype=torch.float64) + 1000.0 * i + 100000.0 * rank)
            pg.submit(i)
        pg.drain()
        if rank == 0:
            for i in (steps - 2, steps - 1):
                seen.append(torch.stack(pg.result(i)).clone())
            np.save(os.path.join(outdir, "pipe.npy"), torch.stack(seen).numpy())
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_world2(tmp_path):
    """The double-buffered asynchronous gather used by bench.py --gpus N: every step's blocks of
    every rank arrive intact although buffers are reused every second step."""
    port = _free_port()
    mp.spawn(_pipe_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "pipe.npy"))       # [steps, world, n]
    assert got.shape == (7, 2, 1000)
    for i in range(7):
        for r in range(2):
            assert np.array_equal(got[i, r], np.arange(1000.0) + 1000.0 * i + 100000.0 * r), (i, r)


def _consume_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pypevoc_amd.batch import PipelinedGather
        n, steps = 257, 9
        seen = {}

        def consume(step, blocks):
            assert step not in seen
            seen[step] = torch.stack(blocks).clone()

        pg = PipelinedGather(n, torch.uint8, torch.device("cpu"), dst=0, consume=consume)
        for i in range(steps):
            buf = pg.buffer(i)
            buf.copy_(((torch.arange(n) + 7 * i + 31 * rank) % 251).to(torch.uint8))
            pg.submit(i)
        pg.drain()
        if rank == 0:
            assert sorted(seen) == list(range(steps))
            np.save(os.path.join(outdir, "consume.npy"), torch.stack([seen[i] for i in range(steps)]).numpy())
        else:
            assert not seen
    finally:
        dist.destroy_process_group()


def test_pipelined_gather_consume_hook_world2(tmp_path):
    """bench.py's rank-0 unpack hook: `consume(step, blocks)` runs exactly once per step, in step order,
    with every rank's block intact, before the receive buffers are reused."""
    port = _free_port()
    mp.spawn(_consume_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "consume.npy"))    # [steps, world, n]
    assert got.shape == (9, 2, 257)
    for i in range(9):
        for r in range(2):
            assert np.array_equal(got[i, r], ((np.arange(257) + 7 * i + 31 * r) % 251).astype(np.uint8)), (i, r)


# ------------------------------------------------------------------ one long signal, sharded by frame ranges
def _frame_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pvoracle
        from pypevoc_amd.batch import frame_shard, gather_results
        x = _signals(1, n=20000)[0] + 0.01 * np.random.default_rng(3).standard_normal(20000)
        nfft, hop, K = 512, 128, 4
        F = pvoracle.nframes(len(x), nfft, hop)
        f0, f1, a, b, drop = frame_shard(len(x), nfft, hop, rank, world)
        o = pvoracle.analyze(x[a:b], 8000.0, nfft, hop, K)       # the oracle stands in for the device
        rows = np.stack([o[k][drop:] for k in ("f", "mag", "ph", "realph", "binno")], axis=1)   # [n, 5, K]
        assert rows.shape[0] == f1 - f0
        full = gather_results(torch.from_numpy(np.ascontiguousarray(rows)), F, dst=0)
        if rank == 0:
            np.save(os.path.join(outdir, "frames.npy"), full.numpy())
    finally:
        dist.destroy_process_group()


def test_frame_range_sharding_of_one_signal_world2(tmp_path):
    """a single long signal shards by frame ranges with a one-frame halo and no exchange step.
    Each rank analyses its sample range (the oracle stands in for the device), drops the halo row, one
    gather concatenates the rows: equal to the unsharded analysis (phase differences across the seam too)."""
    from oracle import pvoracle
    pvoracle.build()
    port = _free_port()
    mp.spawn(_frame_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "frames.npy"))     # [F, 5, K]
    x = _signals(1, n=20000)[0] + 0.01 * np.random.default_rng(3).standard_normal(20000)
    o = pvoracle.analyze(x, 8000.0, 512, 128, 4)
    exp = np.stack([o[k] for k in ("f", "mag", "ph", "realph", "binno")], axis=1)
    assert got.shape == exp.shape
    assert np.array_equal(got, exp)

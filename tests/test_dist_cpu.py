"""The N>1 path on CPU: world_size-2 gloo ranks shard one stream's per-frame series, reduce
sum_i c_i x_i with GLOBAL indices, and one scalar all-reduce reproduces the reference's pooled value
(mean of the pandas EWM, complexity_metrics.py:114-125,:302-309).  Also the one-stream-per-rank
summary reduction bench.py performs after its timed region."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rtvqa_amd import pooling


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, series, alpha, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T = len(series)
    lo, hi = pooling.shard_range(T, rank, world)
    c = pooling.pooling_weights(T, alpha)
    # this rank only ever sees its shard (plus, on a GPU, the 1-frame halo for pair metrics)
    part = torch.tensor([float(np.dot(c[lo:hi], series[lo:hi])), float(hi - lo)], dtype=torch.float64)
    dist.all_reduce(part, op=dist.ReduceOp.SUM)
    # one-stream-per-rank summary: sum of per-rank pooled scalars and frame counts
    mine = pooling.pooled_mean(series[lo:hi], alpha) if hi > lo else 0.0
    summ = torch.tensor([mine, 1.0], dtype=torch.float64)
    dist.all_reduce(summ, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(out_path, np.array([part[0].item(), part[1].item(), summ[0].item(), summ[1].item()]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_pooling_allreduce_matches_single_process(tmp_path):
    rng = np.random.default_rng(7)
    series = rng.normal(2.2e6, 3e5, 299)
    alpha = 0.8
    out = str(tmp_path / "r.npy")
    mp.spawn(_worker, args=(2, _free_port(), series, alpha, out), nprocs=2, join=True)
    pooled, count, ssum, nranks = np.load(out)
    assert count == 299 and nranks == 2
    assert abs(pooled - pooling.pooled_mean(series, alpha)) <= 1e-12 * abs(pooled)
    lo, hi = pooling.shard_range(299, 0, 2)
    assert abs(ssum - (pooling.pooled_mean(series[:hi], alpha) + pooling.pooled_mean(series[hi:], alpha))) < 1e-6


def test_shard_ranges_cover_everything():
    for T in (0, 1, 7, 64, 299):
        for world in (1, 2, 3, 8):
            r = [pooling.shard_range(T, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == T
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1

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


# ---- the product's sharded aggregator, with a CPU stand-in for the kernels ---------------------
def _oracle_series(video, resize_width, resize_height, frame_interval=10, batch_size=100, shard=None):
    """Test stand-in for complexity_series (same contract, oracle arithmetic): lets the sharding /
    global-index / all-reduce logic of calculate_average_scene_complexity_sharded run without a GPU."""
    from oracle import pipeline as pl
    from rtvqa_amd import complexity_metrics as cm
    frames = list(np.asarray(video))
    idx = cm.selected_indices(len(frames), frame_interval)
    T = max(len(idx) - 1, 0)
    lo, hi = pooling.shard_range(T, *shard) if shard else (0, T)
    out = {k: [] for k in ("motion", "dct", "hist", "edge", "orb", "color", "temporal")}
    out["range"] = (lo, hi)
    from oracle import c_oracle as co
    for j in range(lo, hi):
        f, p = frames[idx[j + 1]], frames[idx[j]]
        out["motion"].append(pl.process_frame_complexity((f, p)))
        out["dct"].append(pl.process_dct_frame(f, resize_width, resize_height, "full"))
        out["hist"].append(pl.process_histogram_frame(f, resize_width, resize_height))
        out["edge"].append(pl.process_edge_frame(f, resize_width, resize_height))
        out["orb"].append(pl.process_orb_frame_for_parallel(f))
        out["color"].append(pl.process_color_histogram_frame(f, resize_width, resize_height))
        if j >= 1:
            out["temporal"].append(pl.process_temporal_dct_frame(co.bgr2gray(p), co.bgr2gray(f), resize_width,
                                                                 resize_height, "full"))
    return out


def _sharded_worker(rank, world, port, clip, interval, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rtvqa_amd import complexity_metrics as cm
    got = cm.calculate_average_scene_complexity_sharded(clip, 32, 32, frame_interval=interval, series_fn=_oracle_series)
    np.save(out_path % rank, np.array(got, np.float64))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_scene_complexity_equals_single_process(tmp_path):
    from oracle import pipeline as pl
    rng = np.random.default_rng(11)
    # (world 8 = the node's width: 75/10 leaves two ranks with one sample and six with none of the temporal series'
    # tail; 41/10 gives 3 samples to 8 ranks, so five ranks hold nothing and still take part in the reduction)
    for n, interval, world in ((75, 10, 2), (41, 5, 3), (15, 10, 2), (25, 10, 2), (75, 10, 8), (41, 10, 8)):
        clip = rng.integers(0, 256, (n, 40, 56, 3), dtype=np.uint8)
        clip[:, 18:22, 26:30] = 255  # something for the corner detector at the thumbnail's centre
        out = str(tmp_path / ("r%d_%%d.npy" % n))
        mp.spawn(_sharded_worker, args=(world, _free_port(), clip, interval, out), nprocs=world, join=True)
        want = np.array(pl.calculate_average_scene_complexity(list(clip), 32, 32, frame_interval=interval), np.float64)
        for r in range(world):
            got = np.load(out % r)
            assert got.shape == (8,)
            for k in range(8):
                if np.isnan(want[k]):
                    assert np.isnan(got[k]), (n, k)
                else:
                    assert abs(got[k] - want[k]) <= 1e-9 * max(abs(want[k]), 1e-30), (n, r, k, got[k], want[k])

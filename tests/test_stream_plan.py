"""stream.plan_chunks - the arithmetic of a pass (which frames a chunk uploads, where its samples and their previous frames
sit) - against a brute-force model of the reference's selection (complexity_metrics.py:103-104: 1-based count % interval ==
0; :268-290: sample j measures selected frame S_{j+1} against S_j; :533-537).  No engine, no GPU: every (n, interval, cap,
shard) combination is simulated on a buffer of frame numbers."""
import itertools

import numpy as np
import pytest

from rtvqa_amd import stream
from rtvqa_amd.pooling import shard_range


def _simulate(n, want_q, interval, lo, hi, cap):
    """run the plans on buffers of frame ids; -> (quality frames in order, [(frame, prev)] per sample in order, frames uploaded)"""
    qframes, samples, uploaded = [], [], 0
    for p in stream.plan_chunks(n, want_q, interval, lo, hi, cap):
        buf = np.full(cap + 1, -1)
        for slot, start, count, step in p["copies"]:
            assert 0 <= slot and slot + count <= cap + 1, (p, "copy leaves the buffer")
            src = np.arange(start, start + (count - 1) * step + 1, step)
            assert src[-1] < n
            buf[slot:slot + count] = src
            uploaded += count
        if want_q:
            assert p["rcopies"] == [(0, p["q0"], p["qn"], 1)]
            got = buf[1:1 + p["qn"]]
            assert (got == np.arange(p["q0"], p["q0"] + p["qn"])).all()
            qframes.extend(got.tolist())
        for i in range(p["j1"] - p["j0"]):
            f = buf[1 + p["first"] + i * p["step"]]
            prev = buf[1 + p["prev_slot"]] if i == 0 else buf[1 + p["first"] + (i - 1) * p["step"]]
            assert f >= 0 and prev >= 0, (p, "reads a slot nobody uploaded")
            samples.append((int(f), int(prev)))
        if p["j1"] > p["j0"]:
            assert p["j0"] == lo + len(samples) - (p["j1"] - p["j0"])    # samples arrive in series order, none skipped
    return qframes, samples, uploaded


@pytest.mark.parametrize("want_q", [False, True])
def test_every_chunking_measures_exactly_the_reference_samples(want_q):
    for n, interval, cap in itertools.product((0, 1, 2, 5, 9, 10, 11, 29, 30, 31, 64, 101), (1, 2, 3, 7, 10), (1, 2, 3, 8, 100)):
        sel = [t for t in range(n) if (t + 1) % interval == 0]                 # :103-104
        want = [(sel[j + 1], sel[j]) for j in range(len(sel) - 1)]               # :268-290, previous = the selected frame before
        shards = [None] if want_q else [None, (0, 2), (1, 2), (2, 3), (7, 8)]
        for shard in shards:
            lo, hi = shard_range(len(want), *shard) if shard else (0, len(want))
            q, s, up = _simulate(n, want_q, interval, lo, hi, cap)
            assert s == want[lo:hi], (n, interval, cap, shard)
            if want_q:
                assert q == list(range(n))
                # one upload per frame, plus at most one halo frame per chunk
                assert n <= up <= n + -(-n // cap)
            elif hi > lo:
                # only selected frames ever move: the samples' frames and one previous frame per chunk
                assert up == (hi - lo) + -(-(hi - lo) // cap)


def test_quality_only_passes_have_no_samples():
    plans = stream.plan_chunks(25, True, None, 0, 0, 10)
    assert [(p["q0"], p["qn"]) for p in plans] == [(0, 10), (10, 10), (20, 5)]
    assert all(p["j0"] == p["j1"] for p in plans)


def test_split_passes_move_every_byte_of_either_stream_once():
    """split = the quality pair (planar planes, dense) and the encoded BGR stream (the chunk's selected frames, compact) are
    different bytes: the quality copies cover 0..n-1 once, the BGR copies are exactly the samples' frames plus one previous
    frame per chunk that has samples, and the samples are the reference's (:103-104, :268-290)."""
    for n, interval, cap in itertools.product((0, 1, 2, 9, 10, 11, 30, 31, 64, 101), (1, 2, 3, 7, 10), (1, 3, 8, 100)):
        sel = [t for t in range(n) if (t + 1) % interval == 0]
        want = [(sel[j + 1], sel[j]) for j in range(len(sel) - 1)]
        qframes, samples, moved, chunks_with_samples = [], [], 0, 0
        for p in stream.plan_chunks(n, True, interval, 0, len(want), cap, split=True):
            assert p["qslot"] == 0 and p["rcopies"] == p["qcopies"] == [(0, p["q0"], p["qn"], 1)]
            qframes.extend(range(p["q0"], p["q0"] + p["qn"]))
            m = p["j1"] - p["j0"]
            if not m:
                assert p["copies"] == []
                continue
            chunks_with_samples += 1
            buf = np.full(m + 1, -1)
            for slot, start, count, step in p["copies"]:
                assert 0 <= slot and slot + count <= m + 1
                buf[slot:slot + count] = np.arange(start, start + (count - 1) * step + 1, step)
                moved += count
            assert (buf >= 0).all() and buf.max() < n
            for i in range(m):
                f = buf[1 + p["first"] + i * p["step"]]
                prev = buf[1 + p["prev_slot"]] if i == 0 else buf[1 + p["first"] + (i - 1) * p["step"]]
                samples.append((int(f), int(prev)))
                assert q0_le(p, f)      # a sample's frame lies in the chunk's own source range
        assert qframes == list(range(n)) and samples == want, (n, interval, cap)
        assert moved == len(want) + chunks_with_samples


def q0_le(p, f):
    return p["q0"] <= f < p["q0"] + p["qn"]


def test_chunk_frames_follows_the_references_batch_size_and_the_link():
    """stream.chunk_frames: batch_size counts SELECTED frames (complexity_metrics.py:128, :268-290), so a fused chunk spans
    batch_size * interval source frames; chunks that cross PCIe are capped in bytes but hold at least 16 frames of big formats."""
    fb = 1080 * 1920 * 3
    assert stream.chunk_frames(100, 10, 0, False) == 1000          # config.json's defaults on a resident clip: <= 100 samples per launch
    assert stream.chunk_frames(100, 1, 0, False) == 100 and stream.chunk_frames(100, None, 0, False) == 100
    assert stream.chunk_frames(64, None, 0, False) == 64            # quality only
    assert stream.chunk_frames(100, 10, 2 * fb, False) == (256 << 20) // (2 * fb) == 21     # a 1080p BGR pair from host memory
    assert stream.chunk_frames(100, 10, 2 * fb, True) == 21
    assert stream.chunk_frames(5, 1, 2 * fb, False) == 5            # the caller's batch_size wins when it is smaller
    assert stream.chunk_frames(100, 1, 8 * fb, False) == 16         # 2160p pairs (49.8 MB): 256 MiB would be 5 of them
    assert stream.chunk_frames(100, 1, 32 * fb, False) == (1 << 30) // (32 * fb) == 5      # 8K pairs: 16 of them exceed 1 GiB
    assert stream.chunk_frames(100, 1, 4 * (1 << 30), False) == 1   # a frame larger than every cap still moves
    assert stream.chunk_frames(100, 10, fb // 10 + 2 * (fb // 2), False) == (256 << 20) // (fb // 10 + fb)   # c1ref: planar pair + a tenth of a BGR frame

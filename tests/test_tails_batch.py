"""The batch float tails (tails.py) return the SAME BITS as the per-record forms - which the real reference's output
pins (tests/test_golden_pipeline.py) - on adversarial bins: every count of live bins (the masked sum's grouping),
empty rows, single-bin frames, 1080p- and 2160p-sized totals."""
import numpy as np
import pytest

from rtvqa_amd import _native as N
from rtvqa_amd import tails
from rtvqa_amd.engine import FRAME_DTYPE


def _bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def _counts(rng, n, total):
    c = np.zeros((n, 256), np.uint32)
    for i in range(n):
        k = int(rng.integers(1, 257)) if i % 3 else 1 + i % 256   # every k from 1 to 256 occurs
        idx = rng.choice(256, k, replace=False)
        w = rng.random(k) ** (1 + i % 4)
        v = np.floor(w / w.sum() * total).astype(np.int64)
        v[0] += total - v.sum()
        c[i, idx] = v
    return c


@pytest.mark.parametrize("total", [64 * 64, 1920 * 1080, 3840 * 2160])
def test_gray_entropy_batch_is_bit_identical(total):
    rng = np.random.default_rng(total)
    c = _counts(rng, 800, total)
    c[5] = 0                        # an empty row: -0.0, as the per-record form
    c[6] = 0; c[6, 200] = total     # one bin: entropy -0.0
    want = np.array([tails.gray_entropy(r) for r in c], np.float32)
    got = tails.gray_entropy_batch(c)
    assert got.dtype == np.float32 and np.array_equal(_bits(got), _bits(want))


@pytest.mark.parametrize("total", [64 * 64, 1920 * 1080])
def test_color_entropy_batch_is_bit_identical(total):
    rng = np.random.default_rng(total + 1)
    c = np.stack([_counts(rng, 300, total) for _ in range(3)], axis=1)
    c[7, 1] = 0                     # an empty channel: NaN (:464-465)
    want = [tails.color_entropy(r) for r in c]
    got = tails.color_entropy_batch(c)
    assert got.dtype == np.float32
    for g, w in zip(got, want):
        assert (np.isnan(g) and np.isnan(w)) or _bits(g) == _bits(w)


def test_scalars_have_the_per_record_types_and_values():
    rng = np.random.default_rng(3)
    rec = np.zeros(64, FRAME_DTYPE)
    rec["hist_gray"] = _counts(rng, 64, 4096)
    rec["hist_bgr"] = np.stack([_counts(rng, 64, 4096) for _ in range(3)], axis=1)
    rec["dct_energy"] = rng.random(64) * 1e9
    rec["temporal_dct_l1"] = rng.random(64) * 1e6
    rec["edge_count"] = rng.integers(0, 4096, 64)
    rec["orb_keypoints"] = rng.integers(0, 2, 64)
    rec["mv_d2_hist"] = rng.integers(0, 50, (64, 129))
    rec["sad_blocks"] = rec["mv_d2_hist"].sum(axis=1)
    rec["sad_blocks"][3] = 0
    rec["flow_mag_mean"] = rng.random(64)
    for mode in (N.MOTION_SAD, N.MOTION_FARNEBACK):
        for kind in ("motion", "dct", "temporal", "hist", "color", "edge", "orb"):
            got = tails.scalars(kind, rec, mode)
            want = [tails.scalar(kind, r, mode) for r in rec]
            assert len(got) == 64
            for g, w in zip(got, want):
                assert type(g) is type(w), (kind, type(g), type(w))
                assert g == w or (g != g and w != w), (kind, g, w)

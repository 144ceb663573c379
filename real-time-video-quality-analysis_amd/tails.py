"""counts -> reference scalars: the float tails the reference runs in NumPy after each cv2 call.

The HIP kernels return integer bins / sums (include/vqa.h vqa_frame_metrics); what the reference then does with
them on the host (complexity_metrics.py:413-414, :467-473, :504, :343, :364) happens here, with the same NumPy
expressions on the same dtypes, so a value is bit-identical to the one the reference's process_* callable returns
for the same counts (tests/test_golden_pipeline.py pins that against the real reference's output).

Two forms of every tail: per record (the per-frame callables) and per BATCH of records - one pass over the
[n,256] / [n,3,256] count arrays instead of n x 7 Python calls (40 us per frame, which capped the reference-shaped
API at ~25 k frames/s on one host thread, below the 37 k the kernels deliver).  The batch forms return the SAME
bits as the per-record forms (tests/test_tails_batch.py checks them against each other on adversarial bins): every
elementwise step is the same ufunc on the same values, and every reduction runs NumPy's pairwise sum over the
same elements in the same order (a reduction over the last axis of a C-contiguous array is that sum, row by row).
"""
import numpy as np

from . import _native as N


# ---------------------------------------------------------------------------
# per record
# ---------------------------------------------------------------------------
def gray_entropy(counts):
    hist = counts.astype(np.float32).reshape(256, 1)      # calcHist returns float32 (256,1)
    hist = hist / hist.sum()                              # :413
    return -np.sum(hist[hist > 0] * np.log2(hist[hist > 0]))  # :414


def color_entropy(counts_bgr):
    hist_b, hist_g, hist_r = (counts_bgr[c].astype(np.float32).reshape(256, 1) for c in range(3))
    sb, sg, sr = hist_b.sum(), hist_g.sum(), hist_r.sum()
    if sb == 0 or sg == 0 or sr == 0:                     # :464-465
        return float("nan")
    hist_b, hist_g, hist_r = hist_b / sb, hist_g / sg, hist_r / sr
    return -(np.sum(hist_b * np.log2(hist_b + 1e-8)) + np.sum(hist_g * np.log2(hist_g + 1e-8)) +
             np.sum(hist_r * np.log2(hist_r + 1e-8)))     # :471-473


_SQRT_K = np.sqrt(np.arange(129, dtype=np.float64))


def motion_magnitude(rec, motion_mode):
    if motion_mode == N.MOTION_FARNEBACK:
        return np.float32(rec["flow_mag_mean"])  # np.mean of a float32 array is a float32 (:343)
    nb = int(rec["sad_blocks"])
    if nb == 0:
        return np.float32(0.0)
    return np.float32(np.dot(rec["mv_d2_hist"].astype(np.float64), _SQRT_K) / nb)


def scalar(kind, rec, motion_mode=N.MOTION_SAD):
    if kind == "dct":
        return np.float32(rec["dct_energy"])
    if kind == "temporal":
        return np.float32(rec["temporal_dct_l1"])
    if kind == "hist":
        return gray_entropy(rec["hist_gray"])
    if kind == "color":
        return color_entropy(rec["hist_bgr"])
    if kind == "edge":
        return np.int64(rec["edge_count"])
    if kind == "motion":
        return motion_magnitude(rec, motion_mode)
    if kind == "orb":
        return int(rec["orb_keypoints"])
    raise KeyError(kind)


# ---------------------------------------------------------------------------
# per batch of records
# ---------------------------------------------------------------------------
def gray_entropy_batch(counts):
    """[n,256] integer bins -> float32 [n], element i == gray_entropy(counts[i]) to the bit."""
    h = np.ascontiguousarray(counts).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        p = h / h.sum(axis=1)[:, None]                    # the row sums are the pairwise sums hist.sum() forms
        live = p > 0
        t = p * np.log2(p)                                # same ufuncs on the same values; dead bins are dropped below
    k = live.sum(axis=1)
    out = np.empty(len(h), np.float32)
    # np.sum(hist[hist > 0] * ...) sums the COMPACTED live bins: pairwise summation over k elements groups them by
    # their compacted position, so rows are summed in groups of equal k from a dense [rows, k] array
    for kk in np.unique(k):
        rows = np.flatnonzero(k == kk)
        if kk == 0:                                       # an all-zero row: 0/0 bins are NaN, none is > 0, the sum is empty
            out[rows] = -np.float32(0.0)
            continue
        out[rows] = -(t[rows][live[rows]].reshape(len(rows), kk).sum(axis=1))
    return out


def color_entropy_batch(counts_bgr):
    """[n,3,256] integer bins -> float32 [n], element i == color_entropy(counts_bgr[i]) to the bit (NaN where a
    channel is empty, :464-465)."""
    h = np.ascontiguousarray(counts_bgr).astype(np.float32)
    s = h.sum(axis=2)
    with np.errstate(divide="ignore", invalid="ignore"):
        p = h / s[:, :, None]
        e = (p * np.log2(p + 1e-8)).sum(axis=2)
    out = -(e[:, 0] + e[:, 1] + e[:, 2])
    out[(s == 0).any(axis=1)] = np.nan
    return out


def motion_magnitude_batch(rec, motion_mode):
    if motion_mode == N.MOTION_FARNEBACK:
        return rec["flow_mag_mean"].astype(np.float32)
    nb = rec["sad_blocks"].astype(np.float64)
    hist = rec["mv_d2_hist"].astype(np.float64)
    # row by row through np.dot, as the per-record form: a matrix-vector product may sum in another order
    dots = np.fromiter((np.dot(row, _SQRT_K) for row in hist), np.float64, len(hist))
    with np.errstate(divide="ignore", invalid="ignore"):
        out = (dots / nb).astype(np.float32)
    out[nb == 0] = np.float32(0.0)
    return out


def values(kind, rec, motion_mode=N.MOTION_SAD):
    """The series of `kind` for a batch of records as ONE NumPy array (float32 / int64 / uint32, the dtype scalar() returns)."""
    if kind == "dct":
        return rec["dct_energy"].astype(np.float32)
    if kind == "temporal":
        return rec["temporal_dct_l1"].astype(np.float32)
    if kind == "hist":
        return gray_entropy_batch(rec["hist_gray"])
    if kind == "color":
        return color_entropy_batch(rec["hist_bgr"])
    if kind == "edge":
        return rec["edge_count"].astype(np.int64)
    if kind == "motion":
        return motion_magnitude_batch(rec, motion_mode)
    if kind == "orb":
        return np.ascontiguousarray(rec["orb_keypoints"])
    raise KeyError(kind)


def as_list(kind, v):
    """values() as a list whose items have the types scalar() returns (NumPy scalars; Python ints for the ORB count; a
    Python float NaN where a colour channel is empty, :464-465)"""
    if kind == "orb":
        return v.tolist()
    out = list(v)
    if kind == "color" and np.isnan(v).any():
        out = [float("nan") if x != x else x for x in out]
    return out


def scalars(kind, rec, motion_mode=N.MOTION_SAD):
    """The series of `kind` for a batch of records, as a list whose items have the types scalar() returns."""
    return as_list(kind, values(kind, rec, motion_mode))

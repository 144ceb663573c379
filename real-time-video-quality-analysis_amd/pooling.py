"""Temporal pooling of per-frame series — host-side float64 math.

Mirrors smooth_data (complexity_metrics.py:114-125: pandas
``Series.ewm(alpha=alpha).mean()``, adjust=True) followed by ``np.mean``
(:302-309), and restates the pair as a fixed linear functional so a series
sharded across GPUs can be pooled with one scalar all-reduce.
"""
import numpy as np


def _smooth_loop(x, alpha):
    """the recurrence as pandas runs it: num_t = num_{t-1} (1-a) + x_t, den_t = den_{t-1} (1-a) + 1, y_t = num_t / den_t"""
    out = []
    num = 0.0
    den = 0.0
    decay = 1.0 - alpha
    for v in x.tolist():  # Python floats are IEEE doubles: the same arithmetic as on NumPy scalars, five times faster
        num = num * decay + v
        den = den * decay + 1.0
        out.append(num / den)
    return np.array(out, dtype=np.float64)


_TAPS = {}


def _taps(alpha):
    """(1-a)^i for as many i as a double can see next to (1-a)^0 (2^-60 relative), and their running sums; None when the
    decay is too slow for a short window (then the loop runs)"""
    if alpha not in _TAPS:
        decay = 1.0 - alpha
        k = 1
        if decay > 0.0:
            k = int(np.ceil(-60.0 * np.log(2.0) / np.log(decay))) + 1 if decay < 1.0 else 1 << 30
        _TAPS[alpha] = None if k > 96 else (decay ** np.arange(k, dtype=np.float64), np.cumsum(decay ** np.arange(k, dtype=np.float64)))
    return _TAPS[alpha]


def smooth_data(data, alpha=0.8):
    """pandas ewm(alpha, adjust=True).mean():  y_t = sum_i (1-a)^i x_{t-i} / sum_i (1-a)^i.
    For the reference's alpha (0.8: the 27th tap is below 2^-60) the sum is formed as written - one short convolution,
    the same value as the recurrence to ~1e-16 relative - instead of a Python loop per sample (0.27 of the 0.5 ms the
    entry point's pooling took per call); slow decays, short series and series with non-finite samples take the loop."""
    x = np.asarray(data, dtype=np.float64).reshape(-1)
    taps = _taps(alpha) if 0.0 < alpha <= 1.0 else None
    if taps is None or x.size < 16 or not np.isfinite(x).all():
        return _smooth_loop(x, alpha)
    w, cs = taps
    num = np.convolve(x, w)[:x.size]
    den = np.full(x.size, cs[-1])
    m = min(x.size, cs.size)
    den[:m] = cs[:m]
    return num / den


def pooled_mean(data, alpha=0.8):
    """np.mean(smooth_data(data)); NaN for an empty series, as np.mean([]) gives the reference."""
    s = smooth_data(data, alpha)
    if s.size == 0:
        return float("nan")
    return float(np.mean(s))


def pooling_weights(T, alpha=0.8):
    """c_i with mean(ewm(x)) == sum_i c_i x_i for a series of length T.

    c_i = (1/T) sum_{t>=i} (1-a)^(t-i) / S_t,  S_t = sum_{j<=t} (1-a)^j.
    Data-independent, so rank r can form sum_{i in shard r} c_i x_i from GLOBAL
    indices and a SUM all-reduce of that one float64 gives the pooled value.
    """
    if T <= 0:
        return np.zeros(0, np.float64)
    decay = 1.0 - alpha
    S = np.cumsum(decay ** np.arange(T, dtype=np.float64))
    c = np.zeros(T, np.float64)
    # c_i = (1/T) * sum_{t=i}^{T-1} decay^(t-i) / S_t   — backwards recurrence
    acc = 0.0
    for i in range(T - 1, -1, -1):
        acc = acc * decay + 1.0 / S[i]
        c[i] = acc / T
    return c


def shard_range(T, rank, world):
    """Contiguous split of T items over `world` ranks (first T % world ranks get one more)."""
    base, rem = divmod(T, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)

"""Per-frame checker: what the oracle expects the C ABI's result records to hold for one frame.

TEST INFRASTRUCTURE ONLY (see vqa_oracle.c).  Used by tests/ (the batch-geometry parity tests) and by
bench.py's cpu_baseline leg, where the expected records of a few frames of the synthetic stream are computed on
the host BEFORE the process touches the GPU and compared, outside the timed region, with the records of the
LAST TIMED step (`"verified"` in the bench line).  Never on the product path.

Bars (BASELINE.json north_star): bit-exact for histogram bins, edge counts, SAD sums / motion histograms, SSE
and the ORB count; 1e-4 relative for the DCT and SSIM floats.
"""
import numpy as np

from . import c_oracle as co
from . import pipeline as pl

RTOL = 1e-4
FLOW_RTOL_BORDER = 2e-3  # include/vqa.h, flow_mag_mean: the bar on frames with a border pixel at the in-frame discontinuity


def bgr_planes(h, w):
    return [(w, h, c, 3 * w, 3) for c in range(3)]


def expected(ref, dist, prev, full=True, ssim_modes=("gauss",), motion="sad", planes=None, qpair=None,
             dct_mode="block8"):
    """ref, dist, prev: uint8 [h,w,3] BGR (ref/dist = the quality pair, prev = the frame before dist in the
    distorted stream); qpair = (ref, dist) buffers in another pixel format (yuv420p) with their `planes`.
    dct_mode "full": the energy is checked against the Parseval known answer, the temporal L1 against
    scipy.fft.dctn(norm="ortho") in float64 (the C oracle's O(n^3) full-frame transform of a 1080p plane takes minutes;
    the two agree to 1e-6 where both run, tests/test_oracle_kat.py).  -> dict of expected record fields."""
    h, w = dist.shape[:2]
    out = {}
    if ref is not None:
        qr, qd = qpair if qpair is not None else (ref, dist)
        for mode in ssim_modes:
            sse, ssim = pl.frame_quality(qr, qd, planes or bgr_planes(h, w), mode)
            out["sse"] = [int(v) for v in sse]
            out["ssim_" + mode] = [float(v) for v in ssim]
    g = co.bgr2gray(dist)
    gp = co.bgr2gray(prev) if prev is not None else None
    if dct_mode == "block8":
        e, l1, _ = co.dct8x8(gp, g)
        out["dct_energy"], out["temporal_dct_l1"] = float(e), float(l1)
    elif gp is not None:
        from scipy.fft import dctn
        out["temporal_dct_l1"] = float(np.abs(dctn(gp.astype(np.float64), norm="ortho") - dctn(g.astype(np.float64), norm="ortho")).sum())
    out["sum_gray2"] = int((g.astype(np.int64) ** 2).sum())
    if full:
        cnt, strong, weak = co.canny(g, 100, 200)
        out["edge"] = [int(cnt), int(strong), int(weak)]
        if motion == "sad":
            nb, sad, hist = co.block_sad(gp, g, 7)
            out["sad"] = [int(nb), int(sad)]
            out["mv_d2_hist"] = hist.astype(np.uint32)
        else:
            out["flow_mag_mean"] = float(pl.process_frame_complexity((dist, prev), motion="farneback"))
        out["hist_gray"] = co.hist_u8(g)
        out["hist_bgr"] = np.stack([co.hist_u8(dist, offset=c, step=3) for c in range(3)])
        out["orb"] = int(pl.process_orb_frame_for_parallel(dist))
    return out


def _rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-30)


def compare(exp, crec=None, qrow=None, ssim_mode="gauss", notes=None):
    """Mismatches between the expected dict and one vqa_frame_metrics record / one row of vqa_plane_metrics.
    -> list of strings (empty = the frame verifies).  notes (a list) receives a line for every field that only met a
    documented looser bar (Farneback's border discontinuity)."""
    bad = []
    if qrow is not None and "sse" in exp:
        for p, (s, m) in enumerate(zip(exp["sse"], exp["ssim_" + ssim_mode])):
            if int(qrow[p]["sse"]) != s:
                bad.append("sse[%d] %d != %d" % (p, int(qrow[p]["sse"]), s))
            if _rel(qrow[p]["ssim"], m) > RTOL:
                bad.append("ssim_%s[%d] %.9g vs %.9g" % (ssim_mode, p, float(qrow[p]["ssim"]), m))
    if crec is None:
        return bad
    for f in ("dct_energy", "temporal_dct_l1"):
        if f not in exp:
            continue
        if exp[f] == 0.0:
            if float(crec[f]) != 0.0:
                bad.append("%s %.9g != 0" % (f, float(crec[f])))
        elif _rel(crec[f], exp[f]) > RTOL:
            bad.append("%s %.9g vs %.9g" % (f, float(crec[f]), exp[f]))
    if _rel(crec["dct_energy"], exp["sum_gray2"]) > RTOL:  # Parseval known answer
        bad.append("dct_energy %.9g vs sum gray^2 %d" % (float(crec["dct_energy"]), exp["sum_gray2"]))
    if "edge" in exp:
        got = [int(crec["edge_count"]), int(crec["edge_strong"]), int(crec["edge_weak"])]
        if got != exp["edge"]:
            bad.append("edge (count,strong,weak) %s != %s" % (got, exp["edge"]))
        if int(crec["hyst_overflow"]) == 1:   # (2 = completed by the rescue pass: the count above is exact; 1 is never returned)
            bad.append("hyst_overflow set")
        if int(crec["sum_gray2"]) != exp["sum_gray2"]:
            bad.append("sum_gray2 %d != %d" % (int(crec["sum_gray2"]), exp["sum_gray2"]))
        if "sad" in exp:
            got = [int(crec["sad_blocks"]), int(crec["sad_sum"])]
            if got != exp["sad"]:
                bad.append("sad (blocks,sum) %s != %s" % (got, exp["sad"]))
            if not np.array_equal(crec["mv_d2_hist"], exp["mv_d2_hist"]):
                bad.append("mv_d2_hist differs")
        else:
            r = _rel(crec["flow_mag_mean"], exp["flow_mag_mean"])
            # the looser bar (a border pixel on the in-frame test's discontinuity, include/vqa.h) applies only to a caller
            # that passes `notes` and so records which bar held; without it the strict bar stands
            if r > (FLOW_RTOL_BORDER if notes is not None else RTOL):
                bad.append("flow_mag_mean %.9g vs %.9g" % (float(crec["flow_mag_mean"]), exp["flow_mag_mean"]))
            elif r > RTOL:  # both values are valid evaluations; say which bar held
                notes.append("flow_mag_mean met %.0e, not %.0e (%.3g relative)" % (FLOW_RTOL_BORDER, RTOL, r))
        if not np.array_equal(crec["hist_gray"], exp["hist_gray"]):
            bad.append("hist_gray differs")
        if not np.array_equal(crec["hist_bgr"], exp["hist_bgr"]):
            bad.append("hist_bgr differs")
        if int(crec["orb_keypoints"]) != exp["orb"]:
            bad.append("orb %d != %d" % (int(crec["orb_keypoints"]), exp["orb"]))
    return bad

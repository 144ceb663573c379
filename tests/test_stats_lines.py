"""The FFmpeg-format stats lines (video_processing.py:274-276's stats_file= outputs, which the reference's regexes at :160 /
:166 parse): the chunk-at-a-time writer against the line-at-a-time restatement of vf_psnr.c / vf_ssim.c's print statements,
character for character, including mse 0 -> "inf" and SSIM 1 -> "inf" dB."""
import math
import re

import numpy as np

from rtvqa_amd import video_processing as vp


def _psnr(mse):
    return 10.0 * math.log10(255.0 * 255.0 / mse) if mse > 0 else float("inf")


def _psnr_line(n, sse_row, sizes, comps):  # vf_psnr.c: set_meta() order - mse_avg, mse per component, psnr_avg, psnr per component
    areas = [w * h for w, h in sizes]
    comp_mse = [float(s) / a for s, a in zip(sse_row, areas)]
    total = float(sum(areas))
    mse_avg = sum(m * (a / total) for m, a in zip(comp_mse, areas))
    parts = ["n:%d mse_avg:%0.2f " % (n, mse_avg)] + ["mse_%c:%0.2f " % (c, m) for c, m in zip(comps, comp_mse)]
    parts.append("psnr_avg:%0.2f " % _psnr(mse_avg))
    parts += ["psnr_%c:%0.2f " % (c, _psnr(m)) for c, m in zip(comps, comp_mse)]
    return "".join(parts) + "\n"


def _ssim_line(n, ssim_row, sizes, comps):  # vf_ssim.c: "n:%d Y:%f U:%f V:%f All:%f (%f)\n"
    areas = [w * h for w, h in sizes]
    total = float(sum(areas))
    allv = sum(float(s) * (a / total) for s, a in zip(ssim_row, areas))
    db = 10.0 * math.log10(1.0 / (1.0 - allv)) if allv < 1.0 else float("inf")
    return "n:%d " % n + "".join("%c:%f " % (c.upper(), float(s)) for c, s in zip(comps, ssim_row)) + "All:%f (%f)\n" % (allv, db)


def test_chunk_writer_equals_the_line_writer():
    rng = np.random.default_rng(7)
    for sizes, comps in (([(1920, 1080), (960, 540), (960, 540)], "yuv"), ([(128, 96)] * 3, "rgb"), ([(64, 64)], "y")):
        p = len(sizes)
        sse = rng.integers(0, 9_000_000, (300, p)).astype(np.uint64)
        sse[3] = 0          # identical frames: mse 0.00, psnr inf
        sse[4, 0] = 0
        ssim = rng.random((300, p))
        ssim[7] = 1.0
        ssim[8] = 0.999999999
        assert vp.psnr_stats_lines(11, sse, sizes, comps) == "".join(_psnr_line(11 + i, sse[i], sizes, comps) for i in range(300))
        assert vp.ssim_stats_lines(11, ssim, sizes, comps) == "".join(_ssim_line(11 + i, ssim[i], sizes, comps) for i in range(300))
        assert vp.psnr_stats_line(5, sse[3], sizes, comps) == _psnr_line(5, sse[3], sizes, comps)
        assert "psnr_avg:inf" in vp.psnr_stats_line(5, sse[3], sizes, comps)


def test_the_references_regexes_read_the_first_line(tmp_path):
    """extract_metrics_from_logs (video_processing.py:145-177) keeps the FIRST psnr_avg / All match of each file."""
    sizes = [(16, 16)] * 3
    pl_, sl_ = tmp_path / "p.log", tmp_path / "s.log"
    pl_.write_text(vp.psnr_stats_lines(1, np.array([[256, 512, 768], [1, 1, 1]], np.uint64), sizes, "rgb"))
    sl_.write_text(vp.ssim_stats_lines(1, np.array([[0.5, 0.25, 0.75], [0.9, 0.9, 0.9]]), sizes, "rgb"))
    m = vp.extract_metrics_from_logs(str(pl_), str(sl_), str(tmp_path / "none.json"), "x", 23, 1000, "16x16", 30.0)
    assert m["PSNR"] == float("%0.2f" % (10 * math.log10(65025 / 2.0))) and m["SSIM"] == 0.5
    assert re.match(r"n:1 mse_avg:2\.00 mse_r:1\.00 mse_g:2\.00 mse_b:3\.00 psnr_avg:45\.12 ", pl_.read_text())
    assert "VMAF" not in m and m["CRF"] == 23

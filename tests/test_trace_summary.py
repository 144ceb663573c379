"""scripts/trace_summary.py - the interval arithmetic behind profiles/round*_api_trace_*.json - on a synthetic rocprofv3 trace:
kernel busy / H2D busy / both at once / neither, idle gaps and the per-lane turnaround from the pass's roctx ranges."""
import csv
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(d, name, hdr, rows):
    with open(os.path.join(d, name), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(hdr)
        wr.writerows(rows)


def test_summary_of_a_synthetic_trace(tmp_path):
    d = str(tmp_path)
    us = 1000
    _write(d, "7_kernel_trace.csv", ["Kind", "Queue_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"],
           [["K", 1, "k_a(int)", 1000 * us, 3000 * us], ["K", 2, "k_b", 2500 * us, 4000 * us], ["K", 1, "k_a(int)", 9000 * us, 9500 * us]])
    _write(d, "7_memory_copy_trace.csv", ["Kind", "Direction", "Start_Timestamp", "End_Timestamp"],
           [["C", "MEMORY_COPY_HOST_TO_DEVICE", 500 * us, 2000 * us], ["C", "MEMORY_COPY_DEVICE_TO_HOST", 4000 * us, 4100 * us]])
    _write(d, "7_marker_api_trace.csv", ["Domain", "Function", "Start_Timestamp", "End_Timestamp"],
           [["M", "vqa:call x 0", 0, 5000 * us], ["M", "vqa:upload chunk=0 set=0", 400 * us, 600 * us],
            ["M", "vqa:wait chunk=0 lane=0", 3000 * us, 4200 * us], ["M", "vqa:submit chunk=2 lane=0", 4300 * us, 4400 * us],
            ["M", "vqa:warm x", 6000 * us, 9900 * us]])
    o = json.loads(subprocess.check_output([sys.executable, os.path.join(REPO, "scripts", "trace_summary.py"), d, "selftest"]))
    assert o["timed_calls"] == 1 and o["window_ms"] == 5.0
    assert o["kernel_busy_ms"] == 3.0            # [1, 4) ms: the two kernels overlap; the third lies outside the timed call
    assert o["h2d_busy_ms"] == 1.5 and o["d2h_busy_ms"] == 0.1
    assert o["kernel_and_h2d_overlap_ms"] == 1.0 and o["overlap_frac_of_h2d"] == round(1.0 / 1.5, 4)
    assert o["neither_ms"] == 1.4                # [0, 0.5) and [4.1, 5)
    assert o["idle_gaps"]["count"] == 2 and o["idle_gaps"]["largest_ms"] == [0.9, 0.5] and o["idle_gaps"]["over_100us"] == 2
    assert o["per_queue_kernel_busy_ms"] == {"1": 2.0, "2": 1.5}
    assert o["lane_turnaround_ms"]["lane 0"] == {"count": 1, "mean": 0.1, "max": 0.1}
    assert o["host_stage_ms"]["wait"] == {"count": 1, "total": 1.2, "mean": 1.2} and o["host_stage_ms"]["upload"]["count"] == 1
    assert o["top_kernels_ms"]["k_a"] == {"total": 2.0, "launches": 1}

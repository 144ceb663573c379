"""rocprofv3 --kernel-trace --memory-copy-trace --marker-trace (csv) of scripts/api_trace.py -> one JSON summary:
inside the timed calls ("vqa:call ..." roctx ranges), how long the GPU ran kernels, how long H2D copies ran, how long BOTH
ran at once, how long NEITHER did, per hardware queue and per stream.py lane.

  python3 scripts/trace_summary.py DIR [label [driver log]] > profiles/roundNN_api_trace_<label>.json

Interval arithmetic only (union / intersection of [start, end) ns intervals); column names are looked up by header, so the
script follows rocprofv3's CSV layout rather than assuming positions."""
import csv, glob, json, os, re, sys


def rows(pattern):
    out = []
    for p in sorted(glob.glob(pattern, recursive=True)):
        with open(p, newline="") as f:
            out += list(csv.DictReader(f))
    return out


def col(row, *names):
    for n in names:
        if n in row and row[n] != "":
            return row[n]
    return None


def union(iv):
    iv = sorted((a, b) for a, b in iv if b > a)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    out, i, j = [], 0, 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if b > a:
            out.append([a, b])
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return out


def clip(iv, window):
    return intersect(union(iv), window)


def main():
    d = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(d.rstrip("/"))
    log = sys.argv[3] if len(sys.argv) > 3 else None
    kern = rows(os.path.join(d, "**", "*kernel_trace.csv"))
    cop = rows(os.path.join(d, "**", "*memory_copy_trace.csv"))
    mark = rows(os.path.join(d, "**", "*marker_api_trace.csv"))
    K = [(int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Kernel_Name") or "?", col(r, "Queue_Id") or "?") for r in kern]
    C = [(int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), (col(r, "Direction") or "?")) for r in cop]
    M = [(int(col(r, "Start_Timestamp")), int(col(r, "End_Timestamp")), col(r, "Function", "Name", "Message") or "?") for r in mark]
    calls = union([(a, b) for a, b, name in M if name.startswith("vqa:call")])
    out = {"label": label, "source": "rocprofv3 --kernel-trace --memory-copy-trace --marker-trace -- python3 scripts/api_trace.py",
           "kernels_traced": len(K), "copies_traced": len(C), "marker_ranges": len(M), "timed_calls": len(calls)}
    if log and os.path.isfile(log):  # what the driver printed under the profiler (its rates include the tracing overhead)
        out["driver_output"] = [l.rstrip() for l in open(log) if l.startswith(("resident", "host_", "roctx"))]
    out["git_sha"] = os.environ.get("VQA_GIT_SHA", "") or "unknown"
    if not calls:
        out["error"] = "no 'vqa:call' marker ranges in the trace (markers off?): window = first to last device activity"
        lo = min([a for a, _, _, _ in K] + [a for a, _, _ in C])
        hi = max([b for _, b, _, _ in K] + [b for _, b, _ in C])
        calls = [[lo, hi]]
    win = total(calls)
    kb = clip([(a, b) for a, b, _, _ in K], calls)
    h2d = clip([(a, b) for a, b, dr in C if "HOST_TO_DEVICE" in dr.upper() or dr.upper() in ("H2D", "HTOD")], calls)
    d2h = clip([(a, b) for a, b, dr in C if "DEVICE_TO_HOST" in dr.upper() or dr.upper() in ("D2H", "DTOH")], calls)
    both = intersect(kb, h2d)
    busy = union([tuple(x) for x in kb + h2d + d2h])
    ms = lambda ns: round(ns / 1e6, 3)
    out.update({
        "window_ms": ms(win), "kernel_busy_ms": ms(total(kb)), "h2d_busy_ms": ms(total(h2d)), "d2h_busy_ms": ms(total(d2h)),
        "kernel_and_h2d_overlap_ms": ms(total(both)), "neither_ms": ms(win - total(busy)),
        "kernel_busy_frac": round(total(kb) / win, 4), "h2d_busy_frac": round(total(h2d) / win, 4),
        "overlap_frac_of_h2d": round(total(both) / max(total(h2d), 1), 4),
        "overlap_frac_of_kernel": round(total(both) / max(total(kb), 1), 4),
        "idle_frac": round((win - total(busy)) / win, 4),
    })
    # idle gaps (neither kernel nor copy) inside the calls, largest first
    gaps = []
    for a, b in calls:
        cur = a
        for x, y in intersect(busy, [[a, b]]):
            if x > cur:
                gaps.append(x - cur)
            cur = max(cur, y)
        if b > cur:
            gaps.append(b - cur)
    gaps.sort(reverse=True)
    out["idle_gaps"] = {"count": len(gaps), "largest_ms": [ms(g) for g in gaps[:5]],
                        "over_100us": sum(1 for g in gaps if g > 100e3), "sum_over_100us_ms": ms(sum(g for g in gaps if g > 100e3))}
    # per hardware queue (a HIP stream's queue): kernel busy time inside the calls
    per_q = {}
    for a, b, _, q in K:
        per_q.setdefault(q, []).append((a, b))
    out["per_queue_kernel_busy_ms"] = {q: ms(total(clip(iv, calls))) for q, iv in sorted(per_q.items())}
    # per lane (stream.py's roctx ranges): host time inside each stage, and the lane's device-side idle = from the end of
    # a chunk's wait to the start of the lane's next submit
    stage = {}
    lane_wait_end, lane_gap = {}, {}
    for a, b, name in sorted(M):
        m = re.match(r"vqa:([a-z-]+) chunk=(\d+)(?: (?:lane|set)=(\d+))?", name)
        if not m or not intersect([[a, b]], calls):
            continue
        st, lane = m.group(1), m.group(3)
        stage.setdefault(st, []).append(b - a)
        if lane is not None and st in ("submit", "wait"):
            if st == "submit" and lane in lane_wait_end:
                lane_gap.setdefault(lane, []).append(a - lane_wait_end.pop(lane))
            if st == "wait":
                lane_wait_end[lane] = b
    out["host_stage_ms"] = {k: {"count": len(v), "total": ms(sum(v)), "mean": ms(sum(v) / len(v))} for k, v in sorted(stage.items())}
    out["lane_turnaround_ms"] = {("lane %s" % k): {"count": len(v), "mean": ms(sum(v) / len(v)), "max": ms(max(v))}
                                 for k, v in sorted(lane_gap.items())}
    out["lane_turnaround_what"] = "host time from the end of a lane's wait(chunk k) to the start of its next submit(chunk k + lanes): the lane's engine has nothing enqueued meanwhile"
    # the dominant kernels inside the calls
    per_k = {}
    for a, b, name, _ in K:
        if intersect([[a, b]], calls):
            e = per_k.setdefault(name.split("(")[0][:60], [0, 0])
            e[0] += b - a
            e[1] += 1
    out["top_kernels_ms"] = {k: {"total": ms(v[0]), "launches": v[1]} for k, v in sorted(per_k.items(), key=lambda kv: -kv[1][0])[:8]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

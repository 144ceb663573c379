"""bench.py's host logic on CPU: the N > 1 rank logic (barrier-fenced timing, max over ranks, the scalar all-reduce,
the JSON line's shape) rehearsed under gloo world-2 with the engine stubbed (--stub-engine: no kernels, the line says so),
the fail-loud RCCL rule, the cpu_baseline worker rule and the PMC-traffic staleness check."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402

CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, env=None, timeout=240):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run(cmd, cwd=REPO, env=e, capture_output=True, text=True, timeout=timeout)


def test_default_workload_is_the_headline_config():
    # BASELINE.json's metric is "full metric suite @1080p" = configs[2]
    assert bench.WORKLOADS["c3"]["full"] and (bench.WORKLOADS["c3"]["h"], bench.WORKLOADS["c3"]["w"]) == (1080, 1920)
    r = _run([sys.executable, "bench.py", "--stub-engine", "--backend", "gloo", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-800:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["config"]["id"] == "c3" and line["stub"] is True and line["n_gpus"] == 1
    assert CONTRACT_KEYS <= set(line)


def test_workload_defaults_and_overrides_resolve_in_the_line():
    """The product's configuration is the default (overlap on, two batches in flight on one context each); c3ref and the
    Farneback mode keep one batch in flight; explicit flags win; --steps 0 prints a line instead of crashing."""
    def cfg(*extra):
        r = _run([sys.executable, "bench.py", "--stub-engine", "--backend", "gloo", "--steps", "1", "--warmup", "0", *extra])
        assert r.returncode == 0, r.stderr[-800:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    c = cfg()["config"]
    assert (c["overlap"], c["streams"], c["inflight"], c["motion"], c["dct_mode"], c["ssim_mode"], c["pixfmt"]) == \
        (True, 1, 2, "sad", "block8", "gauss", "bgr24")
    c = cfg("--workload", "c3ref")["config"]
    assert (c["id"], c["streams"], c["inflight"], c["motion"], c["dct_mode"], c["ssim_mode"], c["pixfmt"]) == \
        ("c3ref", 2, 1, "farneback", "full", "ffmpeg", "yuv420p")
    assert cfg("--motion", "farneback")["config"]["inflight"] == 1 and cfg("--dct-mode", "full")["config"]["inflight"] == 2
    c = cfg("--no-overlap", "--inflight", "3", "--streams", "2", "--workload", "c3ref", "--motion", "sad")["config"]
    assert (c["overlap"], c["inflight"], c["streams"], c["motion"]) == (False, 3, 2, "sad")
    line = cfg("--steps", "0")
    assert line["steps"] == 0 and line["value"] == 0.0


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_rank_logic_json_shape(world):
    """The N > 1 rank logic at world 2 and at the node's full width (8): one rank per GPU as the driver launches it."""
    port = _free_port()
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
              "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1",
              "--backend", "gloo", "--stub-engine"])
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly ONE JSON line"
    line = json.loads(lines[0])
    assert CONTRACT_KEYS <= set(line)
    assert line["n_gpus"] == world and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    cfg = line["config"]
    assert cfg["collective"] == "gloo scalar all-reduce" and cfg["rccl_ranks"] is None
    assert cfg["devices"] == list(range(world)) and cfg["parallelism"] == "1 stream/GPU x%d" % world
    # whole-job aggregate: frames of ALL ranks over the max-over-ranks time
    frames = cfg["frames_per_step_per_gpu"] * line["steps"] * world
    assert abs(line["value"] - frames / (line["ms_per_step"] * 1e-3 * line["steps"])) < 1e-3 * line["value"]


def test_gpus_flag_must_match_world_size():
    port = _free_port()
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), "bench.py", "--gpus", "4", "--backend", "gloo", "--stub-engine"])
    assert r.returncode != 0


def test_stub_engine_refuses_nccl():
    r = _run([sys.executable, "bench.py", "--stub-engine"])
    assert r.returncode != 0 and "gloo" in (r.stderr + r.stdout)


def test_rccl_failure_is_fatal_unless_rehearsal(monkeypatch):
    """init_dist: an RCCL bring-up failure exits non-zero (os._exit(3)) unless VQA_BENCH_DEVICE pinned the ranks."""
    code = (
        "import sys, os; sys.path.insert(0, %r)\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=%r, RANK='0', WORLD_SIZE='1')\n"
        "import bench\n"
        "td, used, dev, n = bench.init_dist('nccl', 0, 1, 0, %s, False)\n"
        "print('FELLBACK', used, dev, n)\n"
    )
    # no GPU here: the nccl process group cannot come up
    r = _run([sys.executable, "-c", code % (REPO, str(_free_port()), "False")])
    assert r.returncode == 3 and "FATAL" in r.stderr and "FELLBACK" not in r.stdout
    # the rehearsal (ranks pinned to one device) never touches RCCL: gloo from the start, and it says so
    r = _run([sys.executable, "-c", code % (REPO, str(_free_port()), "True")])
    assert r.returncode == 0 and "FELLBACK gloo cpu None" in r.stdout and "rehearsal" in r.stderr, r.stderr[-600:]


def test_cpu_worker_rule(monkeypatch):
    monkeypatch.delenv("VQA_CPU_WORKERS", raising=False)
    assert bench.cpu_workers(16) == (8, False) and bench.cpu_workers(1) == (1, False) and bench.cpu_workers(192) == (96, False)
    monkeypatch.setenv("VQA_CPU_WORKERS", "5")
    assert bench.cpu_workers(16) == (5, True)
    assert bench.visible_cores() >= 1


def test_pmc_traffic_goes_null_when_sources_changed(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    h = bench.source_hashes()
    good = {"tag": "round9_c3", "frames_per_launch": 128, "git_sha": "abc1234", "source_sha256": h,
            "kernels": {"k_ssim_gauss_p2<256, 8>": {"hbm_bytes": 1000}, "k_ssim_ffmpeg_fast<3>": {"hbm_bytes": 7}}}
    (prof / "round9_c3_pmc.json").write_text(json.dumps(good))
    (prof / "round2_c3_pmc.json").write_text(json.dumps(dict(good, kernels={"k_ssim_gauss<256, 2, 0>": {"hbm_bytes": 1}})))
    t, src = bench.pmc_traffic("c3", "k_ssim_gauss", 256, True)
    assert t == 2000 and src == "profiles/round9_c3_pmc.json @ abc1234"  # newest round wins, scaled to this batch
    stale = dict(good, source_sha256=dict(h, **{"k_quality.hip": "0" * 16}))
    (prof / "round9_c3_pmc.json").write_text(json.dumps(stale))
    t, src = bench.pmc_traffic("c3", "k_ssim_gauss", 256, True)
    assert t is None and src.startswith("stale:")
    for f in ("vqa_capi.hip", "vqa_kernels.hpp", "vqa_dev.hpp"):  # the orchestration and the shared headers count too
        (prof / "round9_c3_pmc.json").write_text(json.dumps(dict(good, source_sha256=dict(h, **{f: "0" * 16}))))
        t, src = bench.pmc_traffic("c3", "k_ssim_gauss", 256, True)
        assert t is None and src.startswith("stale:") and f in src, f
    nohash = {k: v for k, v in good.items() if k != "source_sha256"}
    (prof / "round9_c3_pmc.json").write_text(json.dumps(nohash))
    assert bench.pmc_traffic("c3", "k_ssim_gauss", 256, True)[0] is None
    assert bench.pmc_traffic("c3", "k_ssim_gauss", 256, False)[0] is None  # non-default modes have no PMC pass


def test_verified_checker_accepts_the_oracle_and_rejects_a_wrong_record():
    """bench.py's "verified": the expectation (oracle/check.py, computed before the GPU is touched) against result records.
    A record filled from the oracle itself verifies; one wrong bin / count / float is reported."""
    import numpy as np
    from oracle import check
    from rtvqa_amd.engine import FRAME_DTYPE, PLANE_DTYPE
    assert bench.verify_frames(256) == [0, 1, 15, 16, 17, 128, 255] and bench.verify_frames(2) == [0, 1]
    j, exp = bench._expect_item(("natural", 0, 48, 64, 4, 1, True, "sad", "gauss", False, "block8"))
    assert j == 1
    c = np.zeros(1, FRAME_DTYPE)[0]
    c["dct_energy"], c["temporal_dct_l1"], c["sum_gray2"] = exp["dct_energy"], exp["temporal_dct_l1"], exp["sum_gray2"]
    c["edge_count"], c["edge_strong"], c["edge_weak"] = exp["edge"]
    c["sad_blocks"], c["sad_sum"] = exp["sad"]
    c["mv_d2_hist"], c["hist_gray"], c["hist_bgr"], c["orb_keypoints"] = exp["mv_d2_hist"], exp["hist_gray"], exp["hist_bgr"], exp["orb"]
    q = np.zeros(3, PLANE_DTYPE)
    q["sse"], q["ssim"] = exp["sse"], exp["ssim_gauss"]
    assert check.compare(exp, c, q, "gauss") == []
    c["hist_bgr"][1][7] += 1
    c["edge_count"] += 1
    q["ssim"][2] *= 1.0 + 3e-4
    bad = check.compare(exp, c, q, "gauss")
    assert len(bad) == 3 and any("hist_bgr" in b for b in bad) and any("edge" in b for b in bad) and any("ssim_gauss[2]" in b for b in bad)


def test_dist_always_brings_up_a_group_of_one_without_a_launcher():
    """--dist-always: the N > 1 code path (process group, MAX / SUM all-reduces, all_gather_object, barrier, teardown) with
    the one rank there is, no launcher and no MASTER_* in the environment (bench.py supplies a rendezvous of one).  Here
    under gloo with the engine stubbed; tests/test_gpu_rccl_one_rank.py runs the same path over real RCCL on the GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--stub-engine", "--backend", "gloo", "--dist-always", "--steps", "2", "--warmup", "1"],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-800:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["config"]["collective"] == "gloo scalar all-reduce" and line["config"]["backend"] == "gloo"
    assert line["n_gpus"] == 1 and line["config"]["devices"] == [0]


def test_c1_is_a_workload_and_refuses_the_stub():
    assert bench.WORKLOADS["c1"]["batch"] == 300 and bench.C1_CONFIG["frame_interval"] == 10 and bench.C1_CONFIG["resize_width"] == 64
    r = _run([sys.executable, "bench.py", "--workload", "c1", "--stub-engine", "--backend", "gloo"])
    assert r.returncode != 0 and "c1 has no stub" in (r.stderr + r.stdout)


def test_c1ref_is_c1_with_the_references_own_definitions():
    """c1ref = config.json's five keys plus the three that select what the reference itself computes: FFmpeg's filters on the
    decoded yuv420p planes (video_processing.py:274-276) and Farneback motion (complexity_metrics.py:340); and its CPU side
    - the oracle pipeline with the same definitions - runs here on a tiny clip."""
    import numpy as np
    from rtvqa_amd import synth, video_processing as vp
    wl = bench.WORKLOADS["c1ref"]
    assert wl["ref_true"] and wl["api"] and (wl["h"], wl["w"], wl["batch"]) == (1080, 1920, 300)
    cfg = dict(bench.C1_CONFIG, **bench.C1REF_KEYS)
    vp.validate_config(cfg)
    assert (cfg["ssim_mode"], cfg["pixfmt"], cfg["motion"]) == ("ffmpeg", "yuv420p", "farneback")
    r = _run([sys.executable, "bench.py", "--workload", "c1ref", "--stub-engine", "--backend", "gloo"])
    assert r.returncode != 0 and "c1ref has no stub" in (r.stderr + r.stdout)
    ref = synth.s_natural(21, 72, 96, seed=3)
    dist = synth.distort(ref)
    yr, yd = bench.c1_planar(ref), bench.c1_planar(dist)
    assert yr.shape == (21, 72 * 96 * 3 // 2)
    tup, q = bench.c1_oracle(ref, dist, 1, pool=False, planar=(yr, yd))
    tup_sad, q_bgr = bench.c1_oracle(ref, dist, 1, pool=False)
    assert len(q) == 21 and len(q[0][0]) == 3 and len(tup) == 8
    assert tup[0] != tup_sad[0] and tup[1:] == tup_sad[1:]          # only the motion slot changes with the motion definition
    assert q[0][0] != q_bgr[0][0]                                   # other planes, other sums


@pytest.mark.parametrize("world", [2, 8])
def test_gpus_n_without_a_launcher_starts_n_ranks(world):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: the parent touches no GPU, starts the N ranks as
    fresh torch.distributed.run children and relays rank 0's ONE line and the exit code - round 5's bench printed
    "n_gpus": 1 with rc 0 here, which would have made the first 8-GPU record a 1-GPU measurement."""
    env = {k: None for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e = dict(os.environ)
    for k in env:
        e.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1", "--backend", "gloo", "--stub-engine"],
                       cwd=REPO, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly ONE JSON line"
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["config"]["devices"] == list(range(world))
    assert line["config"]["launched_by"].startswith("bench.py itself") and "starting %d ranks" % world in r.stderr
    frames = line["config"]["frames_per_step_per_gpu"] * line["steps"] * world
    assert abs(line["value"] - frames / (line["ms_per_step"] * 1e-3 * line["steps"])) < 1e-3 * line["value"]


def test_launcherless_run_relays_a_failing_rank():
    """the launcher's exit code is the bench's: a rank that refuses its arguments fails the whole run, and no line is printed"""
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--stub-engine"], cwd=REPO, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # a WORLD_SIZE that disagrees with --gpus stays fatal, also for WORLD_SIZE=1 (one rank started through a launcher)
    r = _run([sys.executable, "bench.py", "--gpus", "2", "--stub-engine", "--backend", "gloo"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def _fake_sysfs(tmp_path, gpus):
    """a sysfs with one CPU node and `gpus` = [(domain, bus, dev, fn, cpulist)] KFD GPU nodes"""
    nodes = tmp_path / "class" / "kfd" / "kfd" / "topology" / "nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for k, (dom, bus, dev, fn, cpus) in enumerate(gpus, 1):
        (nodes / str(k)).mkdir()
        (nodes / str(k) / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain %d\n" % ((bus << 8) | (dev << 3) | fn, dom))
        d = tmp_path / "bus" / "pci" / "devices" / ("%04x:%02x:%02x.%x" % (dom, bus, dev, fn))
        d.mkdir(parents=True)
        (d / "local_cpulist").write_text(cpus + "\n")
    return str(tmp_path)


def test_bind_numa_reads_the_gpus_cpus_from_sysfs_and_binds_before_any_gpu_call(tmp_path, monkeypatch):
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    have = sorted(os.sched_getaffinity(0))
    lo = "%d-%d" % (have[0], have[len(have) // 2]) if len(have) > 1 else str(have[0])
    sysfs = _fake_sysfs(tmp_path, [(0, 0x05, 0, 0, lo), (0, 0x85, 0, 0, "4000-4003"), (1, 0xc5, 0, 0, "%d,%d" % (have[0], have[-1]))])
    assert bench._parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
    cpus, src = bench.gpu_local_cpus(0, sysfs)
    assert cpus == bench._parse_cpulist(lo) and src.endswith("0000:05:00.0/local_cpulist")
    assert bench.gpu_local_cpus(2, sysfs)[0] == {have[0], have[-1]}
    assert bench.gpu_local_cpus(5, sysfs)[0] is None                      # no such GPU: not an error, no binding
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
    assert bench.gpu_local_cpus(0, sysfs)[0] == {have[0], have[-1]}      # device 0 of this process is the node's GPU 2
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-abcdef")
    assert bench.gpu_local_cpus(0, sysfs)[0] is None
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    try:
        rec = bench.bind_numa(0, True, sysfs)
        assert rec["bound"] and rec["cpus"] == len(bench._parse_cpulist(lo) & set(have)) and os.sched_getaffinity(0) == bench._parse_cpulist(lo) & set(have)
        assert bench.bind_numa(1, True, sysfs)["bound"] is False          # the GPU's CPUs are not ours (another cpuset): left alone
        assert bench.bind_numa(0, False, sysfs) == {"bound": False, "why": "--no-bind-numa"}
        assert bench.bind_numa(0, True, str(tmp_path / "nowhere"))["bound"] is False
        os.sched_setaffinity(0, have)
        # an unprivileged container (this pool's GPU boxes): the KFD properties are EPERM, /dev/dri holds exactly the render
        # nodes of the container's GPUs, partition (amdgpu_xcp) nodes have no local_cpulist
        drm = tmp_path / "box"
        for minor, cpus in ((128, "4000-4003"), (129, None), (160, lo), (168, "%d" % have[0])):
            d = drm / "sys" / "class" / "drm" / ("renderD%d" % minor) / "device"
            d.mkdir(parents=True)
            if cpus is not None:
                (d / "local_cpulist").write_text(cpus + "\n")
        dev = drm / "dev"
        dev.mkdir()
        for minor in (129, 160, 168):      # renderD128 is another tenant's: not in the container's /dev/dri
            (dev / ("renderD%d" % minor)).write_text("")
        cpus, src = bench.gpu_local_cpus(0, str(drm / "sys"), str(dev))
        assert cpus == bench._parse_cpulist(lo) and "renderD160" in src
        assert bench.gpu_local_cpus(1, str(drm / "sys"), str(dev))[0] == {have[0]}
        none, why = bench.gpu_local_cpus(2, str(drm / "sys"), str(dev))
        assert none is None and "kfd" in why and "drm" in why
        rec = bench.bind_numa(0, True, str(drm / "sys"), str(dev))
        assert rec["bound"] and "renderD160" in rec["source"]
    finally:
        os.sched_setaffinity(0, have)


def test_cpu_baseline_value_is_the_honest_figure():
    """`value` was the oversubscribed pool (128 workers on a 16-core cgroup, 1.7x slower than 16 workers);
    now the faster of the two pool sizes is `value`, `cores` = the cores that ran it, and the other figure stays beside it."""
    over = dict(value=18.1, workers=128, seconds=10.0, cgroup_cpu_limit=16.0, sample="s",
                at_cgroup_limit=dict(value=30.8, workers=16, seconds=5.0, sample="half"))
    line = bench.finish_cpu_line(dict(over))
    assert line["value"] == 30.8 and line["cores"] == 16 and line["workers"] == 16 and line["value_is"].startswith("at_cgroup_limit")
    assert line["reference_rule"]["value"] == 18.1 and line["reference_rule"]["workers"] == 128 and "at_cgroup_limit" not in line
    same = bench.finish_cpu_line(dict(value=40.0, workers=8, seconds=3.0, cgroup_cpu_limit=None, sample="s"))
    assert same["value"] == 40.0 and same["cores"] == 8 and same["value_is"].startswith("reference_rule")
    slow = bench.finish_cpu_line(dict(value=40.0, workers=32, seconds=3.0, cgroup_cpu_limit=16.0, sample="s",
                                      at_cgroup_limit=dict(value=35.0, workers=16, seconds=2.0, sample="half")))
    assert slow["value"] == 40.0 and slow["cores"] == 16 and slow["at_cgroup_limit"]["value"] == 35.0

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

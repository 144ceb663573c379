"""The N > 1 bring-up over REAL RCCL with the one rank a 1-GPU box has: torch's bundled RCCL next to libvqa_hip.so in
one process - init_process_group("nccl", device_id=...), the probe all-reduce, the MAX / SUM all-reduces of the timing
and the pooled scalars, all_gather_object, barrier - through bench.py (started by the launcher the driver uses) and
through calculate_average_scene_complexity_sharded.  Every child is a FRESH process: the launcher goes first, before
anything in it touches the GPU.  (one scalar all-reduce is the path's only collective.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VQA_BENCH_DEVICE"):
        e.pop(k, None)
    return e


@pytest.mark.parametrize("workload,extra", [("c3", ["--batch", "8"]), ("c1", ["--batch", "30"]), ("c1ref", ["--batch", "30"])])
def test_bench_runs_its_rccl_path_with_one_rank(workload, extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--dist-always", "--workload", workload, "--steps", "2",
           "--warmup", "1", "--cpu-sample", "0", "--e2e-steps", "0", "--api-steps", "0"] + extra
    r = subprocess.run(cmd, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2500:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert cfg["backend"] == "nccl" and cfg["rccl_ranks"] == 1 and cfg["collective"] == "rccl scalar all-reduce", cfg
    assert cfg["devices"] == [0] and line["n_gpus"] == 1 and line["value"] > 0
    assert line["verified"]["ok"] is True
    assert cfg["cpu_affinity"]["bound"] in (True, False) and cfg["launched_by"] == "torch.distributed.run"
    if workload == "c1ref":
        assert (cfg["ssim_mode"], cfg["pixfmt"], cfg["motion"]) == ("ffmpeg", "yuv420p", "farneback")


def test_bench_gpus_1_needs_no_launcher_and_binds_to_the_gpus_cpus():
    """the driver's N = 1 command, plainly; the rank is pinned to the CPUs of its GPU's NUMA node before any GPU call when sysfs
    tells which those are (this pool's boxes: through the render node the container holds)"""
    cmd = [sys.executable, "bench.py", "--gpus", "1", "--batch", "8", "--steps", "2", "--warmup", "1", "--cpu-sample", "0",
           "--e2e-steps", "2", "--e2e-batch", "8", "--api-steps", "0"]
    r = subprocess.run(cmd, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2500:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    aff = line["config"]["cpu_affinity"]
    assert line["n_gpus"] == 1 and line["config"]["launched_by"] == "direct" and line["verified"]["ok"] is True
    # (bound on this pool's boxes - through the render node the container holds; a box whose sysfs tells nothing says why)
    assert (aff["bound"] is True and aff["cpus"] >= 1 and "local_cpulist" in aff["source"]) or (aff["bound"] is False and aff["why"]), aff
    assert line["end_to_end"]["fps"] > 0 and "copy lane" in line["end_to_end"]["overlap"]


_SHARDED = (
    "import os, sys, numpy as np; sys.path.insert(0, %r)\n"
    "import torch, torch.distributed as td\n"
    "td.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
    "from rtvqa_amd import complexity_metrics as cm, synth\n"
    "clip = synth.s_natural(41, 96, 128, seed=8)\n"
    "got = cm.calculate_average_scene_complexity_sharded(clip, 64, 64, frame_interval=3, batch_size=5)\n"
    "want = cm.calculate_average_scene_complexity(clip, 64, 64, frame_interval=3, batch_size=5)\n"
    "assert td.get_backend() == 'nccl'\n"
    "for k, (g, w) in enumerate(zip(got, want)):\n"
    "    assert abs(float(g) - float(w)) <= 1e-12 * max(abs(float(w)), 1e-30), (k, g, w)\n"
    "td.barrier(); td.destroy_process_group()\n"
    "print('SHARDED-RCCL-OK')\n"
)


def test_sharded_aggregator_over_rccl_with_one_rank():
    """calculate_average_scene_complexity_sharded with backend nccl, world 1: the partial sums go through a device tensor
    and RCCL's SUM all-reduce on the engine's device, and come back as the single-process tuple."""
    e = dict(_env(), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-c", _SHARDED % REPO], cwd=REPO, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARDED-RCCL-OK" in r.stdout, (r.stdout[-300:], r.stderr[-2500:])


def test_configs4_shape_with_two_ranks_on_the_one_gpu_and_no_launcher():
    """BASELINE configs[4] in miniature, as far as a 1-GPU box can take it: `bench.py --workload c4 --gpus 2` with NO launcher
    starts its two ranks itself (fresh children), each runs the real 2160p suite on its own stream - here both pinned to
    device 0 (VQA_BENCH_DEVICE: the rehearsal, scalar reductions over gloo because RCCL refuses two ranks on one device) -,
    each verifies its last step against the oracle, and rank 0 prints ONE line for the whole job."""
    e = dict(_env(), VQA_BENCH_DEVICE="0")
    cmd = [sys.executable, "bench.py", "--workload", "c4", "--gpus", "2", "--batch", "4", "--steps", "2", "--warmup", "1",
           "--cpu-sample", "0", "--e2e-steps", "0", "--api-steps", "0"]
    r = subprocess.run(cmd, cwd=REPO, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["id"] == "c4" and cfg["rehearsal_single_device"] is True and cfg["devices"] == [0]
    assert cfg["collective"] == "gloo scalar all-reduce" and cfg["launched_by"].startswith("bench.py itself")
    assert line["verified"]["ok"] is True and line["verified"]["ranks"] == 2
    frames = cfg["frames_per_step_per_gpu"] * line["steps"] * 2
    assert abs(line["value"] - frames / (line["ms_per_step"] * 1e-3 * line["steps"])) < 1e-3 * line["value"]
    assert "starting 2 ranks" in r.stderr

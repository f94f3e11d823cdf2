"""GPU: `python bench.py --gpus 2` must work by itself (no launcher): it starts two fresh rank processes before
touching the GPU; on a 1-GPU box they share the device and rendezvous over gloo (plumbing rehearsal)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--frames", "8", "--streams", "2",
           "--rows", "270", "--cols", "480", "--no-matcher-bench"] + extra
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    one = _run(["--gpus", "1", "--no-cpu-baseline"])
    two = _run(["--gpus", "2"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    # the N > 1 line is a complete record: the host-buffer leg ran on every rank (max over ranks) and rank 0 timed the CPU baseline
    assert two["host_buffers"] is not None and two["host_buffers"]["ranks"] == 2 and two["host_buffers"]["downloaded_equals_device"]
    assert two["value_end_to_end"] == two["host_buffers"]["value"] > 0
    assert two["cpu_baseline"] is not None and two["cpu_baseline"]["value"] > 0 and two["cpu_baseline"]["cores"] >= 1
    assert "rank0_placement" in two["config"]
    assert two["config"]["frames_per_gpu_per_step"] == 8
    # whole-job frames per step doubles with the ranks (weak scaling): value * ms_per_step = frames per step
    f1 = one["value"] * one["ms_per_step"] / 1e3
    f2 = two["value"] * two["ms_per_step"] / 1e3
    assert abs(f1 - 8) < 1e-6 and abs(f2 - 16) < 1e-6
    assert one["host_buffers"] is not None and one["host_buffers"]["downloaded_equals_device"]
    assert one["roofline"] is not None and one["roofline"]["kernel"].startswith("k_") and "clahe_kernel" in one["roofline"]
    assert one["outputs_identical_across_streams"] is True and two["outputs_identical_across_streams"] is True
    assert two["spot_check_vs_oracle"] is not None and two["spot_check_vs_oracle"]["ok"], two["spot_check_vs_oracle"]
    assert one["host_cpu_s_per_step"] > 0
    # round 5: the timed region on three scenes (every rank takes part), the rules the step ran under, the matcher against both peaks
    for d in (one, two):
        sc = d["scenes"]
        assert len(sc["per_scene"]) == 3 and len({s["seed0"] for s in sc["per_scene"]}) == 3
        lo, med, hi = sc["frames_per_s_min_median_max"]
        assert 0 < lo <= med <= hi and sc["per_scene"][0]["frames_per_s"] == d["value"]
        assert "GUARD_S" in d["config"]["rules"]["dehaze_S"] and ">= 4 good matches" in d["config"]["rules"]["overlap"]
    m = one["roofline"]["matcher"]["in_step"]
    assert m["form"] == 4 and 0 < m["frac_of_fp4_peak"] < m["frac_of_i8_peak"] < 1
    assert one["roofline"]["hbm_copy_measured"]["GBps"] > 1000 and one["roofline"]["hbm_copy_measured"]["lut_pass"]["GBps"] > 1000


def test_bench_under_torch_distributed_run():
    """The driver's N > 1 form: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (here two gloo ranks
    sharing the one GPU of the test box)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["UWIP_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--frames", "8",
           "--streams", "2", "--rows", "270", "--cols", "480", "--no-cpu-baseline", "--no-matcher-bench"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and abs(d["value"] * d["ms_per_step"] / 1e3 - 16) < 1e-6 and d["scaling"] == "weak"

#!/usr/bin/env python3
"""bench.py -- frames/s of the per-frame pipe on synthetic 1080p frames.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (bgdehaze -> histretch -> aclahe ->
videostrip-overlap) over one batch of synthetic 1920x1080 uchar3 frames that is
already resident in HBM.  Frames are independent units: with N > 1 every rank
runs the same per-GPU batch on its own GPU (weak scaling, no data-path
collective; torch.distributed is used only for the barrier and the max-over-
ranks time).  Rank 0 prints ONE JSON line with the whole-job frames/s, the HBM
roofline of the dominant kernel (timed live with HIP events on the launch
stream) and the CPU baseline (the oracle, timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--streams", type=int, default=4, help="independent sub-batches in flight per GPU (HIP streams + host threads)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=540)
    ap.add_argument("--cpu-sample-cols", type=int, default=960)
    return ap.parse_args()


def cpu_baseline(rows, cols, full_rows, full_cols):
    """The oracle (CPU restatement, kind = "port"), single thread, on a bounded sample:
    dehaze/histretch/aclahe on ONE frame of reduced size (their cost is linear in the
    pixel count, so seconds are scaled by the pixel ratio), plus the overlap stage on ONE
    full-size frame pair (it always works on the 640-wide resize).  Returns seconds per
    frame at the bench resolution and the breakdown."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import _oracle
    import dehaze_oracle as dz
    from uwimageproc_amd import aclahe, synth
    orc = _oracle.load()
    img = synth.uw_stream(0, 1, rows, cols)[0]
    scale = (full_rows * full_cols) / float(rows * cols)
    t0 = time.time()
    out = dz.to_u8(dz.adaptiveExp_map(dz.normalize_input(img), 15, guard_s=True))
    t1 = time.time()
    st, _ = orc.histretch(out, "RGB")
    v = orc.bgr_to_v(st)
    tab = orc.sweep(v)
    t2 = time.time()
    bs, cl = aclahe.select_parameters(tab)          # scipy, as the reference's ACLAHE.py does
    t3 = time.time()
    orc.hsv_replace_v(st, orc.clahe(v, float(cl), bs, bs))
    t4 = time.time()
    pair = synth.uw_stream(0, 2, full_rows, full_cols)
    t5 = time.time()
    gray = orc.resize_gray(pair[1])
    orc.detect_describe(gray)                        # the key frame's features are cached (kframe->new_img)
    t6 = time.time()
    orc.calcOverlap(pair[0], pair[1], full_cols, full_rows)
    t7 = time.time()
    overlap = (t7 - t6) - (t6 - t5)                  # one detect+describe, one match, one homography per frame
    parts = {"dehaze_s": (t1 - t0) * scale, "histretch_sweep_s": (t2 - t1) * scale, "select_s": t3 - t2,
             "clahe_hsv_s": (t4 - t3) * scale, "overlap_s": overlap}
    return sum(parts.values()), parts


def pmc_traffic(kernel, rows, cols, frames_per_launch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes; tools/prof_summary.py writes profiles/pmc_traffic.json).  The counters are
    collected offline in their own rocprofv3 runs, so this is looked up, not measured in this process;
    None when no matching profile is committed."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        e = d.get(f"{cols}x{rows}", {}).get(kernel)
        return None if e is None else e["hbm_bytes_per_frame"] * frames_per_launch
    except Exception:
        return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU fallback)"
    # one process per GPU; UWIP_BENCH_BACKEND=gloo lets several ranks share one GPU (plumbing smoke test only)
    backend = os.environ.get("UWIP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from uwimageproc_amd import synth
    from uwimageproc_amd.pipeline import FramePipe

    F, H, W = args.frames, args.rows, args.cols
    S = max(1, args.streams)
    assert F % S == 0, "--frames must be divisible by --streams"
    Fs = F // S
    # a few distinct synthetic frames, tiled to the batch (seed = 1234 + index)
    distinct = min(F, 8)
    base = synth.uw_stream(0, distinct, H, W, seed0=1234 + 1000 * rank)      # a different scene per rank, same size
    reps = (F + distinct - 1) // distinct
    src = torch.from_numpy(np.concatenate([base] * reps, axis=0)[:F]).to(dev)
    torch.cuda.synchronize()
    # S independent pipes, each on its own HIP stream and driven by its own host thread: frames are independent
    # units, so the HBM-bound dehaze passes of one half-batch overlap the LDS-bound sweep and the host-side
    # ACLAHE parameter choice of the other
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
    pipes = []
    for i in range(S):
        with torch.cuda.stream(streams[i]):
            pipes.append(FramePipe(dev_index, Fs, H, W))
    pipe = pipes[0]
    parts = [src[i * Fs:(i + 1) * Fs] for i in range(S)]

    def run_step():
        if S == 1:
            pipes[0].run(parts[0])
            return
        import threading
        def work(i):
            with torch.cuda.stream(streams[i]):
                pipes[i].run(parts[i])
        th = [threading.Thread(target=work, args=(i,)) for i in range(1, S)]
        for t in th:
            t.start()
        work(0)
        for t in th:
            t.join()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    for _ in range(args.warmup):
        run_step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    from uwimageproc_amd import sharding
    dt = sharding.max_over_ranks(dt)          # the slowest rank defines the step time

    # per-kernel timing pass (HIP events on the launch stream, inside libuwip)
    roof = None
    kernels = {}
    if rank == 0:
        pipe.ctx.prof_reset()
        pipe.ctx.prof_enable(True)
        nprof = max(1, min(args.steps, 3))
        for _ in range(nprof):
            pipe.run(parts[0])
        torch.cuda.synchronize()
        res = pipe.ctx.prof_results()
        pipe.ctx.prof_enable(False)
        kernels = {k: {"ms_per_step": ms / nprof, "launches_per_step": cnt / nprof} for k, (ms, cnt) in res.items()}
        N = H * W
        # the CLAHE kernel named by north_star: the final per-frame apply (read N + write N per frame)
        if "k_clahe_apply" in res:
            ms, cnt = res["k_clahe_apply"]
            per_launch_bytes = 2.0 * N * Fs * nprof / cnt
            avg_ms = ms / cnt
            achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_clahe_apply", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("k_clahe_apply", H, W, Fs * nprof / cnt), "avg_launch_ms": avg_ms,
                    "algorithmic_bytes_per_launch": per_launch_bytes}
        # for orientation: the kernels that actually dominate the step (the sweep is VALU-issue bound, the guided-filter
        # kernels HBM / VALU bound: DESIGN.md section 5, profiles/r01_sq_counters.txt)
        if roof is not None and kernels:
            tot = sum(v["ms_per_step"] for v in kernels.values())
            top = sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_step"])[:3]
            roof["step_kernel_ms"] = tot
            roof["largest_kernels"] = [{"kernel": k, "ms_per_step": v["ms_per_step"], "share": v["ms_per_step"] / tot} for k, v in top]

    if rank == 0:
        cpu = None
        if not args.no_cpu_baseline:
            r, c = min(args.cpu_sample_rows, H), min(args.cpu_sample_cols, W)
            secs, parts = cpu_baseline(r, c, H, W)
            cpu = {"value": 1.0 / secs, "unit": "frames/s", "cores": 1, "kind": "port",
                   "sample": f"1 frame: dehaze+histretch+aclahe through the CPU oracle at {c}x{r} (seconds scaled by "
                             f"pixel count to {W}x{H}; numpy dehaze, C histretch/CLAHE sweep, scipy parameter choice) "
                             f"+ overlap oracle (C) on one {W}x{H} frame pair",
                   "seconds_per_frame": secs, "parts": parts}
        total_frames = world * F * args.steps
        line = {
            "metric": "frames/sec whole-node, 1080p full pipe (dehaze+stretch+CLAHE+overlap)",
            "value": total_frames / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 (histretch/CLAHE), f64 (dehaze), f32+i8 (overlap)",
            "data": "synthetic",
            "config": {"workload": f"full pipe bgdehaze->histretch->aclahe->videostrip-overlap on {W}x{H} uchar3 frames",
                       "frames_per_gpu_per_step": F, "streams_per_gpu": S, "stages": pipe.stages(),
                       "parallelism": f"frame-batch x{world}"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "kernels": kernels,
        }
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

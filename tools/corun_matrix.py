"""Pairwise co-residency of the pipe's kernels on one MI355X.

Every "actor" is one library call whose time is (almost) one kernel, on a context / stream of its own, over a
64-frame 1080p sub-batch of the pipe's own data:
    sweep    uwip_aclahe_sweep               k_clahe_sweep  (VALU + LDS-atomic bound)
    solve2   uwip_dehaze, only the 1st kernel   k_gf_ws_solve<2 planes, 8-bit p>  (one wave per SIMD, latency bound)
    final2   uwip_dehaze, only the 2nd kernel   k_gf_ws_final<recover>            (HBM latency bound)
    gf1      uwip_guided_filter              the third filter's one-plane solve + final
    band     uwip_clahe_luts 32x32           k_clahe_band (LDS atomics)
    apply    uwip_clahe 8x8                  tile histograms + LUTs + k_clahe_apply (HBM)
    lut      uwip_histretch RGB              k_hist_u8 + k_apply_lut (HBM)
    detect   uwip_overlap_detect             the ~40 cache-resident scale-space / detector kernels
(UWIP_DIAG_GF_ONLY, read by uwip_gf_wave_strip at every call, is what isolates the two guided-filter kernels; the small
kernels around them in uwip_dehaze -- window filter, background light, output pass -- still run and are timed apart.)
For a pair (A, B): nA, nB calls of each, sized to ~T ms per stream, submitted interleaved from one host thread;
    serial = nA tA + nB tB     ideal = max(nA tA, nB tB)     wall = measured
    gain   = serial / wall  (1.0 = the two merely take turns; serial / ideal = perfect co-residency)
and per kernel the stretch = its event-bracketed time in the pair / alone (uwip_prof_*).

    python3 tools/corun_matrix.py [--frames 64] [--ms 60] [--actors sweep,solve2,...] [--json out.json]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["UWIP_TEST_HOOKS"] = "1"              # UWIP_DIAG_GF_ONLY is a measurement hook: dead in a product process
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from uwimageproc_amd import synth
from uwimageproc_amd._native import Context, batch_of

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=64)
ap.add_argument("--ms", type=float, default=60.0)
ap.add_argument("--actors", default="sweep,solve2,final2,gf1,band,apply,lut,detect")
ap.add_argument("--json", default=None)
ap.add_argument("--rows", type=int, default=1080)
ap.add_argument("--cols", type=int, default=1920)
args = ap.parse_args()
F, H, W = args.frames, args.rows, args.cols
dev = torch.device("cuda", 0)

# the pipe's own data: raw frames for the dehaze actors, dehazed + stretched frames and their blurred V for the rest
base = synth.uw_stream(0, min(F, 16), H, W)
raw = torch.from_numpy(np.concatenate([base] * ((F + len(base) - 1) // len(base)))[:F]).to(dev)
c0 = Context(0)
work = torch.empty_like(raw)
rb, wb = batch_of(raw), batch_of(work)
c0.call("uwip_dehaze_histretch", C.byref(rb), C.byref(wb), 15, 3, b"RGB", 2, 98, 0)
v = torch.empty((F, H, W), dtype=torch.uint8, device=dev)
vblur = torch.empty_like(v)
vb, vbb = batch_of(v), batch_of(vblur)
c0.call("uwip_bgr_to_v", C.byref(wb), C.byref(vb))
c0.call("uwip_GaussianBlur3", C.byref(vb), C.byref(vbb), 0)
c0.sync()
c0.close()
torch.cuda.synchronize()


def prof_results(ctx):
    n = C.c_int(0)
    ctx.call("uwip_prof_count", C.byref(n))
    out = {}
    for i in range(n.value):
        name = C.create_string_buffer(128)
        ms, cnt = C.c_double(0), C.c_uint64(0)
        ctx.call("uwip_prof_get", i, name, 128, C.byref(ms), C.byref(cnt))
        if cnt.value:
            out[name.value.decode()] = (ms.value, cnt.value)
    return out


class Actor:
    def __init__(self, kind):
        self.kind = kind
        self.ctx = Context(0)                     # a stream of its own
        self.env = None
        self.key = None
        if kind == "sweep":
            self.tab = torch.empty((F, 5, 51), dtype=torch.float32, device=dev)
            self.fn = lambda: self.ctx.call("uwip_aclahe_sweep", C.byref(vbb), 0, C.c_void_p(self.tab.data_ptr()))
            self.key = "k_clahe_sweep"
        elif kind in ("solve2", "final2"):
            self.out = torch.empty_like(raw)
            self.ob = batch_of(self.out)
            self.fn = lambda: self.ctx.call("uwip_dehaze", C.byref(rb), C.byref(self.ob), 15, 0, None, None, None)
            self.env = "solve" if kind == "solve2" else "final"
            self.key = "k_gf_ws_solve" if kind == "solve2" else "k_gf_ws_final"
        elif kind == "gf1":
            self.p = torch.rand((F, H, W), dtype=torch.float64, device=dev)
            self.q = torch.empty_like(self.p)
            self.fn = lambda: self.ctx.call("uwip_guided_filter", C.byref(rb), C.c_void_p(self.p.data_ptr()), 40,
                                            C.c_double(1e-3), C.c_void_p(self.q.data_ptr()))
            self.key = "k_gf_ws_solve"
        elif kind == "band":
            self.luts = torch.empty((F, 1024, 256), dtype=torch.uint8, device=dev)
            self.fn = lambda: self.ctx.call("uwip_clahe_luts", C.byref(vb), C.c_double(3.0), 32, 32, 0,
                                            C.c_void_p(self.luts.data_ptr()))
            self.key = "k_clahe_band"
        elif kind == "apply":
            self.o = torch.empty_like(v)
            self.obb = batch_of(self.o)
            self.fn = lambda: self.ctx.call("uwip_clahe", C.byref(vb), C.byref(self.obb), C.c_double(3.0), 8, 8, 0)
            self.key = "k_clahe_apply"
        elif kind == "lut":
            self.w = work.clone()
            self.wbb = batch_of(self.w)
            self.fn = lambda: self.ctx.call("uwip_histretch", C.byref(self.wbb), b"RGB", 2, 98)
            self.key = "k_apply_lut"
        elif kind == "detect":
            fh = C.c_void_p()
            self.ctx.call("uwip_features_create", F, C.byref(fh))
            self.feats = fh
            self.fn = lambda: self.ctx.call("uwip_overlap_detect", C.byref(wb), self.feats, 0)
            self.key = None
        else:
            raise SystemExit(f"unknown actor {kind}")
        # a complete first call: workspaces allocated, the guided filter's intermediate planes valid
        os.environ.pop("UWIP_DIAG_GF_ONLY", None)
        self.fn()
        self.ctx.sync()

    def submit(self):
        if self.env:
            os.environ["UWIP_DIAG_GF_ONLY"] = self.env
        else:
            os.environ.pop("UWIP_DIAG_GF_ONLY", None)
        self.fn()

    def close(self):
        if hasattr(self, "feats"):
            self.ctx._l.uwip_features_destroy(self.feats)
        self.ctx.close()


def run(actors, counts):
    """interleaved submission of counts[i] calls of actors[i]; returns (wall ms, [prof per actor])"""
    for a in actors:
        a.ctx.call("uwip_prof_reset")
        a.ctx.call("uwip_prof_enable", 1)
    torch.cuda.synchronize()
    done = [0] * len(actors)
    tot = max(counts)
    t0 = time.perf_counter()
    for i in range(tot):
        for k, a in enumerate(actors):
            want = (i + 1) * counts[k] // tot
            while done[k] < want:
                a.submit()
                done[k] += 1
    for a in actors:
        a.ctx.sync()
    wall = (time.perf_counter() - t0) * 1e3
    res = []
    for a in actors:
        res.append(prof_results(a.ctx))
        a.ctx.call("uwip_prof_enable", 0)
    os.environ.pop("UWIP_DIAG_GF_ONLY", None)
    return wall, res


kinds = args.actors.split(",")
alone = {}
print(f"# co-run matrix: {F} frames of {W}x{H}, ~{args.ms:.0f} ms per stream, GPU_MAX_HW_QUEUES={os.environ['GPU_MAX_HW_QUEUES']}")
print("# alone: ms per call (wall), then the call's kernels (ms per call)")
for kind in kinds:
    a = Actor(kind)
    run([a], [2])
    w1, _ = run([a], [3])
    n = max(2, int(round(args.ms / (w1 / 3))))
    wall, (pr,) = run([a], [n])
    alone[kind] = dict(ms=wall / n, n=n, kernels={k: ms / n for k, (ms, c) in pr.items()}, key=a.key)
    ks = ", ".join(f"{k} {ms:.3f}" for k, ms in sorted(alone[kind]["kernels"].items(), key=lambda kv: -kv[1])[:5])
    print(f"{kind:8s} {wall / n:8.3f} ms   [{ks}]")
    a.close()

rows = []
print("# pairs: A B | nA nB | serial ideal wall (ms) | gain = serial / wall (perfect = serial / ideal) | stretch of A's, B's kernel")
for i, ka in enumerate(kinds):
    for kb in kinds[i:]:
        A, B = Actor(ka), Actor(kb)
        nA, nB = alone[ka]["n"], alone[kb]["n"]
        run([A, B], [2, 2])
        wall, (pa, pb) = run([A, B], [nA, nB])
        serial = nA * alone[ka]["ms"] + nB * alone[kb]["ms"]
        ideal = max(nA * alone[ka]["ms"], nB * alone[kb]["ms"])

        def stretch(kind, pr, n):
            key = alone[kind]["key"]
            if key is None or key not in pr:
                tot_a = sum(alone[kind]["kernels"].values())
                return (sum(ms for ms, c in pr.values()) / n) / tot_a if tot_a > 0 else float("nan")
            return (pr[key][0] / n) / alone[kind]["kernels"][key]
        sa, sb = stretch(ka, pa, nA), stretch(kb, pb, nB)
        rows.append(dict(a=ka, b=kb, nA=nA, nB=nB, serial_ms=serial, ideal_ms=ideal, wall_ms=wall, gain=serial / wall,
                         perfect=serial / ideal, stretch_a=sa, stretch_b=sb))
        print(f"{ka:8s} {kb:8s} | {nA:3d} {nB:3d} | {serial:7.2f} {ideal:7.2f} {wall:7.2f} | {serial / wall:5.2f} ({serial / ideal:4.2f}) | {sa:5.2f} {sb:5.2f}")
        A.close(); B.close()

if args.json:
    with open(args.json, "w") as f:
        json.dump(dict(frames=F, rows=H, cols=W, alone=alone, pairs=rows, env={k: v for k, v in os.environ.items() if k.startswith("UWIP_") or k == "GPU_MAX_HW_QUEUES"}), f, indent=1)

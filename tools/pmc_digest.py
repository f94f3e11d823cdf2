"""Digest the outputs of tools/profile_round.sh <tag> (gpurun_out/) into the summaries committed under profiles/:
  profiles/<tag>_fullpipe_1080p_kernel_stats.csv   rocprofv3 --kernel-trace --stats (per kernel: calls, total, average)
  profiles/<tag>_pmc.json                          per kernel, per FRAME: HBM bytes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE,
                                                   MI355X_MICROARCH.md "HBM"), VALU / LDS wave-instructions, LDS cycles
  profiles/<tag>_counters.txt                      the raw per-launch counter averages
usage: python tools/pmc_digest.py <tag> [--frames 64] [--size 1920x1080]"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0].split("<")[0]



def tree_meta():
    """what tree the counters were collected from: the digest runs here, right after gpurun has merged the CSVs of the tree it
    snapshotted, so git HEAD + a dirty flag over the product sources say which tree that was (bench.py replays the figures of
    this file as `traffic` and quotes this record next to them)"""
    import subprocess, time
    meta = {"collected": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}
    try:
        meta["git_head"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
        meta["dirty"] = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "uwimageproc_amd", "bench.py", "include"], text=True).strip())
    except Exception:
        meta["git_head"], meta["dirty"] = None, None
    return meta

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--size", default="1920x1080")
    a = ap.parse_args()
    g = os.path.join(ROOT, "gpurun_out")
    out = os.path.join(ROOT, "profiles")
    # kernel stats
    def newest(pattern):          # a directory may hold the CSVs of an earlier run of the same tag: digest the latest only
        fs = glob.glob(pattern, recursive=True)
        return [max(fs, key=os.path.getmtime)] if fs else []

    rows = collections.OrderedDict()
    for f in newest(os.path.join(g, f"prof_{a.tag}", "**", "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            e = rows.setdefault(k, [0, 0.0])
            e[0] += int(r["Calls"]); e[1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in rows.values()) or 1.0
    with open(os.path.join(out, f"{a.tag}_fullpipe_1080p_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"])
        for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, c, f"{t/1e6:.3f}", f"{t/c/1e3:.2f}", f"{100*t/tot:.2f}"])
    # counters: average per launch, per kernel
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    pmc_files = []
    for d in glob.glob(os.path.join(g, f"pmc_{a.tag}_*")):
        if os.path.isdir(d):
            pmc_files += newest(os.path.join(d, "**", "*counter_collection.csv"))
    for f in pmc_files:
        for r in csv.DictReader(open(f)):
            c = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"]); c[1] += 1
    with open(os.path.join(out, f"{a.tag}_counters.txt"), "w") as f:
        f.write("# rocprofv3 --pmc passes of: bench.py --steps 3 --warmup 1 --streams 1 --frames 64 (per-launch averages)\n"
                "# SQ_* cycle counters count quad-cycles summed over the chip; FETCH_SIZE / WRITE_SIZE in KB (FETCH_SIZE x2 on gfx950)\n")
        for k in sorted(agg):
            if not k.startswith("k_"):
                continue
            f.write(k + "\n")
            for c in sorted(agg[k]):
                s, n = agg[k][c]
                f.write(f"    {c:28s} {s/n:16.1f}  (n={n})\n")
    path = os.path.join(out, f"{a.tag}_pmc.json")
    try:
        d = json.load(open(path))
    except Exception:
        d = {}
    ent = d.setdefault(a.size, {})
    for k, cs in agg.items():
        if not k.startswith("k_"):
            continue
        avg = {c: s / n for c, (s, n) in cs.items()}
        e = {"frames_per_launch": a.frames}
        if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
            e["hbm_bytes_per_frame"] = (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024 / a.frames
            e["fetch_KB_per_launch"], e["write_KB_per_launch"] = avg["FETCH_SIZE"], avg["WRITE_SIZE"]
            e["hbm_bytes_per_frame_fetch_as_reported"] = (avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024 / a.frames
            e["note"] = ("FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md HBM section).  Calibration in this code base: k_bgr_to_v reads "
                         "exactly 3 B/px and k_hsv_replace_v 4 B/px with dword / dwordx2 loads, and both report half of that; byte-wide "
                         "loads (the pixel reads of k_clahe_sweep) are not calibrated: hbm_bytes_per_frame_fetch_as_reported is the lower bound")
        for c, key in (("SQ_INSTS_VALU", "valu_insts_per_frame"), ("SQ_INSTS_LDS", "lds_insts_per_frame"), ("SQ_INSTS_SALU", "salu_insts_per_frame"),
                       ("SQ_LDS_IDX_ACTIVE", "lds_idx_active_per_frame"), ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_per_frame"),
                       ("SQ_ACTIVE_INST_VALU", "valu_active_quadcycles_per_frame"), ("SQ_WAVE_CYCLES", "wave_quadcycles_per_frame")):
            if c in avg:
                e[key] = avg[c] / a.frames
        ent[k] = e
    d["_meta"] = tree_meta()
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)
    for name in (f"bench_{a.tag}.json", f"bench4k_{a.tag}.json"):
        if os.path.exists(os.path.join(g, name)):
            lines = [l for l in open(os.path.join(g, name)).read().splitlines() if l.startswith("{")]
            if lines:
                json.dump(json.loads(lines[-1]), open(os.path.join(out, name.replace("bench", a.tag + "_bench_line").replace(f"_{a.tag}.json", ".json")), "w"), indent=1)
    print("wrote profiles/", a.tag)


if __name__ == "__main__":
    main()

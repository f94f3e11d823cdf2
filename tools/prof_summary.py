"""Digest rocprofv3 CSV output (kernel stats + optional PMC passes) into the compact
summaries committed under profiles/.
usage: python tools/prof_summary.py <stats_dir> [--fetch DIR] [--write DIR] [--out profiles/NAME]"""
import argparse
import collections
import csv
import glob
import os
import re


def short(name: str) -> str:
    n = name.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0]


def load_stats(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["Percentage"])))
    return rows


def load_pmc(d, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            agg[k][0] += float(r["Counter_Value"])
            agg[k][1] += 1
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--out", required=True)
    ap.add_argument("--frames", type=int, default=0, help="frames per launch of the profiled run (enables pmc_traffic.json)")
    ap.add_argument("--size", default="1920x1080")
    a = ap.parse_args()
    stats = load_stats(a.stats)
    fetch = load_pmc(a.fetch, "FETCH_SIZE") if a.fetch else {}
    write = load_pmc(a.write, "WRITE_SIZE") if a.write else {}
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out + "_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent", "fetch_KB_per_launch", "write_KB_per_launch",
                    "hbm_MB_per_launch_corrected"])
        for name, calls, tot, avg, pct in sorted(stats, key=lambda r: -r[2]):
            fk = fetch[name][0] / fetch[name][1] if name in fetch and fetch[name][1] else ""
            wk = write[name][0] / write[name][1] if name in write and write[name][1] else ""
            # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are KB; on gfx950 FETCH_SIZE reports half of the
            # bytes of a wide coalesced streaming read -> doubled before comparing with byte counts
            hb = (2.0 * fk + wk) * 1024 / 1e6 if fk != "" and wk != "" else ""
            w.writerow([name, calls, f"{tot/1e6:.3f}", f"{avg/1e3:.2f}", f"{pct:.2f}",
                        f"{fk:.1f}" if fk != "" else "", f"{wk:.1f}" if wk != "" else "", f"{hb:.2f}" if hb != "" else ""])
    print("wrote", a.out + "_kernel_stats.csv")
    if a.frames and fetch and write:
        import json
        path = os.path.join(os.path.dirname(a.out) or ".", "pmc_traffic.json")
        try:
            d = json.load(open(path))
        except Exception:
            d = {}
        ent = d.setdefault(a.size, {})
        for name in fetch:
            if name in write and fetch[name][1] and write[name][1]:
                fk, wk = fetch[name][0] / fetch[name][1], write[name][0] / write[name][1]
                ent[name.split("<")[0]] = {"hbm_bytes_per_frame": (2.0 * fk + wk) * 1024 / a.frames, "fetch_KB_per_launch": fk,
                             "write_KB_per_launch": wk, "frames_per_launch": a.frames,
                             "note": "FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section)"}
        json.dump(d, open(path, "w"), indent=1, sort_keys=True)
        print("wrote", path)


if __name__ == "__main__":
    main()

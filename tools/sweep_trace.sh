#!/bin/bash
# usage (GPU box, repo root): tools/sweep_trace.sh <tag>   -> gpurun_out/sweeptrace_<tag>.txt : per-dispatch durations of k_clahe_sweep
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/sweeptrace_$tag --output-format csv -- python3 tools/sweep_only.py 2 > gpurun_out/sweeptrace_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
rows = []
for f in glob.glob(f"gpurun_out/sweeptrace_{tag}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_clahe_sweep" in n or "k_clahe_lut" in n or "k_clahe_tilehist" in n:
            rows.append((int(r["Start_Timestamp"]), n.split("(")[0][-28:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                         r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""),
                         r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("LDS_Block_Size", "")))
rows.sort()
with open(f"gpurun_out/sweeptrace_{tag}.txt", "w") as o:
    for r in rows:
        o.write("%-30s %10.1f us grid=%s,%s,%s vgpr=%s agpr=%s lds=%s\n" % r[1:])
print(open(f"gpurun_out/sweeptrace_{tag}.txt").read())
PY

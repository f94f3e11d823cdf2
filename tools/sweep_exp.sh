#!/bin/bash
# usage (GPU box): tools/sweep_exp.sh "0 1 2 3 4"  -> per-variant per-grid sweep durations (UWIP_SWEEP_EXP diagnostic builds)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for e in $1; do
  export UWIP_SWEEP_EXP=$e
  timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/swexp_$e --output-format csv -- python3 tools/sweep_only.py 2 > gpurun_out/swexp_$e.log 2>&1 || exit 1
  python3 - "$e" <<'PY'
import csv, glob, sys
e = sys.argv[1]
rows = []
for f in glob.glob(f"gpurun_out/swexp_{e}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_clahe_sweep" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
d = [x[1] for x in rows][-5:]
print("variant", e, "grids 32,16,8,4,2 (us):", " ".join("%7.1f" % x for x in d), " sum %.1f" % sum(d), flush=True)
PY
done

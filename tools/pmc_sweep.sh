#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sweep.sh <tag>  -- SQ / LDS counters of the sweep-only run, per k_clahe_sweep dispatch
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set -d gpurun_out/pmcs_${tag}_$i --output-format csv -- python3 tools/sweep_only.py 1 > gpurun_out/pmcs_${tag}_$i.log 2>&1
done
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"gpurun_out/pmcs_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_clahe_sweep" not in r["Kernel_Name"]:
            continue
        agg[r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?")][r["Counter_Name"]] += float(r["Counter_Value"])
with open(f"gpurun_out/pmcs_{tag}.txt", "w") as o:
    for g, d in sorted(agg.items()):
        o.write(f"grid {g}\n")
        for k, v in sorted(d.items()):
            o.write(f"   {k:28s} {v:18.0f}\n")
print(open(f"gpurun_out/pmcs_{tag}.txt").read())
PY

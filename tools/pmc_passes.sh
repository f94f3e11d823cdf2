#!/bin/bash
# usage: tools/pmc_passes.sh <tag> [bench args...]   (run on the GPU box, from the repo root)
# Collects SQ / LDS / HBM counters in separate rocprofv3 --pmc passes under gpurun_out/pmc_<tag>_*.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 1 --warmup 0 --streams 1 --frames 64 --no-cpu-baseline $*"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 10 240 rocprofv3 --pmc $set -d gpurun_out/pmc_${tag}_$i --output-format csv -- $B > gpurun_out/pmc_${tag}_$i.log 2>&1
done

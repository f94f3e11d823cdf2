#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
#   1. the default bench line and the 4K bench line (with the paced 4K@60 stream)
#   2. rocprofv3 --kernel-trace --stats of a single-stream 64-frame run of the SAME pipe (the command bench.py's
#      per-kernel numbers come from: one sub-batch on one stream)
#   3. four separate --pmc passes of that command (SQ instruction / LDS counters, FETCH_SIZE, WRITE_SIZE)
# Outputs under gpurun_out/; digest locally with  python tools/pmc_digest.py <tag>
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "bench (default)"; timeout -k 10 500 python3 bench.py > gpurun_out/bench_${tag}.json 2> gpurun_out/bench_${tag}.err
echo "bench (4k-paced)"; timeout -k 10 500 python3 bench.py --config 4k-paced --no-cpu-baseline --no-matcher-bench > gpurun_out/bench4k_${tag}.json 2> gpurun_out/bench4k_${tag}.err
B="python3 bench.py --steps 3 --warmup 1 --streams 1 --frames 64 --no-cpu-baseline --no-host-buffers --no-matcher-bench --no-large-working-set --no-4k"
echo "kernel stats"; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag} --output-format csv -- $B > gpurun_out/prof_${tag}.log 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "pmc pass $i: $set"
  timeout -k 10 300 rocprofv3 --pmc $set -d gpurun_out/pmc_${tag}_$i --output-format csv -- $B > gpurun_out/pmc_${tag}_$i.log 2>&1
done
# 4. the matcher on BASELINE config 4's size (64 pairs x 2048 x 2048): MFMA counters in two separate --pmc passes of
#    tools/matcher_only.py (the program directly after --), digested by tools/pmc_matcher_digest.py
j=0
for set in "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  j=$((j+1))
  echo "matcher pmc pass $j: $set"
  timeout -k 10 240 rocprofv3 --pmc $set -d gpurun_out/pmcm_${tag}_$j --output-format csv -- python3 tools/matcher_only.py > gpurun_out/pmcm_${tag}_$j.log 2>&1 || true
done
timeout -k 10 120 python3 tools/matcher_only.py > gpurun_out/matcher_${tag}.json 2> /dev/null || true
UWIP_MATCH_FORM=3 timeout -k 10 120 python3 tools/matcher_only.py > gpurun_out/matcher_i8_${tag}.json 2> /dev/null || true
# 5. pairwise co-residency of the pipe's kernels (two streams), and the per-kernel time table of one sub-batch
echo "co-run matrix"; timeout -k 10 300 python3 tools/corun_matrix.py --json gpurun_out/corun_${tag}.json > gpurun_out/corun_${tag}.txt 2>&1 || true
timeout -k 10 120 python3 tools/kernel_times.py 64 3 > gpurun_out/kernel_times_${tag}.txt 2>&1 || true
echo done

#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# 1. the default bench line  2. rocprofv3 kernel stats of a 64-frame single-stream run  3./4. FETCH_SIZE / WRITE_SIZE passes
# Outputs under gpurun_out/; digest locally with
#   python tools/prof_summary.py gpurun_out/prof_<tag> --fetch gpurun_out/pmcf_<tag> --write gpurun_out/pmcw_<tag> \
#          --out profiles/<tag>_fullpipe_1080p_f64 --frames 64 --size 1920x1080
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "bench (default)"; timeout -k 10 500 python3 bench.py > gpurun_out/bench_${tag}.log 2>&1; tail -1 gpurun_out/bench_${tag}.log | cut -c1-400
B="python3 bench.py --steps 3 --warmup 1 --streams 1 --frames 64 --no-cpu-baseline"
echo "kernel stats"; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag} --output-format csv -- $B > gpurun_out/prof_${tag}.log 2>&1
echo "FETCH_SIZE"; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmcf_${tag} --output-format csv -- $B > gpurun_out/pmcf_${tag}.log 2>&1
echo "WRITE_SIZE"; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmcw_${tag} --output-format csv -- $B > gpurun_out/pmcw_${tag}.log 2>&1
echo done

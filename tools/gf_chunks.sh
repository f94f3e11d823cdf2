#!/bin/bash
# usage (GPU box, repo root): tools/gf_chunks.sh  -> k_gf_ws_solve / k_gf_ws_final ms per 64-frame step for UWIP_GF_CHUNKS = default, 1 .. 13
for c in default 1 2 3 4 5 6 7 8 10 13; do
  if [ $c = default ]; then unset UWIP_GF_CHUNKS; else export UWIP_GF_CHUNKS=$c; fi
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --streams 1 --frames 64 --no-cpu-baseline --no-host-buffers --no-matcher-bench --no-large-working-set 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']; print('chunks $c: solve %.3f  final %.3f  step %.1f ms' % (k['k_gf_ws_solve']['ms_per_subbatch'], k['k_gf_ws_final']['ms_per_subbatch'], d['ms_per_step']))" || exit 1
done

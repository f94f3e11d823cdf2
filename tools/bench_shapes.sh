# usage (GPU box, repo root): tools/bench_shapes.sh   -> frames/s for several (streams, frames per step) shapes of bench.py
# measured: 2416 (1 x 256), 2557 (2 x 128), 2602 (4 x 64, the default), 2605 (8 x 32), 2645 (4 x 128), 2666 (8 x 64)
for cfg in "4 256" "2 256" "8 256" "4 512" "2 512" "1 256" "8 512"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --streams $1 --frames $2 --no-cpu-baseline --no-host-buffers --no-matcher-bench --no-large-working-set 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('streams $1 frames $2:', round(d['value'],1), 'fps', round(d['ms_per_step'],1), 'ms')" || exit 1
done

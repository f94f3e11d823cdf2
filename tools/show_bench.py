import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{')][-1]
d = json.loads(l)
print({k: d[k] for k in ('value', 'ms_per_step', 'n_gpus')})
r = d.get('roofline')
if r: print('roofline', {k: r[k] for k in ('kernel', 'achieved', 'frac', 'avg_launch_ms')})
if d.get('cpu_baseline'): print('cpu', d['cpu_baseline']['value'])
tot = 0
for k, v in sorted(d['kernels'].items(), key=lambda kv: -kv[1]['ms_per_subbatch'])[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"  {k:22s} {v['ms_per_subbatch']:8.3f} ms")
print('sum kernels', sum(v['ms_per_subbatch'] for v in d['kernels'].values()))

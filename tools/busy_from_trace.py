"""GPU busy fraction and mean kernel concurrency of a rocprofv3 --kernel-trace directory over its densest window
(usage: python tools/busy_from_trace.py <dir> [from_fraction [to_fraction]] of the trace's span): union of the kernel
intervals / wall, sum of kernel time / wall."""
import csv, glob, sys
d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
upto = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
iv = []
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
iv.sort()
t0, t1 = iv[0][0], max(e for _, e in iv)
lo, hi = t0 + int((t1 - t0) * skip), t0 + int((t1 - t0) * upto)          # the steady part: skip set-up and the side passes
iv = [(max(s, lo), min(e, hi)) for s, e in iv if e > lo and s < hi]
wall = hi - lo
tot = sum(e - s for s, e in iv)
union, cs, ce = 0, None, None
for s, e in iv:
    if cs is None: cs, ce = s, e
    elif s <= ce: ce = max(ce, e)
    else: union += ce - cs; cs, ce = s, e
union += ce - cs
print(f"window {wall/1e6:.1f} ms: GPU busy (some kernel running) {100*union/wall:.1f} %, mean kernels in flight {tot/wall:.2f}, sum of kernel time {tot/1e6:.1f} ms")

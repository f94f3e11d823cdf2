"""Idle gaps of the GPU in a rocprofv3 --kernel-trace directory: over a window [from, to] of the trace's span, the time no
kernel is running, split by the kernel that ENDED before the gap and the one that STARTED after it.
usage: python tools/gaps_from_trace.py <dir> [from_fraction [to_fraction]]"""
import collections, csv, glob, sys
d = sys.argv[1]
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.65
iv = []
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
iv.sort()
t0, t1 = iv[0][0], max(e for _, e, _ in iv)
lo, hi = t0 + int((t1 - t0) * f0), t0 + int((t1 - t0) * f1)
before, after, sizes = collections.Counter(), collections.Counter(), []
ce, cn = None, None
for s, e, n in iv:
    if e <= lo or s >= hi: continue
    if ce is not None and s > ce:
        g = s - ce
        before[cn] += g; after[n] += g; sizes.append(g)
    if ce is None or e > ce: ce, cn = e, n
wall = hi - lo
tot = sum(sizes)
print(f"window {wall/1e6:.0f} ms: idle {tot/1e6:.0f} ms = {100*tot/wall:.1f} % in {len(sizes)} gaps (median {sorted(sizes)[len(sizes)//2]/1e3:.0f} us, "
      f"gaps > 100 us: {sum(g for g in sizes if g > 1e5)/1e6:.0f} ms, > 1 ms: {sum(g for g in sizes if g > 1e6)/1e6:.0f} ms)")
print("idle time by the kernel that ended before the gap:")
for n, g in before.most_common(8): print(f"  {n:28s} {g/1e6:8.1f} ms")
print("idle time by the kernel that started after the gap:")
for n, g in after.most_common(8): print(f"  {n:28s} {g/1e6:8.1f} ms")

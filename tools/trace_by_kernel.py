"""Per-kernel totals of a rocprofv3 --kernel-trace directory: usage  python tools/trace_by_kernel.py <dir> [reps]"""
import csv, glob, sys
d, reps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
acc = {}
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = acc.setdefault(n, [0, 0.0])
        a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in acc.values())
print(f"total {tot/reps:.1f} us per rep")
for n, (c, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n[:44]:44s} {c/reps:6.1f} launches {us/reps:9.1f} us  avg {us/c:7.1f} us")

"""Digest of a rocprofv3 --kernel-trace --memory-copy-trace run of tools/host_trace.py: where the copies sit against
the kernels.  usage: python tools/host_trace_digest.py <dir> [out.txt]"""
import csv, glob, sys
d = sys.argv[1]
kt = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
mt = glob.glob(f"{d}/**/*memory_copy_trace.csv", recursive=True)[0]
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(kt))]
M = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", ""), r["Stream_Id"]) for r in csv.DictReader(open(mt))]
t0 = min(min(k[0] for k in K), min(m[0] for m in M))
out = []
P = out.append
P(f"{len(K)} kernel dispatches, {len(M)} copies; t = ms since the first record")
P("copies:")
for s, e, di, st in sorted(M):
    P(f"  {di:16s} stream {st:>3s}  start {(s-t0)/1e6:9.3f}  end {(e-t0)/1e6:9.3f}  dur {(e-s)/1e6:8.3f} ms")
# union of kernel-busy time and of copy-busy time over the last 2/3 of the run
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
lo = sorted(m[0] for m in M)[len(M) // 3]
hi = max(k[1] for k in K)
kk = [(max(s, lo), min(e, hi)) for s, e, _, _ in K if e > lo]
P(f"window [{(lo-t0)/1e6:.1f}, {(hi-t0)/1e6:.1f}] ms: GPU busy with kernels {union(kk)/1e6:.1f} ms of {(hi-lo)/1e6:.1f} ms")
for di in ("HOST_TO_DEVICE", "DEVICE_TO_HOST"):
    mm = [(max(s, lo), min(e, hi)) for s, e, d2, _ in M if d2 == di and e > lo]
    if mm: P(f"  {di}: link busy {union(mm)/1e6:.1f} ms, sum of copy durations {sum(e-s for s,e in mm)/1e6:.1f} ms")
# gaps: intervals > 0.3 ms in which no kernel runs
iv = sorted((s, e) for s, e, _, _ in K)
ce = iv[0][1]; gaps = []
for s, e in iv[1:]:
    if s - ce > 300000: gaps.append((ce, s))
    ce = max(ce, e)
P("kernel-idle gaps > 0.3 ms:")
for a, b in gaps:
    cop = [f"{di[0]}2{di[-6]}@{st}" for s, e, di, st in M if s < b and e > a]
    P(f"  {(a-t0)/1e6:9.3f} .. {(b-t0)/1e6:9.3f}  ({(b-a)/1e6:6.2f} ms)  copies in flight: {' '.join(cop)}")
# blit kernels?
names = {}
for s, e, n, st in K: names[n] = names.get(n, 0) + (e - s)
P("kernels that look like copies: " + ", ".join(f"{n} {v/1e6:.2f} ms" for n, v in names.items() if any(w in n.lower() for w in ("copy", "blit", "fill"))))
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(txt + "\n")

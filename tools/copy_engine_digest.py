"""Which engine served each big copy of tools/ubench/copy_engine.bin: usage  python tools/copy_engine_digest.py <rocprofv3 dir>"""
import csv, glob, sys
d = sys.argv[1]
rec = []
for f in glob.glob(f"{d}/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e - s > 500000: rec.append((s, e, "SDMA " + r["Direction"].replace("MEMORY_COPY_", "")))
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "copyBuffer" in r["Kernel_Name"] and e - s > 500000: rec.append((s, e, f"BLIT kernel grid {r['Grid_Size_X']} wg {r['Workgroup_Size_X']}"))
rec.sort()
for i, (s, e, what) in enumerate(rec):
    print(f"{i:3d} t={(s-rec[0][0])/1e6:9.2f} ms dur {(e-s)/1e6:7.2f} ms  {what}")

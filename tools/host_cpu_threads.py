"""Where do a rank's host CPU seconds go?  Runs the bench's Rig for a few steps and prints the CPU time every OS thread
of the process used inside the timed region (/proc/self/task/*/stat), with the sub-batch threads identified by their
native ids.  usage: python tools/host_cpu_threads.py [steps] [--host]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv_saved = sys.argv[:]
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5
host = "--host" in sys.argv
sys.argv = ["bench.py"]
import bench
import torch

def snap():
    out = {}
    hz = os.sysconf("SC_CLK_TCK")
    for t in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{t}/stat").read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(t)] = (comm, (int(rest[11]) + int(rest[12])) / hz, int(rest[11]) / hz, int(rest[12]) / hz)
        except Exception:
            pass
    return out

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
rig = bench.Rig(0, dev, 512, 1080, 1920, 8, 1234)
names = {}
sub_cpu = {}
orig = rig.on_all
def on_all(fn):
    def wrapped(i):
        names[threading.get_native_id()] = f"sub-batch {i}"
        c = time.thread_time()
        fn(i)
        sub_cpu[i] = time.thread_time() - c          # this thread's own CPU clock (sub-batch threads exit with the call)
    orig(wrapped)
rig.on_all = on_all
if host:
    rig.host_prepare()
run = rig.run_steps_host if host else rig.run_steps
run(2); rig.drain()
a = snap(); t0 = time.perf_counter(); c0 = time.process_time()
run(steps); rig.drain()
wall = time.perf_counter() - t0; cpu = time.process_time() - c0
b = snap()
print(f"{steps} steps, wall {wall:.3f} s, process CPU {cpu:.3f} s = {cpu / wall:.2f} cores ({'host buffers' if host else 'resident'})")
rows = []
for t, (comm, tot, u, s) in b.items():
    p = a.get(t, (comm, 0, 0, 0))
    rows.append((tot - p[1], u - p[2], s - p[3], t, comm))
for d, u, s, t, comm in sorted(rows, reverse=True)[:24]:
    print(f"  tid {t:8d} {comm:16s} {names.get(t, ''):14s} cpu {d:7.3f} s (user {u:6.3f} sys {s:6.3f}) = {d / wall * 100:5.1f} % of a core")
print("sub-batch threads (thread_time over the timed call):", {k: round(v, 3) for k, v in sorted(sub_cpu.items())}, "sum", round(sum(sub_cpu.values()), 3))
print("threads alive:", len(b), " (exited sub-batch threads of the timed run are not listed: their time is in the process total)")
rig.close()

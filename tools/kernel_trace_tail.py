"""Timeline of the END of a rocprofv3 --kernel-trace run of bench.py (the timed window is the last thing the process launches):
   python tools/kernel_trace_tail.py <dir with *kernel_trace.csv> [number of trailing launches to list]
Prints the mean duration and the mean launch period of physics_kernel over blocks of the run, then the last launches one by one
(start relative to the first listed, duration, kernel) - is a slow window slow kernels, or gaps between them?"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
ph = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in csv.DictReader(open(f)))
phys = [p for p in ph if "physics_kernel<0" in p[2]]
print("physics_kernel<0> launches:", len(phys))
for name, blk in (("first 25", phys[:25]), ("launches 100..125", phys[100:125]), ("50..25 before the end", phys[-50:-25]), ("last 22", phys[-22:])):
    if len(blk) > 1:
        print(f"  {name:24s} mean duration {sum(e - s for s, e, _ in blk) / len(blk) / 1e3:7.1f} us   mean period {(blk[-1][0] - blk[0][0]) / (len(blk) - 1) / 1e3:7.1f} us")
t0 = ph[-n][0]
for s, e, k in ph[-n:]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {k}")

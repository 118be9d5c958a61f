"""Top rows of a rocprofv3 --kernel-trace --stats output directory:  python tools/kernel_stats_top.py gpurun_out/<dir>"""
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU kernel time ms", tot/1e6)
for r in rows[:24]:
    print("%5.1f%% calls %7s avg %8.1f us  %s" % (float(r["TotalDurationNs"])/tot*100, r["Calls"], float(r["AverageNs"])/1e3, r["Name"][:120]))

"""Summarise rocprofv3 --pmc CSV outputs: mean counter value per kernel launch, per kernel name.
   usage: python tools/pmc_summary.py <dir> [<dir> ...]"""
import sys, glob, csv, collections
acc = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (k, _, c), v in per.items():
            acc[(k.replace("(anonymous namespace)::", "").split("(")[0], c)].append(v)
for (k, c), v in sorted(acc.items()):
    print(f"{k[:52]:52s} {c:28s} launches={len(v):4d} mean={sum(v) / len(v):.6g}")

"""Developer tool: per-kernel averages of the counters in a rocprofv3 --pmc output directory (counter_collection csv)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, d in sorted(acc.items(), key=lambda kv: -len(n[kv[0]])):
    print(f"{len(n[k]):5d}  {k}")
    for c, v in sorted(d.items()): print(f"        {c:28s} {v / len(n[k]):16.1f} per launch")

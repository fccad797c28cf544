"""Developer tool: top kernels of a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total ms", round(tot / 1e6, 2))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%5.1f %5s %8.1f  %s" % (float(r["TotalDurationNs"]) / tot * 100, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:110]))

"""The library's own kernels out of a rocprofv3 --kernel-trace --stats run: python tools/stats_summary.py <output dir>
(name, calls, total ms, average ms, share of the GPU time of the whole process)."""
import csv
import glob
import os
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
print(f"{'kernel':44s} {'calls':>7s} {'total ms':>11s} {'avg ms':>10s} {'share':>7s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    name = r["Name"].split("(")[0].replace("void ", "")
    if not name.startswith("k_"):
        continue
    print(f"{name[:44]:44s} {int(r['Calls']):7d} {float(r['TotalDurationNs']) / 1e6:11.2f} {float(r['AverageNs']) / 1e6:10.3f} {float(r['TotalDurationNs']) / tot:7.1%}")
other = sum(float(r["TotalDurationNs"]) for r in rows if not r["Name"].split("(")[0].replace("void ", "").startswith("k_"))
print(f"{'(torch / rocprim / copies: data generation, plumbing)':44s} {'':7s} {other / 1e6:11.2f} {'':10s} {other / tot:7.1%}")

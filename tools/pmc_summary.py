"""Per-kernel sums of rocprofv3 PMC counters: python tools/pmc_summary.py <dir with *counter_collection.csv>."""
import csv
import glob
import os
import sys

per = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0][:60]
        d = per.setdefault(name, {"dispatches": set()})
        d["dispatches"].add(row["Dispatch_Id"])
        d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for name, d in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    n = len(d.pop("dispatches"))
    print(f"{name}  dispatches={n}  " + "  ".join(f"{k}={v:.4e}" for k, v in sorted(d.items())))

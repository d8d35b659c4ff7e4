"""Summarise rocprofv3 --pmc passes: per-kernel average of every counter (k_step / k_rollout rows only)."""
import csv
import glob
import sys
from collections import defaultdict

prefix = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in sorted(glob.glob(prefix + "*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "k_step" not in name and "k_rollout" not in name:
            continue
        short = "k_step" if "k_step" in name else "k_rollout"
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(prefix + "*/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "k_step" in name or "k_rollout" in name:
            dur["k_step" if "k_step" in name else "k_rollout"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in acc:
    d = dur.get(k, [])
    print(f"== {k}: dispatches/pass ~{len(d) // max(1, len(glob.glob(prefix + '*/')))}  avg duration under PMC {sum(d) / max(1, len(d)) / 1e3:.2f} us")
    for c, v in sorted(acc[k].items()):
        print(f"  {c:28s} avg/dispatch {sum(v) / len(v):16.1f}   (n={len(v)})")

"""Per-dispatch durations of the rollout kernel from a rocprofv3 --kernel-trace CSV directory: python tools/tsweep_parse.py <dir> [substr]"""
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_rollout"
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r))
rows.sort()
if rows:
    m = rows[0][2]
    print("kernel", m["Kernel_Name"][:100])
    print({k: m[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X")})
prev_end = None
out = []
for s, e, r in rows:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    out.append("%.1f(g%.1f)" % ((e - s) / 1e3, gap))
    prev_end = e
print("dispatch us (gap to previous end): " + " ".join(out))

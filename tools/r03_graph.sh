#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r03; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/graph_probe.py 20 > $O/graph_probe20.txt 2>&1
timeout -k 10 200 python tools/graph_probe.py 150 > $O/graph_probe150.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/graph_trace -- python3 $GRAFT_REPO_ROOT/tools/graph_probe.py 150 > $O/graph_probe150_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -v amdgpu $O/graph_probe20.txt; grep -v amdgpu $O/graph_probe150.txt
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r03/graph_trace/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_rollout_pc" in r["Kernel_Name"] or "k_refresh" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        nm = "k_refresh" if "k_refresh" in r["Kernel_Name"] else "k_rollout_pc"
        print(nm, "start %.1f us" % ((int(r["Start_Timestamp"]) - t0) / 1e3), "dur %.1f us" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3), "queue", r.get("Queue_Id"), "stream", r.get("Stream_Id"))
PY

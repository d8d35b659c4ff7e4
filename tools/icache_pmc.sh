#!/bin/bash
# instruction-cache counters of fused-rollout launches of 1 .. 400 steps (is the head of a launch instruction-fetch bound?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
grep -o -E "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_CACHE)[A-Z0-9_]*)\b" $O/counters_avail.txt | sort -u > $O/icache_names.txt
cat $O/icache_names.txt | tr '\n' ' '; echo
i=0
for ctrs in "$(head -4 $O/icache_names.txt | tr '\n' ' ')" "$(sed -n 5,8p $O/icache_names.txt | tr '\n' ' ')" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_VALU"; do
  [ -z "$(echo $ctrs)" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/ic_$i -- python3 $R/tools/icache_run.py > $O/ic_$i.log 2>&1 || { echo "pass $i ($ctrs) failed"; tail -3 $O/ic_$i.log; }
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r03p/ic_*/")):
    dur = {}
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rollout" in r["Kernel_Name"]:
                dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ids = sorted(dur)
    rows = collections.defaultdict(dict)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rollout" in r["Kernel_Name"]:
                rows[r["Counter_Name"]][int(r["Dispatch_Id"])] = rows[r["Counter_Name"]].get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    print(d, "rollout dispatches:", len(ids), "durations us:", " ".join("%.1f" % dur[i] for i in ids))
    for c, v in sorted(rows.items()):
        print("   %-36s %s" % (c, " ".join("%12.0f" % v.get(i, float('nan')) for i in ids)))
PY
rm -rf $O/ic_*/

#!/bin/bash
# usage: PMC="CTR1 CTR2 ..." tools/prof_pass.sh <tag> <bench args...>  -- one rocprofv3 PMC pass; every rollout dispatch listed with its duration
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $R/gpurun_out/pass_${tag} -- python3 $R/bench.py --no-cpu-baseline --no-also "$@" > $R/gpurun_out/pass_${tag}.log 2>&1 || echo "pass failed"
cd $R && python3 - <<PY
import csv, glob, collections
dur = {}
for f in glob.glob("gpurun_out/pass_${tag}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for f in sorted(glob.glob("gpurun_out/pass_${tag}/**/*counter_collection.csv", recursive=True)):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "rollout" not in r["Kernel_Name"]: continue
        rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for d, c in rows.items():
        print("${tag}", "dispatch", d, "us %.1f" % dur.get(d, -1), {k: "%.4e" % v for k, v in c.items()})
PY
rm -rf gpurun_out/pass_${tag}/

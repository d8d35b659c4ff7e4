#!/bin/bash
# usage: tools/prof_tlb.sh <tag> <bench args...>  -- UTCL1 (per-CU TLB) counters of the rollout kernels, one pass
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/tlb_${tag} -- python3 $R/bench.py --no-cpu-baseline --no-also "$@" > $R/gpurun_out/tlb_${tag}.log 2>&1 || echo "pass failed"
cd $R && python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/tlb_${tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "rollout" not in k: continue
    print("${tag}", k, {c: "%.3e" % (v / cnt[(k, c)]) for c, v in d.items()}, "launches", max(cnt[(k, c)] for c in d))
PY

#!/bin/bash
# usage: tools/prof_steady.sh <tag> <bench args...>  -- memory-pipeline counters of every rollout dispatch, listed one by one
# (run with --warmup 1600 --steps 200: the first dispatch is the warm-up from reset, the last the steady-state one)
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum" \
            "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
            "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 90 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/steady_${tag}_$i -- python3 $R/bench.py --no-cpu-baseline --no-also "$@" > $R/gpurun_out/steady_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
cd $R && python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/steady_${tag}_*/**/*counter_collection.csv", recursive=True)):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "rollout" not in r["Kernel_Name"]: continue
        rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for d, c in rows.items():
        print("${tag}", "dispatch", d, {k: "%.4e" % v for k, v in c.items()})
PY
rm -rf gpurun_out/steady_${tag}_*/

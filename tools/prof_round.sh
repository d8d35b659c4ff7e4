#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof_round.sh <tag> [bench args]     default bench args = the driver's: --gpus 1 --steps 20 --warmup 5
# 1. kernel trace + stats of that bench command   2. HBM traffic counters (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
tag=$1; shift
ARGS=${@:-"--gpus 1 --steps 20 --warmup 5"}
R=$GRAFT_REPO_ROOT
rm -rf "$R"/gpurun_out/prof_${tag}_trace "$R"/gpurun_out/prof_${tag}_FETCH_SIZE "$R"/gpurun_out/prof_${tag}_WRITE_SIZE      # (gpurun merges: stale runs would be counted twice)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_trace -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-boundary-leg > $R/gpurun_out/prof_${tag}_trace.log 2>&1 || echo "trace pass failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/prof_${tag}_$c -- python3 $R/bench.py $ARGS --no-cpu-baseline --no-boundary-leg --launch eager > $R/gpurun_out/prof_${tag}_$c.log 2>&1 || echo "$c pass failed"
done
cd $R && python3 tools/prof_summary.py $tag "$ARGS"

#!/bin/bash
# the early producer step (round 3): parity of the rollout kernels, then launch duration vs T against the previous build (rl_ptg_amd/lib/exp/libptg_env_head.so)
O=gpurun_out/r03p; mkdir -p $O
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; echo "rc=$rc $*" >> $O/status
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; cat $O/status; exit 1; fi; }
: > $O/status
run 900 $O/tests.log python -m pytest tests/test_rollout_parity.py tests/test_batch_edges.py tests/test_hip_parity.py tests/test_split_layout.py -m gpu -q -x
tail -3 $O/tests.log
grep -q "rc=0" $O/status || { cat $O/status; tail -40 $O/tests.log; exit 1; }
HEADLIB=$PWD/rl_ptg_amd/lib/exp/libptg_env_head.so
for round in 1 2; do
  for lib in new head; do
    if [ $lib = head ]; then export PTG_LIB_PATH=$HEADLIB; else unset PTG_LIB_PATH; fi
    run 200 $O/ab_${lib}_row32_$round.txt python tools/prologue_ab.py 65536
    TS_DTYPE=float64 run 200 $O/ab_${lib}_row64_$round.txt python tools/prologue_ab.py 65536
  done
done
unset PTG_LIB_PATH
for lib in new head; do
  if [ $lib = head ]; then export PTG_LIB_PATH=$HEADLIB; else unset PTG_LIB_PATH; fi
  TS_LAYOUT=feature run 200 $O/ab_${lib}_fm32.txt python tools/prologue_ab.py 65536
  TS_LAYOUT=split run 200 $O/ab_${lib}_split32.txt python tools/prologue_ab.py 65536
  run 200 $O/ab_${lib}_row32_4096.txt python tools/prologue_ab.py 4096
  run 200 $O/ab_${lib}_row32_262144.txt python tools/prologue_ab.py 262144
done
cat $O/status; cat $O/ab_*.txt

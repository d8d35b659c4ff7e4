#!/bin/bash
# usage: r03_ab_quick.sh <variant lib names...>: launch duration vs T (tools/prologue_ab.py) for experiment builds, row-major float32 and float64
O=gpurun_out/r03p; mkdir -p $O
for round in 1 2; do
for v in "$@"; do
  if [ $v = product ]; then unset PTG_LIB_PATH; else export PTG_LIB_PATH=$PWD/rl_ptg_amd/lib/exp/libptg_env_$v.so; fi
  timeout -k 10 200 python tools/prologue_ab.py 65536 2>/dev/null | tee $O/abq_${v}_row32_$round.txt || exit 1
  [ $round = 1 ] && { TS_DTYPE=float64 timeout -k 10 200 python tools/prologue_ab.py 65536 2>/dev/null | tee $O/abq_${v}_row64_$round.txt || exit 1; }
done
done

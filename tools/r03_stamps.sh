#!/bin/bash
# in-kernel phase stamps of one 20-step launch for the given -DPTG_STAMPS builds (single translation unit: g_stamps is per TU)
O=gpurun_out/r03p; mkdir -p $O
for v in "$@"; do
  PTG_LIB_PATH=$PWD/rl_ptg_amd/lib/exp/libptg_env_$v.so timeout -k 10 120 python tools/stamps.py 20 > $O/stamps_$v.txt 2>&1 || { echo "FAILED $v"; tail -5 $O/stamps_$v.txt; exit 1; }
  echo "== $v"; grep -A4 "rep 3" $O/stamps_$v.txt | grep "since own"
done

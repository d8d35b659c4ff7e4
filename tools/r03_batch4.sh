#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; echo "rc=$rc $*" >> $O/batch4.status
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; cat $O/batch4.status; exit 1; fi; }
: > $O/batch4.status
run 900 $O/tests4.log python -m pytest tests -m gpu -q
run 300 $O/vecenv2.txt python tools/bench_vecenv.py
cat $O/batch4.status; tail -15 $O/tests4.log; grep -v amdgpu $O/vecenv2.txt

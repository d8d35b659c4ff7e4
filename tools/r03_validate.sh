#!/bin/bash
# full validation of the committed code on the GPU box: every -m gpu test in ONE process, smoke(), the driver's bench command, the default bench
O=gpurun_out/r03v; mkdir -p $O
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; echo "rc=$rc $*" >> $O/status
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; cat $O/status; exit 1; fi; }
: > $O/status
run 1000 $O/tests.log python -m pytest tests -m gpu -q -x
run 200 $O/smoke.log python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')"
PTG_BENCH_DEBUG=1 run 300 $O/bench20.json python bench.py --gpus 1 --steps 20 --warmup 5
cat $O/status; tail -3 $O/tests.log; tail -2 $O/smoke.log; python tools/bench_line.py bench20 $O/bench20.json | cut -c1-400

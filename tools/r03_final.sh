#!/bin/bash
# final artefacts of round 3 on the committed kernels: full GPU suite, smoke, the driver's bench command (twice) and the default bench run, then the rocprofv3 round
O=gpurun_out/r03f; mkdir -p $O
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; echo "rc=$rc $*" >> $O/status
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; cat $O/status; exit 1; fi; }
: > $O/status
run 1000 $O/tests.log python -m pytest tests -m gpu -q -x
run 200 $O/smoke.log python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')"
PTG_BENCH_DEBUG=1 run 300 $O/bench20_a.json python bench.py --gpus 1 --steps 20 --warmup 5
PTG_BENCH_DEBUG=1 run 300 $O/bench20_b.json python bench.py --gpus 1 --steps 20 --warmup 5
run 400 $O/bench_default.json python bench.py
cat $O/status; tail -3 $O/tests.log; tail -2 $O/smoke.log
for f in bench20_a bench20_b bench_default; do python tools/bench_line.py $f $O/$f.json | cut -c1-330; done
bash tools/prof_round.sh r03 > $O/prof_round.log 2>&1; tail -12 $O/prof_round.log

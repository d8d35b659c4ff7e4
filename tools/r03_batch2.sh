#!/bin/bash
# round-3 batch 2: new tests, persistent-step probe (sc1 protocol), bench with union timing, 2-rank gloo rehearsal, rocprof summary
O=gpurun_out/r03; mkdir -p $O
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; echo "rc=$rc $*" >> $O/batch2.status
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; cat $O/batch2.status; exit 1; fi; }
: > $O/batch2.status
run 300 $O/tests2.log python -m pytest tests/test_batch_edges.py tests/test_cabi.py -m gpu -x -q -k "captured or graph or cabi or handles"
run 120 $O/persist2.txt tools/bin/persist_probe 2000
run 300 $O/bench20_b.json python bench.py --gpus 1 --steps 20 --warmup 5
PTG_BENCH_BACKEND=gloo run 300 $O/bench20_2rank_gloo.json python bench.py --gpus 2 --steps 20 --warmup 5 --no-also
PTG_BENCH_BACKEND=gloo PTG_BENCH_DEBUG=1 run 300 $O/bench20_2rank_gloo_b.json python bench.py --gpus 2 --steps 20 --warmup 5 --no-also --no-boundary-leg
PTG_BENCH_DEBUG=1 run 300 $O/bench20_1rank_noalso.json python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-boundary-leg --no-cpu-baseline
bash tools/prof_round.sh r03 > $O/prof_round.log 2>&1; echo "rc=$? prof_round" >> $O/batch2.status
cat $O/batch2.status; tail -5 $O/tests2.log

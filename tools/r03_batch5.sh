#!/bin/bash
O=gpurun_out/r03; mkdir -p $O
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; echo "rc=$rc $*" >> $O/batch5.status
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; cat $O/batch5.status; exit 1; fi; }
: > $O/batch5.status
run 600 $O/tests5.log python -m pytest tests/test_bench_path.py tests/test_batch_edges.py tests/test_vec_env_host.py -m gpu -q
PTG_BENCH_DEBUG=1 run 300 $O/bench20_c.json python bench.py --gpus 1 --steps 20 --warmup 5
PTG_BENCH_DEBUG=1 run 300 $O/bench20_d.json python bench.py --gpus 1 --steps 20 --warmup 5
PTG_BENCH_BACKEND=gloo PTG_BENCH_DEBUG=1 run 300 $O/bench20_2rank_c.json python bench.py --gpus 2 --steps 20 --warmup 5 --no-also
PTG_BENCH_BACKEND=gloo PTG_BENCH_DEBUG=1 run 300 $O/bench8k_2rank_c.json python bench.py --gpus 2 --envs 8192 --steps 20 --warmup 5 --no-also --no-boundary-leg
PTG_BENCH_DEBUG=1 run 300 $O/bench8k_1rank_c.json python bench.py --gpus 1 --envs 8192 --steps 20 --warmup 5 --no-also --no-boundary-leg --no-cpu-baseline
run 300 $O/bench_default_c.json python bench.py
cat $O/batch5.status; tail -5 $O/tests5.log
for f in bench20_c bench20_d bench20_2rank_c bench8k_1rank_c bench8k_2rank_c bench_default_c; do python tools/bench_line.py $f $O/$f.json | cut -c1-260; grep "timed region" $O/$f.json; done

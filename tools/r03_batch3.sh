#!/bin/bash
# round-3 batch 3: the whole GPU suite with the new kernels / host API, VecEnv rates, the bench at every BASELINE config, 2-rank rehearsals
O=gpurun_out/r03; mkdir -p $O
run() { local lim=$1 log=$2; shift 2; timeout -k 10 $lim "$@" > $log 2>&1; local rc=$?; echo "rc=$rc $*" >> $O/batch3.status
        if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: $*"; cat $O/batch3.status; exit 1; fi; }
: > $O/batch3.status
run 900 $O/tests3.log python -m pytest tests -m gpu -x -q
run 300 $O/vecenv.txt python tools/bench_vecenv.py
bash tools/r03_configs.sh > $O/configs.log 2>&1; echo "rc=$? configs" >> $O/batch3.status
run 200 $O/b1_8192.json python bench.py --gpus 1 --envs 8192 --steps 20 --warmup 5 --no-also --no-boundary-leg --no-cpu-baseline
PTG_BENCH_BACKEND=gloo run 200 $O/b2_8192.json python bench.py --gpus 2 --envs 8192 --steps 20 --warmup 5 --no-also --no-boundary-leg
run 200 $O/b1_65536.json python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-boundary-leg --no-cpu-baseline
PTG_BENCH_BACKEND=gloo run 200 $O/b2_65536.json python bench.py --gpus 2 --steps 20 --warmup 5 --no-also --no-boundary-leg
cat $O/batch3.status; tail -4 $O/tests3.log

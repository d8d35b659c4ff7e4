#!/bin/bash
# usage: tools/prof_pmc2.sh <tag> <bench args...>  -- two SQ passes only
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/bench.py --no-cpu-baseline --no-also "$@" > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_ > gpurun_out/pmc_${tag}_summary.txt 2>&1; cat gpurun_out/pmc_${tag}_summary.txt

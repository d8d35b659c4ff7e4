#!/bin/bash
# usage (GPU box): tools/phase_pmc.sh <tag>  -- per-dispatch counters of the early (just after reset) and the stationary phase
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" \
            "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
            "TCC_REQ_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/ph_${tag}_$i -- python3 $R/tools/phase_run.py > $R/gpurun_out/ph_${tag}_$i.log 2>&1 || echo "pass $i failed"
done
cd $R && python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/ph_${tag}_*/")):
    dur = {}
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rollout" in r["Kernel_Name"]:
                dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ids = sorted(dur)
    rows = collections.defaultdict(dict)
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rollout" in r["Kernel_Name"]:
                rows[r["Counter_Name"]][int(r["Dispatch_Id"])] = float(r["Counter_Value"])
    if len(ids) < 46:
        print(d, "dispatches", len(ids)); continue
    early, late = ids[1:4], ids[-6:-1]
    print(d, "duration us early %s  late %s" % (["%.1f" % dur[i] for i in early], ["%.1f" % dur[i] for i in late]))
    for c, v in sorted(rows.items()):
        e = sum(v[i] for i in early) / len(early); l = sum(v[i] for i in late) / len(late)
        print("   %-40s early %14.0f   late %14.0f   ratio %.3f" % (c, e, l, e / l if l else float("nan")))
PY
rm -rf gpurun_out/ph_${tag}_*/

"""Condense the rocprofv3 outputs of tools/prof_round.sh into profiles/<tag>_*.{csv,md,json} (the files that are committed)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

tag = sys.argv[1]
bench_args = sys.argv[2] if len(sys.argv) > 2 else "--gpus 1 --steps 20 --warmup 5"
K = int(re.search(r"--steps (\d+)", bench_args).group(1)) if "--steps" in bench_args else 2000
W = int(re.search(r"--warmup (\d+)", bench_args).group(1)) if "--warmup" in bench_args else 400
# bench.py runs the W warm-up steps of a rollout leg as TWO launches (W // 2 steps, then the rest right before the clock) when W >= 2:
# the timed launch is dispatch number NW of its kernel, the steady-state rollouts follow it
NW = 2 if W >= 2 else 1
os.makedirs("profiles", exist_ok=True)
ours = ("k_step", "k_rollout", "k_reset", "k_build", "k_fill", "k_init", "k_zero", "k_extract", "k_refresh", "k_vn")
LAY = {"0": "row", "1": "feature", "2": "sb3_flat", "3": "split"}


def short(name):
    m = re.search(r"(k_rollout_pc|k_step_hot)<(\d)[^>]*?(float|double)(, (?:true|false))?>", name)
    if m:
        tag = LAY[m.group(2)] + ("" if m.group(3) == "float" else ",f64") + (",info" if m.group(4) == ", true" else "")
        return f"{m.group(1)}<{tag}>"
    for k in ("k_refresh", "k_extract_keys", "k_step", "k_reset", "k_build_records", "k_build_argmin", "k_build_fast", "k_fill_noise", "k_init_state", "k_zero_noise_count"):
        if k in name:
            return k
    return name[:60]


def newest(pattern):
    """files of the most recent profiler run only (gpurun merges its output directory: an older run's <pid>_*.csv may still lie beside the new one)"""
    fs = glob.glob(pattern, recursive=True)
    if not fs:
        return []
    last = max(fs, key=os.path.getmtime)
    pid = os.path.basename(last).split("_")[0]
    return [f for f in fs if os.path.basename(f).split("_")[0] == pid and os.path.dirname(f) == os.path.dirname(last)]


rows = []
for f in newest(f"gpurun_out/prof_{tag}_trace/**/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        rows.append(r)
with open(f"profiles/{tag}_kernel_stats.csv", "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["Kernel", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        nm = r["Name"]
        w.writerow([short(nm) if any(k in nm for k in ours) else nm[:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])

trace = defaultdict(list)
meta = {}
for f in newest(f"gpurun_out/prof_{tag}_trace/**/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        trace[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        meta[k] = {x: r[x] for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X")}

pmc = defaultdict(lambda: defaultdict(list))          # kernel -> counter -> [(dispatch id, value)]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in newest(f"gpurun_out/prof_{tag}_{c}/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))

lines = [f"# rocprofv3 summary, {tag}", "",
         f"Command: `rocprofv3 --kernel-trace --stats -- python3 bench.py {bench_args} --no-cpu-baseline` (1x MI355X, N = 65536 envs, BS1/OP1, float32,",
         "in-kernel RNG).  bench.py runs four legs, each on a fresh handle: the headline `ptg_rollout` with row-major observations, `ptg_step`",
         "(K launches replayed as one hipGraph), `ptg_rollout` with feature-major observations and with float64 row-major observations (`<row,f64>`).",
         "Dispatches of `k_rollout_pc<row>` in trace order:",
         f"the warm-up launch(es) from reset ({W} steps as {NW} launch(es)), THE TIMED {K}-step LAUNCH (bench.py's `roofline.avg_launch_us`), then the two 400-step",
         "steady-state rollouts (`steady_state`; 250 + 150 steps each); `k_rollout_pc<feature>`: warm-up, timed.  `k_refresh` is the table refresher's",
         "rolling-pass kernel (forked from the rollout's stream; only for launches long enough to need one, DESIGN.md section 5).  HBM counters: separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the",
         "same command with `--launch eager`.  FETCH_SIZE is doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of streamed reads); counter unit KiB.", "",
         "| kernel | dispatches | avg us | min us | max us | VGPR | SGPR | LDS B | scratch | block | grid |", "|---|---|---|---|---|---|---|---|---|---|---|"]
for k in sorted(trace, key=lambda x: -sum(v[1] for v in trace[x])):
    if not any(k.startswith(o) for o in ours):
        continue
    d, m = [x[1] for x in sorted(trace[k])], meta[k]
    lines.append(f"| {k} | {len(d)} | {sum(d) / len(d) / 1e3:.2f} | {min(d) / 1e3:.2f} | {max(d) / 1e3:.2f} | {m['VGPR_Count']} | {m['SGPR_Count']} | {m['LDS_Block_Size']} | {m['Scratch_Size']} | {m['Workgroup_Size_X']} | {m['Grid_Size_X']} |")
lines.append("")
for k in sorted(trace):
    if k.startswith("k_rollout_pc"):
        d = [x[1] for x in sorted(trace[k])]
        lines.append(f"`{k}` dispatches in order [us]: " + ", ".join(f"{x / 1e3:.1f}" for x in d) +
                     (f"  -> timed {K}-step launch: **{d[NW] / 1e3:.1f} us** = {d[NW] / 1e3 / K:.3f} us per step" if len(d) > NW else "") +
                     (f"; second 400-step steady rollout: {sum(d[NW + 1 + (len(d) - NW - 1) // 2:]) / 1e3:.1f} us = {sum(d[NW + 1 + (len(d) - NW - 1) // 2:]) / 1e3 / 400:.3f} us per step" if len(d) > NW + 2 else ""))
# start / end stamps of the headline launch and of every table-refresher dispatch near it (the interval bench.py reports
# is the union of the two; since round 3 the pass at the head of a launch runs INSIDE the rollout kernel and no k_refresh precedes it)
hk = "k_rollout_pc<row>"
if hk in trace and len(trace[hk]) > NW:
    disp = sorted(trace[hk])
    refs = sorted(trace.get("k_refresh", []))
    lines += ["", f"Dispatch stamps around `{hk}` (ns, relative to the start of the timed {K}-step launch; from the kernel trace):", "",
              "| dispatch | start | end | duration us |", "|---|---|---|---|"]
    t0 = disp[NW][0]
    for name, lst in (("k_rollout_pc<row> warm-up launch", disp[0:NW]), (f"k_rollout_pc<row> TIMED {K}-step launch", disp[NW:NW + 1]),
                      ("k_rollout_pc<row> next launch (steady_state leg)", disp[NW + 1:NW + 2])):
        for st, du in lst:
            lines.append(f"| {name} | {st - t0} | {st + du - t0} | {du / 1e3:.2f} |")
    near = [(st, du) for st, du in refs if disp[0][0] - 50000 <= st <= (disp[NW + 1][0] + disp[NW + 1][1] if len(disp) > NW + 1 else disp[NW][0] + disp[NW][1] + 50000)]
    for st, du in near:
        lines.append(f"| k_refresh | {st - t0} | {st + du - t0} | {du / 1e3:.2f} |")
    inside = [1 for st, du in refs if st + du > disp[NW][0] - 20000 and st < disp[NW][0] + disp[NW][1]]
    lines.append("")
    lines.append(f"k_refresh dispatches that end within 20 us before the timed launch or overlap it: **{len(inside)}**" +
                 (" (the head pass is inside the rollout kernel; rolling passes only in launches long enough to need one)" if not inside else ""))
# the bench line of the PROFILED run: its own event-based figures for the same launches
try:
    line = [l for l in open(f"gpurun_out/prof_{tag}_trace.log") if l.startswith("{")][-1]
    bj = json.loads(line)
    legs = [("headline (row-major float32)", bj["roofline"])] + [(k, v["roofline"]) for k, v in bj.get("also", {}).items()]
    lines += ["", "bench.py's own figures in this same profiled run (HIP events attached to the launches).  Two things to know when comparing: (i) with the",
              "profiler attached the event pair reads 4-7 us MORE than the dispatch duration above (tools/tsweep.py under rocprofv3: 10.5-15.9 vs 6.5-7.2 us at",
              "T = 1, 37-40 vs 33-37 at T = 20), without it the events match these dispatch durations (6.3-6.6 us at T = 1, 33-35 at T = 20); (ii) the whole",
              "profiled process runs 8-10 % slower than an un-profiled one (steady state 1.60-1.65 vs 1.47-1.53 us per step; MI355X_MICROARCH.md, DVFS",
              "give-back item 2: never compare a profiled arm with an un-profiled one).  The un-profiled driver-argument run is profiles/r03_bench_driver_args.json."]
    for nme, r in legs:
        lines.append(f"* {nme}: avg_launch_us = {r['avg_launch_us']:.2f} over {r['launches_timed']} launch(es), frac = {r['frac']:.3f}" +
                     (f" (kernel only {r['kernel_only_us']:.2f} us, refresher beside it {r['refresh_us']:.2f} us)" if "kernel_only_us" in r else ""))
    if "steady_state" in bj:
        lines.append(f"* steady_state: {bj['steady_state']['us_per_step']:.3f} us per step, frac = {bj['steady_state']['frac']:.3f}")
except Exception as e:
    lines += ["", f"(no bench line found in gpurun_out/prof_{tag}_trace.log: {e})"]
traffic = {"source": f"profiles/{tag}_summary.md"}
lines += ["", "| kernel | dispatch | FETCH_SIZE KiB (raw) | read bytes (x2 corrected) | WRITE_SIZE KiB | HBM bytes | per step / launch |", "|---|---|---|---|---|---|---|"]
for k in sorted(pmc):
    if not (k.startswith("k_rollout_pc") or k.startswith("k_step_hot")):
        continue
    fs = [v for _, v in sorted(pmc[k].get("FETCH_SIZE", []))]
    ws = [v for _, v in sorted(pmc[k].get("WRITE_SIZE", []))]
    layout = k[k.index("<") + 1:-1]
    dt = "float32"
    if layout.endswith(",info"):
        continue
    if layout.endswith(",f64"):
        layout, dt = layout[:-4], "float64"
    if k.startswith("k_rollout_pc") and len(fs) > NW and len(ws) > NW:
        f, w_ = fs[NW], ws[NW]                                # the timed launch (trace order: warm-up launch(es), timed, ...)
        tot = 2 * f * 1024 + w_ * 1024
        lines.append(f"| {k} | timed {K}-step launch | {f:.1f} | {2 * f * 1024:.0f} | {w_:.1f} | {tot:.0f} | {tot / K:.0f} B per step = {tot / K / 65536:.1f} B per env-step |")
        traffic[f"rollout_{layout}_{dt}"] = {"bytes_per_step": tot / K, "steps_in_measured_launch": K}
        if len(fs) > NW + 2 and len(ws) > NW + 2 and len(fs) == len(ws):      # the two 400-step steady-state rollouts (each one or more launches): the second one
            h2 = (len(fs) - NW - 1) // 2
            f, w_ = sum(fs[NW + 1 + h2:]), sum(ws[NW + 1 + h2:])
            tot = 2 * f * 1024 + w_ * 1024
            lines.append(f"| {k} | second 400-step steady rollout ({h2} launch(es)) | {f:.1f} | {2 * f * 1024:.0f} | {w_:.1f} | {tot:.0f} | {tot / 400:.0f} B per step = {tot / 400 / 65536:.1f} B per env-step |")
            traffic[f"rollout_{layout}_{dt}_steady"] = {"bytes_per_step": tot / 400, "steps_in_measured_launch": 400}
    elif k.startswith("k_step_hot") and fs and ws:
        f, w_ = sum(fs) / len(fs), sum(ws) / len(ws)
        tot = 2 * f * 1024 + w_ * 1024
        lines.append(f"| {k} | average of {len(fs)} launches | {f:.1f} | {2 * f * 1024:.0f} | {w_:.1f} | {tot:.0f} | {tot / 65536:.1f} B per env-step |")
        traffic[f"step_{layout}_{dt}"] = {"bytes_per_launch": tot}
open(f"profiles/{tag}_summary.md", "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(f"profiles/{tag}_traffic.json", "w"), indent=1)
if len(traffic) > 1:                                          # what bench.py reports as roofline.traffic
    json.dump(traffic, open("profiles/traffic_latest.json", "w"), indent=1)
print("\n".join(lines))

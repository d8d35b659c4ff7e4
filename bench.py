#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched PtG env step on MI355X (BASELINE.json metric), with roofline and CPU baseline.

    python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one rank per GPU, RCCL); started WITHOUT a launcher (`python bench.py --gpus N`, WORLD_SIZE unset) it spawns the N ranks itself,
before anything in the parent touches a GPU.  Weak scaling: every rank owns --envs (default 65 536) envs; no per-step
communication.  The one collective of the path -- the all-gather of finished-episode returns -- is issued only when the timed
window contains an episode boundary ("at episode boundaries only"; the host knows the step on which a synchronised
batch's episodes end, ptg_steps_to_episode_end); its cost is reported by the `episode_boundary` leg, which times a window that
does contain one.

A "step" is one vector step of the hot path over the whole batch (N_envs env-steps per rank).  Headline: `ptg_rollout`
(K steps fused into as few launches as fit) writing ROW-MAJOR [N, F] float32 observations -- the layout of the boundary
(DummyVecEnv hands SB3 one row per env).  `also`: the same rollout with feature-major [F, N] observations, and `ptg_step`
(one launch per vector step, the K launches replayed as one hipGraph).  Actions, state, observations, rewards and done
flags are resident in HBM; nothing crosses PCIe inside the timed region.
Workload: BASELINE.json configs[2] -- N = 65 536 envs, BS1/OP1, synthetic 38-day trace (32-day episodes), 'mod'
features, discrete sticky actions, in-kernel counter RNG for the state-change noise.

Timing.  `value` is wall clock over the K timed steps: barrier + synchronize, start, K steps (+ the finished-episode all-gather when
the window holds an episode boundary), synchronize, stop, barrier; N > 1: the MAX of that over the ranks.  `roofline` uses device
time: every timed launch carries a HIP event pair stamped at the kernel's begin and end (ptg_profile, hipExtLaunchKernelGGL) -- what
rocprofv3 --kernel-trace reports for the dispatch -- because an event recorded from Python on an idle stream also counts the host's
launch latency.  The interval is the UNION of the rollout kernel and the table refresher that may run beside it (ptg_profile_read_ex:
first start to last end; `refresh_us` = the helper's own duration, 0 when the launch needed none -- the pass at the head of a launch is
part of the rollout kernel itself since round 3).  `roofline` = the K timed steps exactly as the driver asked for them (few steps
right after a reset: launch prologue and drain are not amortised); `steady_state` = a 400-step launch run after the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy peak


def algorithmic_bytes_per_env_step(obs_dim, out_bytes, path):
    """Compulsory HBM bytes per env-step of the shipped SoA layout.  Process tables, price series and
    the noise tape are cache-resident and not counted."""
    action = 4                               # int32 action
    state = 16 + 16                          # StA {i, j, k, flags} + StB {cum_rew f64, act_ep_d, noise_ctr}; StC {n_changes, ep_ptr}
                                             # is touched on penalised state changes / resets only and is not counted
    out = obs_dim * out_bytes + out_bytes + 1    # obs row + reward + done
    if path == "rollout":                    # state stays in registers between the steps of one launch
        return action + out
    return action + 2 * state + out


def _cpu_sample(po, spec, n_envs, n_steps, threads, seed):
    m = spec.markets[0]
    consts = dict(spec.consts, scenario=m["scenario"], rew_l_b=m["rew_l_b"], rew_u_b=m["rew_u_b"], r_0=m["r_0"])
    # eps_ind=None (episode offset 0): the synthetic trace has one episode, and n_envs reference envs would exhaust eps_ind
    market = dict(el=m["el"], pot_rew=m["pot_rew"], part_full=m["part_full"], gas=m["gas"], eua=m["eua"], eps_ind=None)
    env = po.OracleVecEnv(consts, spec.tables, market, n_envs, ep_index0=0)
    rng = np.random.default_rng(seed)
    env.set_noise_tape(rng.normal(0, consts["noise"], (n_envs, 256)))
    env.reset()
    cur = rng.integers(0, 5, n_envs)
    tapes = []
    for _ in range(n_steps):
        sw = rng.random(n_envs) < 1 / 12.0
        cur = np.where(sw, rng.integers(0, 5, n_envs), cur)
        tapes.append(cur.astype(np.int32))
    for t in range(5):
        env.step_reuse(tapes[t], n_threads=threads)
    t0 = time.perf_counter()
    for t in range(5, n_steps):
        env.step_reuse(tapes[t], n_threads=threads)
    dt = time.perf_counter() - t0
    env.close()
    return n_envs * (n_steps - 5) / dt, dt


def cpu_baseline(spec, n_envs=131072, n_steps=105, seed=7):
    """Oracle (CPU restatement of the reference, oracle/ptg_oracle.c) timed on this host's cores: bounded samples, all cores
    (the reported value) and one thread (SURVEY.md §8(d) asks for both)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ptg_oracle as po
    po.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                  # a container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) beats the visible core count
        quota = None
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            quota = None if q == "max" else int(q) / int(p)
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = None if q <= 0 else q / p
        if quota:
            cores = max(1, min(cores, int(quota + 0.5)))
    except (OSError, ValueError):
        pass
    if cores <= 16:                                       # small hosts / quotas: keep the sample inside a few seconds
        n_envs, n_steps = 65536, 305
    v_all, dt_all = _cpu_sample(po, spec, n_envs, n_steps, cores, seed)        # >= 512 envs per thread and step on a 256-core host
    v_one, dt_one = _cpu_sample(po, spec, 4096, 205, 1, seed)
    return {"value": v_all, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle/ptg_oracle.c (OpenMP over envs), {n_envs} envs x {n_steps - 5} steps of the same workload, {dt_all:.1f} s wall",
            "single_thread": {"value": v_one, "unit": "env-steps/s", "cores": 1, "sample": f"4096 envs x 200 steps, {dt_one:.1f} s wall"},
            # the reference's own Python cannot travel to the GPU box; its figure is the survey's, reported only
            "reference_python": {"value": [1.0e4, 1.8e4], "unit": "env-steps/s per core", "measured_in_this_run": False,
                                 "host": "build container, 1 of 8 vCPU Intel Xeon @ 2.10 GHz, unmodified env/ptg_gym_env.py, 1 env",
                                 "source": "BASELINE.md section 3 (random actions 1.03e4 ... held actions 1.79e4)",
                                 "published_pipeline_fps": 166,      # the reference's only published figure: SB3 time/fps of a whole PPO run
                                 }}


def spawn_ranks(n):
    """`python bench.py --gpus N` without torch.distributed.run: start the N ranks as fresh child processes (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1) and exit with the worst of their codes.  The parent makes no
    HIP call (torch.cuda.device_count() does not initialise the GPU on this image)."""
    import socket
    import subprocess
    backend = os.environ.get("PTG_BENCH_BACKEND", "nccl")
    try:
        import torch
        have = torch.cuda.device_count()
    except Exception:
        have = 0
    if backend == "nccl" and have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible; RCCL needs one GPU per rank "
                         "(PTG_BENCH_BACKEND=gloo rehearses the N > 1 path with several ranks per GPU)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--path", choices=["step", "rollout"], default="rollout",
                    help="timed path: ptg_rollout (K steps fused) or ptg_step (one launch per step); the other legs go under \"also\"")
    ap.add_argument("--launch", choices=["graph", "eager"], default="graph", help="step path: replay the K launches as one hipGraph, or launch eagerly")
    ap.add_argument("--no-also", dest="also", action="store_false")
    ap.add_argument("--no-steady", dest="steady", action="store_false", help="skip the 400-step steady-state launch after the timed region")
    ap.add_argument("--scenario", type=int, default=1)
    ap.add_argument("--operation", default="OP1")
    ap.add_argument("--out-dtype", choices=["float32", "float64"], default="float32")
    ap.add_argument("--obs-layout", choices=["row", "feature", "sb3_flat", "split"], default="row",
                    help="observation matrix layout: row-major [N][F] (the boundary's), feature-major [F][N], SB3's flattened [N][F+5] rows, or the 16-column env part + series indices (\"split\")")
    ap.add_argument("--p-switch", type=float, default=1.0 / 12.0, help="per-step probability of drawing a new action")
    ap.add_argument("--noise", choices=["rng", "tape"], default="rng", help="in-kernel counter RNG or a device-filled tape")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mixed-scenarios", action="store_true",
                    help="BASELINE.json configs[4]'s per-rank leg: business scenarios 1, 2, 3 mixed inside the batch (global env e -> scenario e %% 3), one operation level")
    ap.add_argument("--no-boundary-leg", dest="boundary_leg", action="store_false",
                    help="skip the episode_boundary leg (a window that contains the episode end + the finished-episode all-gather)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)                     # no launcher: this process becomes the launcher (and never touches a GPU)

    import torch
    import torch.distributed as dist
    from rl_ptg_amd import dist as ptg_dist
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    from rl_ptg_amd.synthetic import sticky_actions_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; running with {world} rank(s)", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the env step has no CPU path")
    backend = os.environ.get("PTG_BENCH_BACKEND", "nccl")       # "gloo": rehearse the N > 1 path with several ranks on one GPU
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    coll_device = device if backend == "nccl" else torch.device("cpu")
    # PTG_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, barriers, the collectives) with ONE rank -- the rehearsal of the
    # RCCL calls a one-GPU box allows (profiles/r02_bench_rccl_1rank.log)
    multi = world > 1 or bool(os.environ.get("PTG_BENCH_FORCE_DIST"))
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    coll_stream = torch.cuda.Stream(device=device)
    K, W, n = args.steps, args.warmup, args.envs
    n_total = n * world
    if args.mixed_scenarios:
        from rl_ptg_amd.prep import EnvSpec
        spec = EnvSpec.merge_scenarios([synthetic_spec(scenario=sc, operation=args.operation, eps_len_d=32)[0] for sc in (1, 2, 3)])
    else:
        spec, _ = synthetic_spec(scenario=args.scenario, operation=args.operation, eps_len_d=32)

    def configure(eng):
        """what every leg's fresh handle gets: this rank's place in the job (episode plan, RNG streams, scenario mix)"""
        eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
        eng.set_global_env_offset(first_ptr - n_total)
        if args.mixed_scenarios:
            eng.set_market_assignment(ptg_dist.mixed_scenario_assignment(n_total, world, rank, len(spec.markets)))
    first_ptr, stride = ptg_dist.episode_plan(n_total, world, rank)
    STEADY_T = 400

    def measure(path, launch, layout, out_dtype, steady):
        """W warm-up steps, then exactly K timed steps of `path`.  Returns a dict: wall seconds, per-launch kernel times, ..."""
        torch.cuda.empty_cache()                              # each leg starts from a fresh allocator state (no recycled multi-GB blocks)
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=local_rank, out_dtype=out_dtype, obs_layout=layout, obs_pitch="auto")
        F = eng.obs_dim
        configure(eng)
        if args.noise == "rng":
            eng.set_noise_rng(seed=20250614)                  # counter-based draws inside the kernels
        else:
            eng.fill_noise_tape(seed=20250614, per_env_len=min(max(K + W + 8, 64), 1024))
        extra = STEADY_T * 2 if (steady and path == "rollout") else 0
        again = K if (path == "step" and launch == "graph") else 0      # the graph leg's K steps once more, eagerly, for per-launch kernel times
        actions = sticky_actions_device(K + W + extra + again, n, seed=1234 + rank, device=device, p_switch=args.p_switch)
        eng.reset()
        bufs = None
        if path == "rollout":
            rows = max(K, W, STEADY_T if extra else 1)
            # zero-filled once: every page of the output buffers has been written before the timed region.  (feature-major: [rows, F, N]
            # views of [rows, F, pitch] storage when the engine chose a plane pitch, HipEngine obs_pitch="auto")
            bufs = (eng.alloc_obs(rows, zero=True), torch.zeros((rows, n), dtype=eng.out_dtype, device=device),
                    torch.zeros((rows, n), dtype=torch.uint8, device=device))

        def run(t0, cnt):
            if cnt <= 0:
                return
            if path == "rollout":
                eng.rollout(actions[t0:t0 + cnt], bufs[0][:cnt], bufs[1][:cnt], bufs[2][:cnt])
            else:
                for t in range(t0, t0 + cnt):
                    eng.step(actions[t], want_final=False)

        # untimed warm-up: the W steps go through every call the timed region makes (kernel-attached events, the stream events,
        # the finished-episode query and its collective), so the timed K steps do not pay first-call costs of the host side
        # (rollout path: the W warm-up steps run as two launches, the second one right before the timed region -- the timed call then
        #  finds the host's code path warm, as every call of a training loop after the first does; the state it starts from is the same)
        W1 = W // 2 if (path == "rollout" and W >= 2) else W
        wev = torch.cuda.Event(enable_timing=True)
        eng.profile(True)
        wev.record()
        run(0, W1)
        wev.record()
        r, l, _ = eng.finished_episodes()
        with torch.cuda.stream(coll_stream):
            for _ in range(3):                                # (the first collectives on a stream cost 2.7 ms, 100 us, 70 us; then 58: tools/agcost.py)
                ptg_dist.all_gather_finished(r, l, device=coll_device)
        torch.cuda.synchronize()
        eng.sync()
        eng.profile_read()
        eng.profile(False)
        graph = None
        if path == "step" and launch == "graph":                # K ptg_step launches captured once, replayed as one hipGraph
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    run(W, K)
            torch.cuda.current_stream(device).wait_stream(side)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if graph is None:
            eng.profile(True)                                 # kernel-attached begin / end events on every launch from here on
        run(W1, W - W1)                                       # the rest of the warm-up steps (their launch records are dropped below)
        n_launch = eng.rollout_launches(K) if path == "rollout" else K        # kernel launches inside the timed region
        # Does an episode end inside the timed window?  Host-known for a synchronised batch; all ranks must take the same branch (a
        # collective sits behind it), so the answer is agreed on OUTSIDE the clock.  Without a boundary nothing can have finished:
        # no query, no collective (the all-gather belongs to episode boundaries, not to the step path).
        s_end = eng.steps_to_episode_end()
        boundary = ptg_dist.episode_boundary_in_window(s_end, K, device=coll_device)
        # the slices the timed calls take are made before the clock starts (harness work, not the env's)
        if path == "rollout":
            timed_args = (actions[W:W + K], bufs[0][:K], bufs[1][:K], bufs[2][:K])
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        if graph is not None:                      # (stream events only where no kernel-attached ones exist)
            ev0.record()
        tA = time.perf_counter()
        if graph is not None:
            graph.replay()
            ev1.record()
        elif path == "rollout":
            eng.rollout(*timed_args)
        else:
            run(W, K)
        tB = time.perf_counter()
        r_all, l_all = (), ()
        if boundary:
            r, l, _ = eng.finished_episodes()      # episodic-return reduction: one all-gather over the ranks
            if multi:
                with torch.cuda.stream(coll_stream):   # on a stream of its own: the collective runs beside the step kernels, not behind them
                    r_all, l_all = ptg_dist.all_gather_finished(r, l, device=coll_device)
            else:
                r_all, l_all = r, l
        tD = time.perf_counter()
        eng.sync()                                 # ptg_sync: polls the stream (no interrupt wake-up latency), then reports kernel-flagged errors
        torch.cuda.synchronize()                   # (the contract's synchronise: nothing is left to wait for)
        elapsed = time.perf_counter() - t_start    # this rank's time since the common start (barrier + synchronize); MAX over ranks below
        if os.environ.get("PTG_BENCH_DEBUG"):
            print("timed region pieces us: ev0 %.0f run %.0f fin+gather %.0f sync %.0f total %.0f" % ((tA - t_start) * 1e6, (tB - tA) * 1e6, (tD - tB) * 1e6, (time.perf_counter() - tD) * 1e6, elapsed * 1e6), file=sys.stderr)
        if multi:
            dist.barrier()
        eng.sync()
        span_ms = ev0.elapsed_time(ev1) if graph is not None else 0.0      # stream events around the replayed graph (host launch latency included)
        launch_us = kernel_us = helper_us = None
        if graph is None:
            kernel_us, helper_us, launch_us = eng.profile_read_ex()      # launch_us: the union of each launch and its helper
            kernel_us, helper_us, launch_us = kernel_us[-n_launch:], helper_us[-n_launch:], launch_us[-n_launch:]      # (the timed launches only)
            eng.profile(False)
        else:
            # a replayed graph takes no kernel-attached events: the same K steps are launched once more, eagerly and OUTSIDE the timed
            # region, for the kernel's own duration (what rocprofv3 --kernel-trace reports per dispatch); `value` stays the graph replay
            eng.profile(True)
            run(W + K, K)
            kernel_us, helper_us, launch_us = eng.profile_read_ex()
            eng.profile(False)
        res = {"pitch": eng.pitch if layout == "feature" else None, "elapsed": elapsed, "span_ms": span_ms, "launch_us": launch_us, "kernel_us": kernel_us, "helper_us": helper_us, "n_launch": n_launch,
               "n_fin": len(r_all), "F": F, "steady": None, "rerun": graph is not None, "boundary": boundary, "steps_to_episode_end": s_end,
               "elapsed_rank": elapsed, "device_us_rank": float(np.sum(launch_us)) if launch_us is not None else span_ms * 1e3}
        if extra:                                             # steady state: two more 400-step launches, the second one counted
            eng.profile(True)
            for q in range(2):
                t0 = W + K + q * STEADY_T
                eng.rollout(actions[t0:t0 + STEADY_T], bufs[0][:STEADY_T], bufs[1][:STEADY_T], bufs[2][:STEADY_T])
            us = eng.profile_read()
            eng.profile(False)
            per = len(us) // 2
            res["steady"] = {"steps": STEADY_T, "launch_us": [float(u) for u in us[per:]]}
        if multi:
            # every rank's (wall, device time) to every rank: the MAX is the job's time; the list shows whether a rank lags
            mine = torch.tensor([float(rank), elapsed, res["device_us_rank"]], dtype=torch.float64, device=coll_device)
            every = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            every = torch.stack(every).cpu().numpy()
            res["per_rank"] = [{"rank": int(a[0]), "wall_us": float(a[1]) * 1e6, "device_us": float(a[2])} for a in every]
            res["ranks_seen"] = int(len(set(int(a[0]) for a in every)))
            res["elapsed"] = float(every[:, 1].max())
            dev_sum, dev_max = res["device_us_rank"], float(every[:, 2].max())
            if launch_us is not None and len(launch_us):
                res["launch_us"] = launch_us * (dev_max / max(dev_sum, 1e-9))      # slowest rank's device time, same launch count
            else:
                res["span_ms"] = dev_max * 1e-3
        eng.close()
        return res

    def roofline(path, res, out_bytes, steps):
        b_alg = algorithmic_bytes_per_env_step(res["F"], out_bytes, path)
        if res["launch_us"] is not None and len(res["launch_us"]):
            per_launch_s = float(np.mean(res["launch_us"])) * 1e-6
            launches, how = len(res["launch_us"]), ("HIP events attached to each timed kernel launch and to the table refresher beside it, if any "
                                                    "(ptg_profile_read_ex): first start to last end")
            if res.get("rerun"):
                how = ("HIP events attached to each kernel launch of an eager re-run of the same K steps right after the timed graph replay "
                       "(a replayed graph takes no per-kernel events); graph replay incl. boundaries: %.2f us per launch" % (res["span_ms"] * 1e3 / res["n_launch"]))
        else:                                                 # graph replay: stream events around the K back-to-back launches
            launches = res["n_launch"]
            span_s = res["span_ms"] * 1e-3 if res["span_ms"] > 0 else res["elapsed"]      # (no events at all: the wall clock)
            per_launch_s, how = span_s / launches, "HIP events around the replayed graph / launches (boundaries included)"
        bytes_per_launch = b_alg * n * steps / launches       # a rollout launch covers steps / launches steps (of <= 65 536 envs each)
        achieved = bytes_per_launch / per_launch_s / 1e9
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": None, "kernel": "k_step_hot" if path == "step" else "k_rollout_pc",
                "algorithmic_bytes_per_env_step": b_alg, "avg_launch_us": per_launch_s * 1e6, "launches_timed": launches,
                "steps_per_launch": steps / launches, "timing": how}
        if res.get("kernel_us") is not None and len(res["kernel_us"]):
            roof["kernel_only_us"] = float(np.mean(res["kernel_us"]))      # the rollout / step kernel's own duration
            roof["refresh_us"] = float(np.mean(res["helper_us"]))          # k_refresh beside it (0: the launch needed none; the head pass is inside the kernel)
        return roof

    out_bytes = 4 if args.out_dtype == "float32" else 8
    tj = {}
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:                                          # measured with rocprofv3 PMC passes on this workload (profiles/*_summary.md)
            tj = json.load(open(tpath))
        except Exception:
            tj = {}

    def traffic(path, layout, dtype, steps_per_launch):
        """HBM bytes per launch from the committed rocprofv3 counter passes (NOT measured in this run): (bytes, source) or (None, None)."""
        key = f"{path}_{layout}_{dtype}"
        ent = tj.get(key)
        if not ent or n != 65536:
            return None, None
        per = ent["bytes_per_step"] * steps_per_launch if path == "rollout" else ent["bytes_per_launch"]
        return per, tj.get("source")

    def leg(path, layout, dtype, steady=False):
        res = measure(path, args.launch, layout, dtype, steady)
        ob = 4 if dtype == "float32" else 8
        roof = roofline(path, res, ob, K)
        roof["traffic"], src = traffic(path, layout, dtype, roof["steps_per_launch"])
        if src:
            roof["traffic_source"] = src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not measured in this run)"
        name = (f"ptg_rollout, K steps fused ({res['n_launch']} kernel launch(es): <= 65536 envs x <= ~400 steps each)" if path == "rollout" else
                f"ptg_step, one launch per vector step ({'K launches replayed as one hipGraph' if args.launch == 'graph' else 'eager launches'})")
        d = {"path": name, "obs_layout": layout, "obs_dtype": dtype, "value": n_total * K / res["elapsed"], "unit": "env-steps/s",
             "ms_per_step": res["elapsed"] * 1e3 / K, "roofline": roof, "episode_boundary_in_window": bool(res["boundary"]),
             "steps_to_episode_end": res["steps_to_episode_end"]}
        if "per_rank" in res:
            d["per_rank"] = res["per_rank"]
            d["rccl_ranks_seen" if backend == "nccl" else "ranks_seen"] = res["ranks_seen"]
        if res["steady"]:
            us = res["steady"]["launch_us"]
            b_alg = roof["algorithmic_bytes_per_env_step"]
            t = float(np.sum(us)) * 1e-6                      # all launches covering the 400 steps (one per <= 65 536 envs)
            d["steady_state"] = {"what": f"{STEADY_T}-step ptg_rollout after the timed region (stationary state mix), kernel-attached events",
                                 "us_per_step": t * 1e6 / STEADY_T, "achieved": b_alg * n * STEADY_T / t / 1e9, "unit": "GB/s",
                                 "frac": b_alg * n * STEADY_T / t / 1e9 / HBM_PEAK_GBPS, "env_steps_per_s_device": n * STEADY_T / t}
        return d, res

    def boundary_leg(B=20, chunk=100):
        """The window the step path's timed region never holds: the last B steps of an episode INCLUDING the terminating one (generic
        kernel: termination, auto-reset over the episode plan, finished-episode compaction), the finished-episode query and the
        all-gather of the episodic returns over the ranks -- the one collective of the path, at the one place it belongs."""
        torch.cuda.empty_cache()
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=local_rank, out_dtype=args.out_dtype, obs_layout=args.obs_layout, obs_pitch="auto")
        configure(eng)
        eng.set_noise_rng(seed=20250614)
        F = eng.obs_dim
        acts = sticky_actions_device(chunk + B, n, seed=4321 + rank, device=device, p_switch=args.p_switch)
        bufs = (eng.alloc_obs(chunk, zero=True), torch.zeros((chunk, n), dtype=eng.out_dtype, device=device),
                torch.zeros((chunk, n), dtype=torch.uint8, device=device))
        eng.reset()
        pre = eng.steps_to_episode_end() - B
        t = 0
        while t < pre:                                        # (the same action rows again and again: only the position matters here)
            c = min(chunk, pre - t)
            eng.rollout(acts[:c], bufs[0][:c], bufs[1][:c], bufs[2][:c])
            t += c
        eng.sync()
        timed_args = (acts[chunk:chunk + B], bufs[0][:B], bufs[1][:B], bufs[2][:B])
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.rollout(*timed_args)
        t1 = time.perf_counter()
        r, l, _ = eng.finished_episodes()
        t2 = time.perf_counter()
        r_all, l_all = ptg_dist.all_gather_finished(r, l, device=coll_device) if multi else (r, l)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if multi:
            dist.barrier()
        out = {"what": f"the last {B} steps of the episode incl. the terminating step + finished-episode query + all-gather of the returns over {world} rank(s)",
               "steps": B, "wall_us": (t3 - t0) * 1e6, "enqueue_us": (t1 - t0) * 1e6, "finished_query_us": (t2 - t1) * 1e6, "all_gather_us": (t3 - t2) * 1e6,
               "finished_local": int(len(r)), "finished_gathered": int(len(r_all)), "dropped": eng.finished_dropped(),
               "mean_return": float(np.mean(r_all)) if len(r_all) else None, "mean_length": float(np.mean(l_all)) if len(l_all) else None,
               "env_steps_per_s_in_window": n * B / (t3 - t0)}
        eng.close()
        return out

    def wake():
        """EXPERIMENT (PTG_BENCH_WAKE=1; off by default): device pre-conditioning before the first leg -- a scratch handle runs two
        200-step rollouts and is closed again.  Measured: it does not help, it hurts (33.1-37.2 us for the timed launch against 33.2-33.6
        without it): the leg that allocates its buffers FIRST in the process is the fast one (profiles/r03_bench_spread.txt), which is
        also why the headline leg stays the first leg (PTG_BENCH_ORDER=head_last: 34.4-37.4 us)."""
        eng = HipEngine(spec.consts, spec.tables, spec.markets, min(n, 65536), device=local_rank, out_dtype="float32", obs_layout="row")
        eng.set_episode_plan(spec.eps_ind, first_ptr, stride)
        eng.set_noise_rng(seed=1)
        a = sticky_actions_device(200, eng.n, seed=99, device=device, p_switch=args.p_switch)
        eng.reset()
        bufs = (eng.alloc_obs(200), torch.empty((200, eng.n), dtype=eng.out_dtype, device=device), torch.empty((200, eng.n), dtype=torch.uint8, device=device))
        for _ in range(2):
            eng.rollout(a, *bufs)
        eng.sync()
        eng.close()
        del bufs
    order = os.environ.get("PTG_BENCH_ORDER", "head_first")          # experiment knob: "head_last" runs the other legs before the headline leg
    if os.environ.get("PTG_BENCH_WAKE"):
        wake()
    if order == "head_first":
        head, res = leg(args.path, args.obs_layout, args.out_dtype, steady=args.steady and n <= 131072)     # (400-step buffers: 3.7 GB per 65 536 envs)
    also = {}
    if args.also:
        other = "rollout" if args.path == "step" else "step"
        also[other + "_" + args.obs_layout] = leg(other, args.obs_layout, args.out_dtype)[0]
        alt = "feature" if args.obs_layout != "feature" else "row"
        also["rollout_" + alt] = leg("rollout", alt, args.out_dtype)[0]
        if args.out_dtype == "float32":                   # the reference's declared dtype (env/ptg_gym_env.py:166-202): float64 observations / rewards
            also["rollout_" + args.obs_layout + "_float64"] = leg("rollout", args.obs_layout, "float64")[0]
    if order != "head_first":
        head, res = leg(args.path, args.obs_layout, args.out_dtype, steady=args.steady and n <= 131072)

    boundary = None
    if args.boundary_leg:
        try:
            boundary = boundary_leg()
        except Exception as ex:                               # reported, never fatal for the headline
            boundary = {"error": repr(ex)}
    if rank == 0:
        ppath = os.path.join(ROOT, "profiles", "hbm_probe_latest.json")
        if os.path.exists(ppath):                     # stream rates measured on this chip next to the vendor peak
            try:
                pj = json.load(open(ppath))
                head["roofline"]["measured_stream_GBps"] = dict({k: v for k, v in pj.items() if k != "source"},
                                                                source=str(pj.get("source", "profiles/r01_hbm_probe.txt")) + " (not measured in this run)")
            except Exception:
                pass
        line = {
            "metric": "env-steps/sec at N=65536 envs; achieved HBM GB/s vs roofline",
            "value": head["value"], "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 observations / rewards" if args.out_dtype == "float32" else "f64 observations / rewards") +
                     "; env state, reward coefficients and cum_rew in f64",
            "data": "synthetic",
            "config": {"workload": f"N={n} envs/GPU, {'BS1+BS2+BS3 mixed (env e -> scenario e % 3)' if args.mixed_scenarios else 'BS' + str(args.scenario)}/{args.operation}, synthetic 38-day trace (32-day episodes), "
                                   f"'mod' features, discrete sticky actions (p_switch={args.p_switch:.4f}), noise: {args.noise}",
                       "path": head["path"], "envs_per_gpu": n, "envs_total": n_total, "obs_dtype": args.out_dtype,
                       "obs_dim": res["F"], "obs_layout": args.obs_layout, "obs_plane_pitch": res.get("pitch"), "parallelism": f"env-sharded x{world}, no per-step collective"},
            "roofline": head["roofline"],
            "finished_episodes_gathered": int(res["n_fin"]),
        }
        for key in ("episode_boundary_in_window", "steps_to_episode_end", "per_rank", "rccl_ranks_seen", "ranks_seen"):
            if key in head:
                line[key] = head[key]
        if "steady_state" in head:
            line["steady_state"] = head["steady_state"]
        if also:
            line["also"] = also
        if boundary is not None:
            line["episode_boundary"] = boundary
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(spec)
        print(json.dumps(line))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

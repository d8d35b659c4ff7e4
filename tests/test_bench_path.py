"""The pieces bench.py's N > 1 path stands on: the host's knowledge of the next episode boundary
(ptg_steps_to_episode_end -- the finished-episode all-gather is issued only in windows that contain one; the coupling it preserves is the
reference's shared ep_index, /root/reference/env/ptg_gym_env.py:9,487-493), the dropped-episode counter of the finished-episode ring, and
bench.py starting its own ranks when no launcher did."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_without_launcher_spawns_its_ranks_cpu():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must not exit with "use a launcher": the parent spawns the ranks.  On a box
    without a GPU the RCCL path refuses up front (one GPU per rank), and the gloo rehearsal path reaches the ranks, each of which then
    reports the missing GPU -- the env step has no CPU path."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU variant of this test runs the spawned ranks for real")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "one GPU per rank" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=dict(env, PTG_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stderr.count("bench.py needs a ROCm GPU") == 2       # both ranks were started and said so


@pytest.mark.gpu
def test_bench_without_launcher_two_gloo_ranks_on_one_gpu():
    """The whole N > 1 branch of bench.py, self-spawned, two ranks sharing the one GPU over gloo: one JSON line from rank 0, both ranks
    seen, no collective inside a window without an episode boundary, and the episode_boundary leg gathers every rank's episodes."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs", "4096", "--steps", "10", "--warmup", "3",
                        "--no-also", "--no-steady"], env=dict(env, PTG_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and [p["rank"] for p in d["per_rank"]] == [0, 1]
    assert d["episode_boundary_in_window"] is False and d["finished_episodes_gathered"] == 0
    assert d["config"]["envs_total"] == 8192 and d["value"] > 0 and d["roofline"]["refresh_us"] >= 0.0
    b = d["episode_boundary"]
    assert b["finished_local"] == 4096 and b["finished_gathered"] == 8192 and b["dropped"] == 0 and b["mean_length"] == 4603.0


@pytest.mark.gpu
def test_steps_to_episode_end_and_dropped_counter():
    import torch
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=2, operation="OP2", eps_len_d=1, train_steps=200000)
    n = 8
    eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout="row")
    eng.set_episode_plan(spec.eps_ind, n, n)
    eng.set_noise_rng(3)
    eng.reset()
    ep_len = int(spec.consts["eps_sim_steps"]) - 5                      # 139 steps: the step taken at k = eps_sim_steps - 6 terminates (:508-511)
    assert eng.steps_to_episode_end() == ep_len
    rng = np.random.default_rng(0)
    acts = torch.from_numpy(rng.integers(0, 5, (ep_len, n)).astype(np.int32)).cuda()
    o, r, d = eng.rollout(acts[:100])
    assert eng.steps_to_episode_end() == ep_len - 100 and int(d.sum()) == 0
    o, r, d = eng.rollout(acts[:ep_len - 100])                          # ... up to and including the terminating step
    eng.sync()
    assert int(d[-1].sum()) == n and int(d[:-1].sum()) == 0 and eng.steps_to_episode_end() == ep_len
    k = eng.get_state("k"); k[3] += 1
    eng.set_state("k", k)                                               # de-synchronised by hand: the host no longer knows
    assert eng.steps_to_episode_end() == 0
    k[3] -= 1
    eng.set_state("k", k)
    assert eng.steps_to_episode_end() == ep_len
    # the finished-episode ring holds max(2 n, 1024) entries: 130 episodes x 8 envs without a query overflow it by 16 + 8 already listed
    assert eng.finished_dropped() == 0
    r1, l1, _ = eng.finished_episodes()
    assert len(r1) == n and eng.finished_dropped() == 0
    for _ in range(130):
        eng.rollout(acts)
    eng.sync()
    r2, l2, _ = eng.finished_episodes(cap=1000)                          # 1040 finished, ring 1024, cap 1000
    assert len(r2) == 1000 and eng.finished_dropped() == 16 + 24
    assert set(l2.tolist()) == {ep_len}
    eng.close()

"""obs_layout="split" (PTG_OBS_SPLIT): the env-dependent 14 columns of SB3's flattened observation plus the hour / day index of the
env's market windows.  The full flat row rebuilt from it must equal the SB3_FLAT layout's row (itself checked against
oracle/sb3_flat_oracle.py) bit for bit -- hot kernels, generic kernels, reset rows and terminal observations -- and a first layer
evaluated through rl_ptg_amd.policy_split.FirstLayerSplit must equal the plain Linear on the flat row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("raw_modified", ["mod", "raw"])
def test_split_rows_rebuild_the_flat_rows(raw_modified):
    import torch
    from rl_ptg_amd.engine import HipEngine
    from rl_ptg_amd.policy_split import FirstLayerSplit, flat_rows_from_split
    from rl_ptg_amd.prep import synthetic_spec
    spec, _ = synthetic_spec(scenario=2, operation="OP2", eps_len_d=1, raw_modified=raw_modified, train_steps=200000)
    n, K = 300, 300                                       # 1-day episodes: 139 steps -> two terminations; 300 envs: a ragged last wave
    acts = np.random.default_rng(6).integers(0, 5, (K, n)).astype(np.int32)
    res = {}
    for layout in ("sb3_flat", "split"):
        eng = HipEngine(spec.consts, spec.tables, spec.markets, n, device=0, out_dtype="float32", obs_layout=layout)
        eng.set_episode_plan(spec.eps_ind, n, n)
        eng.set_noise_rng(8)
        o0 = eng.reset().clone()
        o, r, d = eng.rollout(acts[:250])
        so, fo = [], []
        for t in range(250, K):
            oo, rr, dd = eng.step(acts[t], want_final=True)
            eng.sync()
            so.append(oo.clone())
            if bool(dd.any()):
                fo.append(eng.final_obs.clone())
        series = eng.market_feature_series()
        res[layout] = (eng.obs_dim, o0, o, r, d, torch.stack(so), fo, series)
        eng.close()
    assert res["split"][0] == 16 and res["sb3_flat"][0] == (40 if raw_modified == "mod" else 31)
    series = res["split"][7]
    back = lambda x: flat_rows_from_split(x, series, raw_modified)
    assert torch.equal(back(res["split"][1]), res["sb3_flat"][1])              # reset rows (generic kernel)
    assert torch.equal(back(res["split"][2]), res["sb3_flat"][2])              # fused rollout incl. the terminating step's post-reset rows
    assert torch.equal(back(res["split"][5]), res["sb3_flat"][5])              # per-step launches
    assert torch.equal(res["split"][3], res["sb3_flat"][3]) and torch.equal(res["split"][4], res["sb3_flat"][4])
    assert int(res["split"][4].sum()) == n and len(res["split"][6]) == 1 == len(res["sb3_flat"][6])
    assert torch.equal(back(res["split"][6][0]), res["sb3_flat"][6][0])        # terminal observations
    # the indices are whole numbers inside the series, and they move: one hour per 6 steps
    hi = res["split"][2][:, :, 14]
    assert torch.equal(hi, hi.round()) and int(hi.max()) + 13 <= series["featA"].size and len(torch.unique(hi[:, 0])) > 20
    # first layer: split evaluation == Linear on the flat rows
    g = torch.Generator(device="cuda").manual_seed(1)
    H = 64
    W = torch.randn((H, res["sb3_flat"][0]), generator=g, device="cuda", dtype=torch.float64)
    b = torch.randn(H, generator=g, device="cuda", dtype=torch.float64)
    fl = FirstLayerSplit({k: v.astype(np.float64) for k, v in series.items()}, raw_modified, device="cuda").prepare(W, b)
    y_split = fl(res["split"][2].double())
    y_flat = res["sb3_flat"][2].double() @ W.t() + b
    assert torch.allclose(y_split, y_flat, rtol=1e-12, atol=1e-12)

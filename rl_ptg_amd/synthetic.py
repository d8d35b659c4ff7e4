"""Synthetic day-ahead price traces (BASELINE.json "synthetic 32-day price traces").

The reference ships real EPEX/THE/EUA files (data/spot_market_data/*.csv); BASELINE.json's configurations use a
synthetic trace instead.  The loader contract is the reference's (src/rl_utils.py:21-43,133-142): hourly
electricity and daily gas prices arrive in Euro/MWh and are divided by 10 to ct/kWh, EUA prices in Euro/t are
used as they are, and `len(gas) - 6` must be divisible by `eps_len_d` -- hence 38 days for 32-day episodes.
Values are rounded to two decimals in file units, like the reference's data files.
"""
import numpy as np

SYNTH_SEED = 20250614


def synthetic_market_csv_units(days=38, seed=SYNTH_SEED):
    """(el [Euro/MWh, hourly, days*24], gas [Euro/MWh, daily], eua [Euro/t, daily]) in FILE units."""
    rng = np.random.default_rng(seed)
    h = np.arange(days * 24)
    el = np.clip(8 + 6 * np.sin(2 * np.pi * h / 24) + 3 * np.sin(2 * np.pi * h / 168) + rng.normal(0, 4, len(h)), -10, 90)
    gas = np.clip(3 + 0.5 * np.cumsum(rng.normal(0, 0.2, days)), 0.4, 32)
    eua = np.clip(70 + np.cumsum(rng.normal(0, 1.5, days)), 23, 98)
    return np.round(el * 10, 2), np.round(gas * 10, 2), np.round(eua, 2)


def synthetic_market(days=38, seed=SYNTH_SEED):
    """(el, gas [ct/kWh], eua [Euro/t]) as the reference's import_market_data would return them."""
    el, gas, eua = synthetic_market_csv_units(days, seed)
    return el.astype(float) / 10, gas.astype(float) / 10, eua.astype(float)


def sticky_actions_device(n_steps, n_envs, seed, device, p_switch=1.0 / 12.0):
    """Synthetic action tape [n_steps, n_envs] int32 on `device`: per-env i.i.d. uniform{0..4} actions held for
    ~Geom(p_switch) steps; p_switch = 1 gives the adversarial uniform-random-every-step tape."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    out = torch.empty((n_steps, n_envs), dtype=torch.int32, device=device)
    if n_steps == 0 or n_envs == 0:
        return out
    # a handful of launches per column block instead of five per step: held actions = the draw at the latest switch time
    # (forward fill through a running maximum of the switch times)
    block = max(1, (1 << 26) // n_steps)
    t_idx = torch.arange(n_steps, device=device).unsqueeze(1)
    for c0 in range(0, n_envs, block):
        m = min(block, n_envs - c0)
        sw = torch.rand((n_steps, m), generator=g, device=device) < p_switch
        sw[0] = True
        new = torch.randint(0, 5, (n_steps, m), generator=g, device=device, dtype=torch.int32)
        last = torch.cummax(torch.where(sw, t_idx, torch.zeros_like(t_idx)), dim=0).values
        out[:, c0:c0 + m] = torch.gather(new, 0, last)
    return out

"""Sharding of the env batch over the GPUs of one node.

Env instances are independent: GPU g of G owns the contiguous global env range [g*n, (g+1)*n) with its own
copy of the (~8 MB) tables and price series, and NO per-step communication.  The reference's only cross-env
coupling, the module-global `ep_index` (env/ptg_gym_env.py:9,487-493), has a closed form when the envs of a
vector step terminate together: env with GLOBAL index e takes eps_ind[N_total + e + m*N_total] at its m-th reset
(`episode_plan`).  The one collective is an all-gather of the finished-episode (return, length) lists for the
episodic-return statistic (SB3's rollout/ep_rew_mean) -- RCCL over xGMI on GPUs (backend "nccl"), gloo in CPU tests.
"""
import numpy as np


def shard_range(n_total, world_size, rank):
    """Contiguous global env range of `rank`; n_total must divide evenly (weak scaling uses n_total = n * world_size)."""
    if n_total % world_size != 0:
        raise ValueError(f"n_total={n_total} is not divisible by world_size={world_size}")
    n = n_total // world_size
    return rank * n, (rank + 1) * n


def episode_plan(n_total, world_size, rank):
    """(first_ptr, stride) for ptg_set_episode_plan on this rank's shard."""
    lo, _ = shard_range(n_total, world_size, rank)
    return n_total + lo, n_total


def mixed_scenario_assignment(n_total, world_size, rank, n_sets):
    """Per-env market set for a mixed-scenario batch: global env e gets set e % n_sets."""
    lo, hi = shard_range(n_total, world_size, rank)
    return (np.arange(lo, hi) % n_sets).astype(np.uint8)


INLINE = 15          # finished episodes per rank that travel with the count in the first (usually only) collective

_PINNED = {}         # (numel, dtype) -> pinned host staging tensor (device collectives only)


def _pinned(numel, dtype, tag):
    import torch
    key = (numel, dtype, tag)
    t = _PINNED.get(key)
    if t is None:
        t = _PINNED[key] = torch.empty(numel, dtype=dtype).pin_memory()
    return t


def _gather_rows(x, dev, group, world):
    """x: CPU tensor [m, c], the same shape on every rank -> NumPy [world, m, c]: one collective.  On a GPU device the rows go
    through pinned staging buffers with asynchronous copies and ONE stream synchronisation (no pageable copies)."""
    import torch
    import torch.distributed as dist
    flat = x.reshape(-1)
    if dev.type == "cuda":
        src_h = _pinned(flat.numel(), flat.dtype, "src")
        dst_h = _pinned(flat.numel() * world, flat.dtype, "dst")
        src_h.copy_(flat)
        src = src_h.to(dev, non_blocking=True)
        out = torch.empty(flat.numel() * world, dtype=flat.dtype, device=dev)
        dist.all_gather_into_tensor(out, src, group=group)
        dst_h.copy_(out, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        return dst_h.numpy().reshape((world,) + tuple(x.shape)).copy()
    out = torch.empty((world,) + tuple(x.shape), dtype=x.dtype)
    try:
        dist.all_gather_into_tensor(out.view(-1), flat.contiguous(), group=group)
    except (RuntimeError, NotImplementedError, AttributeError):       # backend without the flat form
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous(), group=group)
        out = torch.stack(parts)
    return out.numpy()


def all_gather_finished(returns, lengths, device=None, group=None):
    """All-gather variable-length finished-episode lists over the process group.

    Returns (returns_all, lengths_all) ordered by rank.  ONE collective in the common cases: every rank sends [count, then its
    first INLINE (return, length) pairs] as 1 + 2 * INLINE float64; only when some rank finished more than INLINE episodes a second
    collective carries the remainders (padded to the longest).  With no initialised process group this is the identity."""
    import torch
    import torch.distributed as dist
    returns = np.asarray(returns, dtype=np.float64)
    lengths = np.asarray(lengths, dtype=np.int64)
    if not (dist.is_available() and dist.is_initialized()):
        return returns, lengths
    world = dist.get_world_size(group)
    dev = torch.device("cpu") if device is None else torch.device(device)
    c = len(returns)
    head = np.zeros((1, 1 + 2 * INLINE))
    head[0, 0] = c
    k = min(c, INLINE)
    head[0, 1:1 + 2 * k:2] = returns[:k]
    head[0, 2:2 + 2 * k:2] = lengths[:k]                    # episode lengths are far below 2**53: exact as float64
    heads = _gather_rows(torch.from_numpy(head), dev, group, world)[:, 0, :]
    counts = heads[:, 0].astype(np.int64)
    m = int(counts.max()) - INLINE
    rest = None
    if m > 0:                                 # rare: a rank with more than INLINE finished episodes since the last call
        payload = np.zeros((m, 2))
        if c > INLINE:
            payload[:c - INLINE, 0] = returns[INLINE:]
            payload[:c - INLINE, 1] = lengths[INLINE:]
        rest = _gather_rows(torch.from_numpy(payload), dev, group, world)
    r_all, l_all = [], []
    for rk in range(world):
        ck = int(counts[rk])
        kk = min(ck, INLINE)
        r_all.append(heads[rk, 1:1 + 2 * kk:2])
        l_all.append(heads[rk, 2:2 + 2 * kk:2].astype(np.int64))
        if ck > INLINE:
            r_all.append(rest[rk, :ck - INLINE, 0])
            l_all.append(rest[rk, :ck - INLINE, 1].astype(np.int64))
    return np.concatenate(r_all), np.concatenate(l_all)


def episode_boundary_in_window(steps_to_episode_end, n_steps, device=None, group=None):
    """Does an episode end within the next `n_steps` vector steps on ANY rank?  `steps_to_episode_end` is this rank's
    HipEngine.steps_to_episode_end() (0 = unknown to the host: a de-synchronised batch, where an episode may end on any step).  All
    ranks get the same answer (one all-reduce of a flag when a process group is up), so the finished-episode all-gather behind it is
    entered by everybody or by nobody.  SURVEY.md 8(e): the collective belongs to episode boundaries, not to the step path."""
    mine = steps_to_episode_end == 0 or steps_to_episode_end <= n_steps
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bool(mine)
    import torch
    flag = torch.tensor([1.0 if mine else 0.0], dtype=torch.float64, device=torch.device("cpu") if device is None else torch.device(device))
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    return bool(flag.item() > 0)


def merge_moments(parts):
    """Chan merge of per-shard moments: parts [world, T, 3] of (count, mean, M2) -> [T, 3], shards taken in rank order
    (every rank computes the same floating-point result).  NumPy arrays or torch tensors."""
    is_torch = hasattr(parts[0], "clone")
    first = parts[0].clone() if is_torch else np.array(parts[0], dtype=np.float64)
    c, m, M = first[..., 0], first[..., 1], first[..., 2]
    for p in parts[1:]:
        cb, mb, Mb = p[..., 0], p[..., 1], p[..., 2]
        tot = c + cb
        safe = tot + (tot == 0)                      # a step with no envs on either side stays (0, 0, 0)
        delta = mb - m
        m = m + delta * cb / safe
        M = M + Mb + delta * delta * c * cb / safe
        c = tot
    if is_torch:
        import torch
        return torch.stack([c, m, M], dim=-1)
    return np.stack([c, m, M], axis=-1)


def all_merge_moments(moments, group=None):
    """[T, 3] per-step (count, mean, M2) of this rank's envs -> the moments over all ranks' envs (identity without a process
    group): one all-gather, then merge_moments."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return moments
    import torch
    world = dist.get_world_size(group)
    src = moments.contiguous()
    if src.is_cuda and dist.get_backend(group) != "nccl":      # CPU-side backends (gloo rehearsals): collective on a host copy
        src = src.cpu()
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src, group=group)
    return merge_moments(torch.stack(parts)).to(moments.device).contiguous()

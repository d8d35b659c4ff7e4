"""Stand-in for the `gymnasium` package -- ORACLE HARNESS ONLY (never product code).

The reference env (env/ptg_gym_env.py:4-5,23,35,145,156,166-202,485) uses Gymnasium only for
  * the `gym.Env` base class: `np_random` property and `reset(seed=)` reseeding, and
  * the `spaces.Box / Discrete / Dict` constructors.
Gymnasium 0.28.1 (requirements.txt:3) is not installed in the build container and cannot be
fetched, so this ~40-line stand-in lets the UNMODIFIED reference file import and run when
tests/golden/make_golden.py generates golden vectors.  Seeding restates
gymnasium.utils.seeding.np_random (0.28): Generator(PCG64(SeedSequence(seed))).
"""
import numpy as np
from . import spaces  # noqa: F401


class Env:
    metadata = {"render_modes": []}
    render_mode = None
    _np_random = None

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(None)))
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

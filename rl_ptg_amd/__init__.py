"""rl_ptg_amd: MI355X-native batched Power-to-Gas environment (HIP kernels behind a C ABI, SB3 VecEnv surface).

    from rl_ptg_amd import PtGVecEnv, PTGEnv, HipEngine, EnvConfig, Preprocessing, EnvSpec
"""
__version__ = "0.1.0"

_LAZY = {"PtGVecEnv": "vec_env", "PTGEnv": "vec_env", "HipEngine": "engine", "PtgError": "engine", "EnvConfig": "config",
         "Preprocessing": "prep", "EnvSpec": "prep", "synthetic_spec": "prep", "load_op_tables": "tables", "load_data": "market", "import_market_data": "market"}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(f".{_LAZY[name]}", __name__), name)
    raise AttributeError(name)

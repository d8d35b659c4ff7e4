"""C ABI checks that need no GPU: the library loads, exports every entry point include/ptg_env.h declares, the ctypes
structs match the header's field order, and calls fail loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re

import pytest

from rl_ptg_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header():
    return open(os.path.join(ROOT, "include", "ptg_env.h")).read()


def test_library_builds_and_exports_every_declared_symbol():
    _lib.build()
    L = _lib.lib()
    declared = sorted(set(re.findall(r"\b(ptg_[a-z_0-9]+)\s*\(", _header())))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} is declared in include/ptg_env.h but not exported"
    assert sorted(_lib.EXPORTS) == declared
    assert L.ptg_abi_version() == int(re.search(r"#define PTG_ABI_VERSION (\d+)", _header()).group(1))


def test_ctypes_config_matches_header_field_order():
    body = re.search(r"typedef struct ptg_config \{(.*?)\} ptg_config;", _header(), re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        typ, names = decl.split(None, 1)
        for nme in names.split(","):
            fields.append((nme.strip(), typ))
    assert [f for f, _ in fields] == [f for f, _ in _lib.PtgConfig._fields_]
    for (nme, typ), (_, ct) in zip(fields, _lib.PtgConfig._fields_):
        assert (typ == "double") == (ct is C.c_double) and (typ == "int32_t") == (ct is C.c_int32), nme


def test_ctypes_struct_sizes_match_the_c_compiler(tmp_path):
    """sizeof() as gcc sees include/ptg_env.h == the ctypes mirrors (catches padding / field drift)"""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu\\n", sizeof(ptg_tables), sizeof(ptg_market), sizeof(ptg_config));return 0;}\n'
                   % os.path.join(ROOT, "include", "ptg_env.h"))
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(_lib.PtgTables), C.sizeof(_lib.PtgMarket), C.sizeof(_lib.PtgConfig)]


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = _lib.lib()
    h = C.c_void_p()
    cfg = _lib.PtgConfig(sim_step=600, time_step_op=2, price_ahead=13, eps_sim_steps=100)
    rc = L.ptg_create(C.byref(cfg), C.byref(_lib.PtgTables()), (_lib.PtgMarket * 1)(), 1, 4, 0, C.byref(h))
    assert rc == -2 and b"no HIP device" in L.ptg_last_error(None)
    from rl_ptg_amd.engine import HipEngine
    with pytest.raises(RuntimeError, match="no CPU path"):
        HipEngine({}, {}, [], 4)


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    pkg = os.path.join(ROOT, "rl_ptg_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "ptg_oracle" not in txt and "oracle/" not in txt, f

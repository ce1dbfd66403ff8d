"""The C-ABI shared library: loads, exports every symbol include/thermalporous_hip.h declares, struct
layouts agree between the header and the ctypes stub, and it refuses to work without a GPU (no
compute call is made here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "thermalporous_hip.h")).read()


def declared_functions():
    return sorted(set(re.findall(r"^\s*(?:int|const char \*)\s*(tp_[a-z0-9_]+)\s*\(", HEADER, flags=re.M)))


def test_library_exports_every_declared_symbol(hip_lib):
    from thermalporous_amd import engine
    names = declared_functions()
    assert len(names) >= 40
    assert sorted(engine.API_SYMBOLS) == names, "engine.API_SYMBOLS and the header disagree"
    for n in names:
        assert hasattr(hip_lib, n), n
    assert hip_lib.tp_version() >= 100


def _c_struct_fields(name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), HEADER, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, rest = decl.split(None, 1)
        for v in rest.split(","):
            v = v.strip()
            m = re.match(r"(\w+)\[(\d+)\]", v)
            out.append((m.group(1), ty, int(m.group(2))) if m else (v, ty, 1))
    return out


@pytest.mark.parametrize("name", ["tp_grid", "tp_params", "tp_source", "tp_options", "tp_solve_info"])
def test_ctypes_structs_match_header(name):
    from thermalporous_amd import engine
    cls = getattr(engine, name)
    cty = {"int32_t": C.c_int32, "int64_t": C.c_int64, "double": C.c_double}
    fields = _c_struct_fields(name)
    assert [f[0] for f in fields] == [f[0] for f in cls._fields_]
    for (fname, ty, n), (_, pty) in zip(fields, cls._fields_):
        assert pty == (cty[ty]*n if n > 1 else cty[ty]), fname


def test_no_cpu_fallback(hip_lib):
    """On a machine without a GPU tp_create must fail loudly; on the GPU box this test is skipped."""
    import torch
    if torch.cuda.is_available() or os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    from thermalporous_amd.engine import HipEngine, EngineError
    import cases
    spec, *_ = cases.c1_homogeneous(N=6)
    with pytest.raises(EngineError):
        HipEngine(spec, dict(pc="cpr"))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "thermalporous_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f

"""The C-ABI library loads and exports exactly the entry points include/byolo.h declares, and the
ctypes prototype table matches the header (no compute calls: runs without a GPU)."""
import os
import re
import subprocess

import pytest

from conftest import REPO


def _header_functions():
    text = open(os.path.join(REPO, "include", "byolo.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = re.findall(r"BYOLO_API\s+([\w\s\*]+?)\s*\b(byolo_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S)
    return {name: (ret.strip(), [a.strip() for a in args.split(",")] if args.strip() not in ("", "void") else [])
            for ret, name, args in decls}


def test_header_vs_prototypes_vs_exports():
    from byolo import _lib
    hdr = _header_functions()
    assert len(hdr) >= 28
    assert set(hdr) == set(_lib.PROTOTYPES), set(hdr) ^ set(_lib.PROTOTYPES)
    for name, (ret, args) in hdr.items():
        assert len(args) == len(_lib.PROTOTYPES[name][1]), name      # same arity as the header
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert {e for e in exported if e.startswith("byolo_")} == set(hdr)
    # nothing else leaks (internal kernels / launchers are hidden)
    assert not [e for e in exported if not e.startswith("byolo_") and not e.startswith("_")], exported


def test_version_and_handle_less_errors():
    import ctypes
    from byolo import _lib
    assert b"gfx950" in _lib.lib.byolo_version()
    h = ctypes.c_void_p()
    cfg = _lib.Cfg(100, 64, 3, 2, 0.1, 1000, 0.5, 0, 0)          # 100 % 32 != 0  (yolov3.py:207)
    rc = _lib.lib.byolo_create(ctypes.byref(cfg), 0, ctypes.byref(h))
    assert rc == _lib.ERR_ARG and b"multiple of 32" in _lib.lib.byolo_last_error(None)
    assert _lib.lib.byolo_destroy(None) == 0


def test_no_torch_types_in_the_abi():
    text = open(os.path.join(REPO, "include", "byolo.h")).read()
    assert "torch" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S).lower()
    assert "at::" not in text and "hipStream_t stream" not in text


def test_plan_opts_struct_matches_the_header():
    """byolo_plan_opts (include/byolo.h) field for field against the ctypes Structure of byolo/_lib.py: same names, same order,
    int32_t / float, and the size the library itself reports (the plan knobs live in the handle)."""
    import ctypes
    from byolo import _lib
    text = open(os.path.join(REPO, "include", "byolo.h")).read()
    body = re.search(r"typedef struct byolo_plan_opts \{(.*?)\} byolo_plan_opts;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, names = decl.split(None, 1)
        fields += [(n.strip(), {"int32_t": ctypes.c_int32, "float": ctypes.c_float}[ty]) for n in names.split(",")]
    assert fields == list(_lib.PlanOpts._fields_)
    h = ctypes.c_void_p()
    cfg = _lib.Cfg(64, 96, 3, 2, 0.1, 1000, 0.5, 0, 0)
    assert _lib.lib.byolo_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == 0
    o = _lib.PlanOpts()
    assert _lib.lib.byolo_get_plan_opts(h, ctypes.byref(o)) == 0
    assert o.struct_bytes == ctypes.sizeof(_lib.PlanOpts)
    o.struct_bytes -= 4                                          # a binding built against another header is refused
    assert _lib.lib.byolo_set_plan_opts(h, ctypes.byref(o)) == _lib.ERR_ARG and b"struct_bytes" in _lib.lib.byolo_last_error(h)
    o.struct_bytes += 4
    o.wino_split_bn = 64
    assert _lib.lib.byolo_set_plan_opts(h, ctypes.byref(o)) == _lib.ERR_ARG          # a field outside its range
    _lib.lib.byolo_destroy(h)


def test_two_handles_in_one_process_keep_their_own_plans(monkeypatch):
    """The environment fills a handle's plan options ONCE, at byolo_create; afterwards a handle's plan is its own: two handles of the
    same model in one process, one planned with Winograd-in-split-f16 everywhere and the 3x3 + 1x1 pairs fused, one with neither,
    report different plans through byolo_plan_* -- and a variable set AFTER a handle exists does not reach it."""
    from conftest import build_model
    from test_planner import _plan
    for k in ("BYOLO_WINO_SPLIT", "BYOLO_B2B", "BYOLO_PRECISION"):
        monkeypatch.delenv(k, raising=False)
    _, a = build_model("bayesian_yolov3_aleatoric", 608, 608, T=30)
    _, b = build_model("bayesian_yolov3_aleatoric", 608, 608, T=30)
    assert a.engine.plan_opts() == b.engine.plan_opts() and a.engine.plan_opts()["b2b"] == 1
    b.engine.set_plan_opts(b2b=0, wino_split=0)
    monkeypatch.setenv("BYOLO_B2B", "0")                        # too late for `a`: it was created with the default
    pa, pb = _plan(a.engine, 8, 30), _plan(b.engine, 8, 30)
    assert sum(1 for s in pa[0] if s[1]) == 3 and sum(1 for s in pb[0] if s[1]) == 0          # fused pairs
    assert pa[2] != pb[2]                                       # Winograd scratch is part of the arena
    assert a.engine.plan_opts()["b2b"] == 1 and b.engine.plan_opts()["wino_split"] == 0
    _, c = build_model("bayesian_yolov3_aleatoric", 608, 608, T=30)                           # a NEW handle reads the environment
    assert c.engine.plan_opts()["b2b"] == 0
    b.engine.set_plan_opts(b2b=1, wino_split=1)                 # ... and back: the same plan as `a`
    assert _plan(b.engine, 8, 30) == pa
    with pytest.raises(KeyError):
        a.engine.set_plan_opts(no_such_field=1)
    for m in (a, b, c):
        m.engine.close()

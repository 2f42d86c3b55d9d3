"""The C-ABI library loads and exports exactly the entry points include/byolo.h declares, and the
ctypes prototype table matches the header (no compute calls: runs without a GPU)."""
import os
import re
import subprocess

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

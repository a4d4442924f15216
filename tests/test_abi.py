"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the header declares."""
import ctypes
import os
import re

from conftest import ROOT


def test_library_exports_header_symbols():
    from contourist_amd import _ffi, build
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "contourist_hip.h")).read()
    declared = set(re.findall(r"\b(cx_[a-z0-9_]+)\s*\(", header))
    declared -= {"cx_ctx"}
    assert declared == set(_ffi.SYMBOLS), declared ^ set(_ffi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    L = _ffi.load()
    assert b"gfx950" in L.cx_version()


def test_no_cpu_fallback_import():
    """the product package never imports the oracle"""
    import subprocess
    import sys
    code = ("import sys; import contourist_amd.tetrahedral, contourist_amd.grid_field, contourist_amd.surface_geometry;"
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)

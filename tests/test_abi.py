"""The C-ABI library builds for gfx950 (cross-compile, no GPU needed), loads, and exports every
symbol include/faststyle_hip.h declares; the ctypes prototype table covers exactly that set."""
import ctypes
import os
import re

from faststyle_amd import _lib, build as fsbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(fs_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_and_prototype_table_agree():
    assert declared_symbols() == sorted(_lib.PROTOTYPES)


def test_product_library_builds_and_exports_every_symbol():
    so = fsbuild.build()
    assert os.path.exists(so)
    lib = ctypes.CDLL(so)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    bound = _lib.load()
    assert b"gfx950" in bound.fs_version()
    # pure host-side entry points work without a GPU
    ho, wo = ctypes.c_int(), ctypes.c_int()
    assert bound.fs_tnet_out_shape(474, 712, ctypes.byref(ho), ctypes.byref(wo)) == 0
    assert (ho.value, wo.value) == (476, 712)
    assert bound.fs_tnet_out_shape(40, 100, ctypes.byref(ho), ctypes.byref(wo)) != 0   # REFLECT needs >= 41
    assert b"41" in bound.fs_last_error()
    assert bound.fs_tnet_workspace_bytes(1, 256, 256, 0) > 0


def test_engine_refuses_to_run_without_a_gpu_or_library(monkeypatch):
    import pytest
    import torch
    from faststyle_amd import engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.FaststyleError):
        engine.Engine()                       # no CPU fallback in the product path
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libfaststyle_hip.so")
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.FaststyleError):
        _lib.load()

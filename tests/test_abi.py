"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every
symbol include/fy_cosy3.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "fy_cosy3.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fy_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from fangyan_tts_amd import build
    return ctypes.CDLL(build.build(verbose=False))


def test_every_declared_symbol_is_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 10
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_error_string_and_version(lib):
    lib.fy_last_error.restype = ctypes.c_char_p
    assert lib.fy_version() >= 100
    assert isinstance(lib.fy_last_error(), bytes)


def test_product_path_fails_loudly_without_library(monkeypatch):
    from fangyan_tts_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libfy_cosy3.so")
    with pytest.raises(_lib.FyError):
        _lib.lib()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fangyan_tts_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f

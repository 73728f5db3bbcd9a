"""CPU-side check: libkrs_hip.so builds for gfx950, loads, and exports every
symbol include/krs.h declares (no compute call: there is no GPU here)."""

import os
import re

from keras_rs_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "krs.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(krs_[a-z0-9_]+)\s*\(", text)))


def test_header_and_symbol_list_agree():
    assert _declared() == sorted(L.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from keras_rs_amd.build import build

    build()
    lib = L.lib()
    for name in _declared():
        assert hasattr(lib, name), f"libkrs_hip.so does not export {name}"
    assert lib.krs_version() == 100


def test_struct_layouts_match_header():
    assert L.TABLE_DT.itemsize == 32 and L.FEATURE_DT.itemsize == 24
    import ctypes

    assert ctypes.sizeof(L.GemmEpilogue) == 80

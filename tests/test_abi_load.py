"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "cglb_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cglb_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    from cglb_amd import _lib
    declared = _declared_symbols()
    assert declared, "no declarations parsed from include/cglb_hip.h"
    assert sorted(_lib.SIGNATURES) == declared
    assert os.path.exists(_lib.lib_path()), "libcglb_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.lib_path())
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/cglb_hip.h but not exported"
    assert _lib.load().cglb_version() >= 100

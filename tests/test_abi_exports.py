"""CPU: the C-ABI library loads and exports every symbol include/odrift.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'odrift.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(odr_[a-z0-9_]+)\s*\(', src)))


def test_header_and_binding_agree():
    from opendrift_amd import _abi
    assert _declared() == _abi.EXPORTS


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from opendrift_amd import _abi
    lib = ctypes.CDLL(_abi.LIB_PATH)
    missing = [s for s in _declared() if not hasattr(lib, s)]
    assert not missing, missing
    assert b'gfx950' in _abi.load().odr_version()


def test_missing_library_fails_loudly(monkeypatch):
    from opendrift_amd import _abi
    monkeypatch.setattr(_abi, '_lib', None)
    monkeypatch.setattr(_abi, 'LIB_PATH', '/nonexistent/libodrift_hip.so')
    with pytest.raises(_abi.OdrError):
        _abi.load()

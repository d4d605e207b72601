"""The C-ABI library builds, loads and exports every symbol include/vlmo_hip.h
declares (no compute: runs without a GPU)."""
import ctypes
import os
import re

from exploremultimodal_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'vlmo_hip.h')).read()
    return sorted(set(re.findall(r'\b(vlmo_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_header_symbols():
    import __graft_entry__ as ge
    if not os.path.exists(hip.LIB_PATH):
        ge.build()
    L = ctypes.CDLL(hip.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f'{n} declared in vlmo_hip.h but not exported'
    assert sorted(hip.exported_symbols()) == names, 'hip.py binding list and header disagree'
    L.vlmo_abi_version.restype = ctypes.c_int
    assert L.vlmo_abi_version() == hip.ABI_VERSION == 5


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    monkeypatch.setattr(hip, 'LIB_PATH', '/nonexistent/libvlmo_hip.so')
    monkeypatch.setattr(hip, '_lib', None)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        hip.lib()

"""CPU: the C-ABI library loads and exports every symbol include/ralf_hip.h declares."""
import os
import re

from conftest import ROOT


def header_symbols():
    src = open(os.path.join(ROOT, "include", "ralf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ralf_[a-z0-9_]+)\s*\(", src)))


def test_header_matches_abi_table_and_library():
    from ralf_amd import _lib
    from ralf_amd._abi import SIGNATURES

    syms = header_symbols()
    assert syms, "no symbols parsed from header"
    assert sorted(SIGNATURES) == syms
    L = _lib.lib()
    for s in syms:
        assert hasattr(L, s), f"libralf_hip.so does not export {s}"
    assert L.ralf_abi_version() == 26
    assert isinstance(L.ralf_last_error(), bytes)


def test_workspace_query_is_pure_host():
    from ralf_amd import _lib

    L = _lib.lib()
    assert L.ralf_knn_topk_ip_workspace_bytes(0, 64, 1, 16) == 0
    w = L.ralf_knn_topk_ip_workspace_bytes(61548, 1792, 1024, 17)
    assert w >= 61548 * 1024 * 4


def test_graft_entry_build_agrees_with_the_header():
    """__graft_entry__.build() is the driver's "does it build" check: it must accept the library the tree builds (its ABI assertion once
    lagged one version behind the header)"""
    import __graft_entry__ as g

    g.build()

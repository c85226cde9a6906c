"""CPU: the C-ABI shared library builds, loads and exports every symbol include/mdfnet_hip.h declares
(no compute calls without a GPU), and the product refuses to run without a GPU / without the library."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mdfnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdf_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from mdfnet_hip import build, SIGNATURES
    path = build.build()
    h = ctypes.CDLL(path)
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(h, name), f"{name} declared in include/mdfnet_hip.h but not exported"
    assert sorted(SIGNATURES) == declared, "python binding and header disagree"
    h.mdf_abi_version.restype = ctypes.c_int
    assert h.mdf_abi_version() == 1


def test_error_convention_without_gpu():
    """Argument errors are reported through the return code + thread-local message, never abort."""
    import mdfnet_hip
    l = mdfnet_hip.lib()
    rc = l.mdf_depth_regress_fwd(None, None, 0, None, 1, 1, 1, 1, None)
    assert rc == -1 and b"null" in l.mdf_last_error()
    rc = l.mdf_hypos_fit_fwd(7, ctypes.c_void_p(8), None, None, 0, None, ctypes.c_void_p(8), 1, 1, 1, 1, None)
    assert rc == -1 and b"mode" in l.mdf_last_error()


def test_ops_refuse_cpu_tensors():
    from mdfnet_hip import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.depth_regress(torch.rand(1, 4, 4, 4), torch.rand(1, 4, 1, 1))


def test_missing_library_fails_loudly(monkeypatch):
    import mdfnet_hip
    monkeypatch.setattr(mdfnet_hip, "_lib", None)
    monkeypatch.setattr(mdfnet_hip, "LIB_PATH", "/nonexistent/libmdfnet_hip.so")
    with pytest.raises(mdfnet_hip.MdfHipError, match="no fallback"):
        mdfnet_hip.lib()


def test_integration_notes_name_every_abi_entry():
    """INTEGRATION.md maps every entry of the C ABI to the reference code it replaces."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [name for name in _declared_symbols() if name not in text]
    assert not missing, missing

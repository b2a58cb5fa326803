"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/rvc_amd.h declares (no compute calls: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "rvc_amd.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rvc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_library_agree():
    import __graft_entry__ as ge
    ge.build()
    from rvc_amd import _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/rvc_amd.h but not exported"
    assert sorted(_native.SYMBOLS) == declared, "ctypes table and header drifted apart"
    assert lib.rvc_abi_version() == _native.ABI_VERSION == 4


def test_errors_are_reported_not_swallowed():
    from rvc_amd import _native
    lib = _native._lib
    need = ctypes.c_size_t()
    assert lib.rvc_knn_workspace_bytes(100, 10, 768, 5, ctypes.byref(need)) != 0  # k != 8
    assert b"k must be 8" in lib.rvc_last_error()
    cfg = _native.DecoderConfig()
    cfg.kind = 7
    h = ctypes.c_void_p()
    assert lib.rvc_decoder_create(ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"unknown decoder kind" in lib.rvc_last_error()


def test_no_cpu_fallback():
    """The product path must fail loudly without a HIP device instead of computing on the host."""
    import torch
    from rvc_amd import _native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_native.NativeError):
        _native.knn_index_build(torch.zeros(4, 768))

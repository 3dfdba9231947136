"""CPU: the C-ABI shared library builds for gfx950, loads, and exports exactly what include/*.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from michigan_amd import build
    return build.build(verbose=False)


def _declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            src = open(os.path.join(ROOT, "include", fn)).read()
            src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
            names |= set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"


def test_ctypes_mirror_matches_header(lib_path):
    from michigan_amd import _cabi
    assert set(_cabi.EXPORTED_SYMBOLS) == _declared_symbols()
    be = _cabi.HipBackend(lib_path)
    assert be.mg_abi_version() == _cabi.MG_ABI_VERSION
    assert be.mg_sizeof_desc(0) == ctypes.sizeof(_cabi.ConvDesc)
    assert be.mg_sizeof_desc(1) == ctypes.sizeof(_cabi.WgradDesc)
    assert be.mg_stats_workspace(1, 8 * 512 * 512, 128) > 0


def test_argument_validation_without_gpu(lib_path):
    """Entry points validate their arguments before touching the device: bad descriptors come back
    as error codes with a message (translated to RuntimeError), never as a crash."""
    from michigan_amd import _cabi
    be = _cabi.HipBackend(lib_path)
    d = _cabi.ConvDesc()
    with pytest.raises(RuntimeError, match="null tensor pointer"):
        be.mg_conv_taps(d, None)
    d.in_, d.wt, d.out = 64, 64, 64
    d.dtype, d.ntaps, d.Cin = _cabi.MG_BF16, 9, 12
    with pytest.raises(RuntimeError, match="multiple of 8"):
        be.mg_conv_taps(d, None)
    w = _cabi.WgradDesc()
    with pytest.raises(RuntimeError, match="null tensor pointer"):
        be.mg_conv_wgrad(w, None)
    with pytest.raises(RuntimeError, match="bad geometry"):
        be.mg_channel_stats(64, _cabi.MG_F32, 1, 100, 6, 1, 64, 64, None)
    # group (iv), collectives: arguments are checked before RCCL is even looked for
    with pytest.raises(RuntimeError, match="null"):
        be.mg_comm_unique_id(None)
    with pytest.raises(RuntimeError, match="null communicator"):
        be.mg_allreduce_stats(0, 64, 16, 1, None)
    with pytest.raises(RuntimeError, match="null communicator"):
        be.mg_allreduce_grads(0, 64, 16, None)


def test_kernels_are_gfx950_code_objects(lib_path):
    blob = open(lib_path, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    assert b"gfx942" not in blob and b"sm_" not in blob[:0]

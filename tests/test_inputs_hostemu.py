"""CPU check of the *kernel source* of the device input pipeline: michigan_amd/csrc/mg_inputs.hip is compiled unchanged
for the host with g++ against a stand-in for the HIP runtime (tests/hostemu/mg_common.h: one thread per block, blocks
in sequence) and run through the same C ABI on host buffers, against the oracle.  Bit-exact where the GPU tests are.
Test infrastructure only -- the product never loads this build; the -m gpu tests (tests/test_gpu_inputs.py) run the
real gfx950 kernels against the same checker and add what a serial run cannot show (workgroup scan, grid coverage).
"""
import ctypes
import os
import shutil
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INPUT_FNS = ("mg_input_crop_u8", "mg_onehot_labels", "mg_orient_to_rgb_u8", "mg_generate_hole_u8", "mg_noise_octaves",
             "mg_nearest_table", "mg_orient_rgb_table", "mg_noise_field_len", "mg_resize_bicubic_u8", "mg_bicubic_ksize", "mg_bicubic_table", "mg_inputs_set_option")


@pytest.fixture(scope="module")
def host_kernels(tmp_path_factory):
    from michigan_amd import _cabi
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    d = tmp_path_factory.mktemp("hostemu")
    shutil.copy(os.path.join(ROOT, "michigan_amd", "csrc", "mg_inputs.hip"), d / "mg_inputs.hip")
    shutil.copy(os.path.join(ROOT, "tests", "hostemu", "mg_common.h"), d / "mg_common.h")       # found first: same directory
    so = str(d / "libmg_inputs_host.so")
    subprocess.run(["g++", "-O1", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-x", "c++",
                    "-I" + os.path.join(ROOT, "include"), str(d / "mg_inputs.hip"), "-o", so], check=True)
    be = _cabi.HipBackend.__new__(_cabi.HipBackend)
    be._lib = ctypes.CDLL(so)
    for fn in INPUT_FNS:
        f = getattr(be._lib, fn)
        f.argtypes, f.restype = _cabi._PROTOS[fn]
    be._lib.mg_last_error = lambda: b"(host emulation build)"
    be.name = "hostemu"
    return be


@pytest.fixture
def both_on_host(host_kernels, monkeypatch):
    """Point tests/test_gpu_inputs._both at (kernel source on the host CPU, emulator)."""
    from michigan_amd import _cabi
    from oracle.cabi_emulator import EmulatorBackend
    import test_gpu_inputs as T

    def _both(fn, tensors):
        outs = []
        for be in (host_kernels, EmulatorBackend()):
            prev = _cabi.set_backend(be)
            try:
                outs.append(fn(*tensors, "cpu"))
            finally:
                _cabi.set_backend(prev)
        return outs
    monkeypatch.setattr(T, "_both", _both)
    return T


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("geom", [(2, 40, 40, 40, 3, 24), (3, 33, 37, 45, 1, 32)], ids=str)
def test_crop_kernel_source(both_on_host, geom, mode):
    if mode == 0 and geom[1] != geom[3]:
        pytest.skip("images are never nearest-resized")
    both_on_host.test_crop_flip_matches_oracle(geom, mode)


@pytest.mark.parametrize("geom", [(2, 64, 64, 71, 71, 3), (1, 50, 40, 37, 64, 1), (2, 96, 96, 40, 40, 3)], ids=str)
def test_bicubic_kernel_source(both_on_host, geom):
    both_on_host.test_bicubic_resize_matches_oracle(geom)


def test_onehot_kernel_source(both_on_host):
    both_on_host.test_onehot_matches_oracle()


def test_orient_rgb_kernel_source(both_on_host):
    both_on_host.test_orient_to_rgb_matches_oracle()


@pytest.mark.parametrize("geom", [(3, 96, 80), (2, 31, 45)], ids=str)
def test_hole_kernel_source(both_on_host, geom):
    both_on_host.test_generate_hole_matches_oracle(geom)
    both_on_host.test_generate_hole_empty_orientation_mask()


@pytest.mark.parametrize("size,n", [(64, 2), (40, 1), (100, 1), (72, 1), (136, 1)])
def test_noise_kernel_source(both_on_host, size, n):
    both_on_host.test_noise_octaves_match_oracle(size, n)


@pytest.mark.parametrize("size,n", [(64, 1), (100, 1), (72, 2)])
def test_noise_tiled_kernel_source(both_on_host, host_kernels, size, n):
    """The LDS-tiled variant behind mg_inputs_set_option(0, 1)."""
    host_kernels.mg_inputs_set_option(0, 1)
    try:
        both_on_host.test_noise_octaves_match_oracle(size, n)
    finally:
        host_kernels.mg_inputs_set_option(0, 0)


def test_noise_kernel_source_is_bit_identical(both_on_host):
    """Stronger than the GPU tolerance: on IEEE hardware without contraction the kernel's operation order reproduces
    the oracle's float32 result exactly."""
    from michigan_amd import inputs
    fields = torch.randn(1, inputs_len(64), dtype=torch.float64, generator=torch.Generator().manual_seed(2)) * 0.25 + 0.5
    a, b = both_on_host._both(lambda f, dev: inputs.noise_from_fields(f, 64), (fields,))
    assert torch.equal(a, b)


def inputs_len(size):
    from oracle import inputs_oracle as IO
    return sum(s * s * 3 for s in IO.noise_octave_sizes(size))

"""ctypes binding of libmichigan_hip.so -- the ONLY route from Python to the GPU kernels.

Mirrors include/michigan_hip.h one to one (struct layouts are checked against
``mg_sizeof_desc`` at load time).  There is deliberately no CPU or eager-PyTorch
fallback here: if the shared library is missing or a call fails, the caller gets
a RuntimeError.  The test-suite may swap the backend object (``set_backend``) for
the contract emulator that lives under ``oracle/`` -- product code never does.
"""
from __future__ import annotations

import ctypes
import os

MG_F32, MG_BF16 = 0, 1
MG_ACT_NONE, MG_ACT_RELU, MG_ACT_LRELU, MG_ACT_TANH = 0, 1, 2, 3
MG_EPI_PLAIN, MG_EPI_SPADE = 0, 1
MG_MAX_TAPS = 64
MG_ABI_VERSION = 6
MG_COMM_ID_BYTES = 128

_i32, _f32, _vp, _i64 = ctypes.c_int32, ctypes.c_float, ctypes.c_void_p, ctypes.c_int64


class ConvDesc(ctypes.Structure):
    """struct mg_conv_desc (include/michigan_hip.h)."""
    _fields_ = [
        ("in_", _vp), ("wt", _vp), ("out", _vp), ("bias", _vp), ("resid", _vp), ("x", _vp),
        ("mean", _vp), ("rstd", _vp), ("gamma_out", _vp),
        ("dtype", _i32), ("N", _i32), ("Hin", _i32), ("Win", _i32), ("Cin", _i32),
        ("Hout", _i32), ("Wout", _i32), ("Cout", _i32), ("Cout_gemm", _i32), ("CoutP", _i32),
        ("Hj", _i32), ("Wj", _i32), ("isy", _i32), ("isx", _i32),
        ("osy", _i32), ("osx", _i32), ("ooy", _i32), ("oox", _i32),
        ("ntaps", _i32), ("epilogue", _i32), ("act", _i32), ("slope", _f32), ("x_up", _i32),
        ("tap_dy", ctypes.c_int8 * MG_MAX_TAPS), ("tap_dx", ctypes.c_int8 * MG_MAX_TAPS),
        ("mask_slope", _f32),
    ]


class WgradDesc(ctypes.Structure):
    """struct mg_wgrad_desc (include/michigan_hip.h)."""
    _fields_ = [
        ("x", _vp), ("dy", _vp), ("dw", _vp), ("dbias", _vp),
        ("dtype", _i32), ("N", _i32), ("Hin", _i32), ("Win", _i32), ("Cin", _i32),
        ("Hj", _i32), ("Wj", _i32), ("Cg", _i32), ("isy", _i32), ("isx", _i32),
        ("ntaps", _i32), ("splitk", _i32), ("flags", _i32),
        ("tap_dy", ctypes.c_int8 * MG_MAX_TAPS), ("tap_dx", ctypes.c_int8 * MG_MAX_TAPS),
        ("det_ws", _vp), ("det_ws_bytes", _i64),
    ]


class GradSlot(ctypes.Structure):
    """struct mg_grad_slot (include/michigan_hip.h): one entry of the device-resident table mg_grad_drain walks."""
    _fields_ = [
        ("gemm", _vp), ("dbias_gemm", _vp), ("dst0", _vp), ("dst1", _vp), ("dbias0", _vp), ("dbias1", _vp),
        ("w_sn", _vp), ("u", _vp), ("v", _vp), ("sigma", _vp), ("s", _vp),
        ("cout", _i32), ("cin", _i32), ("taps", _i32), ("rows", _i32), ("cols", _i32), ("swapped", _i32),
        ("first_block", _i64),
    ]


class NormApply2Desc(ctypes.Structure):
    """struct mg_norm_apply2_desc (include/michigan_hip.h)."""
    _fields_ = [("dh", _vp * 2), ("h", _vp * 2), ("g1", _vp * 2), ("sums", _vp * 2),
                ("x", _vp), ("mean", _vp), ("rstd", _vp), ("dx", _vp), ("P", _i64),
                ("dtype", _i32), ("C", _i32), ("up", _i32), ("H", _i32), ("W", _i32),
                ("act", _i32 * 2), ("slope", _f32 * 2), ("inv_count", _f32)]


class PackJob(ctypes.Structure):
    """struct mg_pack_job (include/michigan_hip.h)."""
    _fields_ = [("w0", _vp), ("w1", _vp), ("dst", _vp), ("sigma", _vp),
                ("dtype", _i32), ("cout", _i32), ("cin", _i32), ("taps", _i32), ("rows_p", _i32), ("cols_p", _i32),
                ("mode", _i32), ("pad_", _i32), ("first_block", _i64)]


class PyramidDesc(ctypes.Structure):
    """struct mg_pyramid_desc (include/michigan_hip.h)."""
    _fields_ = [("plane", _vp * 8), ("nstride", _i64 * 8),
                ("nplanes", _i32), ("N", _i32), ("H", _i32), ("W", _i32), ("nlev", _i32), ("cout", _i32), ("dtype", _i32), ("pad_", _i32),
                ("h", _i32 * 8), ("w", _i32 * 8), ("out", _vp * 8)]


class SnLayer(ctypes.Structure):
    """struct mg_sn_layer (include/michigan_hip.h)."""
    _fields_ = [("w", _vp), ("u", _vp), ("v", _vp), ("u_copy", _vp), ("v_copy", _vp), ("sigma", _vp),
                ("t1", _vp), ("t2", _vp), ("partial", _vp),
                ("rows", _i32), ("cols", _i32), ("first_block_k1", _i32), ("first_block_k3", _i32)]


# name -> (argtypes, restype); descriptors are passed by reference.
_PROTOS = {
    "mg_conv_taps": ([ctypes.POINTER(ConvDesc), _vp], _i32),
    "mg_conv_wgrad": ([ctypes.POINTER(WgradDesc), _vp], _i32),
    "mg_wgrad_det_workspace": ([ctypes.POINTER(WgradDesc)], _i64),
    "mg_stats_workspace": ([_i32, _i64, _i32], _i64),
    "mg_channel_stats": ([_vp, _i32, _i32, _i64, _i32, _i32, _vp, _vp, _vp], _i32),
    "mg_channel_stats_finalize": ([_vp, _i32, _i32, _i64, _i32, _f32, ctypes.c_double, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _i32),
    "mg_norm_finalize": ([_vp, _i32, _i32, ctypes.c_double, _f32, _f32, _vp, _vp, _vp, _vp, _vp], _i32),
    "mg_norm_act_fwd": ([_vp, _vp, _i32, _i32, _i64, _i32, _vp, _vp, _i32, _f32, _vp, _vp], _i32),
    "mg_norm_bwd_reduce": ([_vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp, _vp, _i32, _f32, _vp, _vp, _vp, _vp], _i32),
    "mg_norm_bwd_apply": ([_vp, _vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _i32, _f32, _vp, _vp], _i32),
    "mg_norm_bwd_apply2": ([ctypes.POINTER(NormApply2Desc), _vp], _i32),
    "mg_norm_apply2_supported": ([_i32, _i32], _i32),
    "mg_norm_bwd_reduce_up": ([_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _f32, _vp, _vp, _vp, _vp], _i32),
    "mg_act_bwd": ([_vp, _vp, _vp, _i32, _i64, _i32, _f32, _vp], _i32),
    "mg_upsample2x_fwd": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_upsample2x_bwd": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_reflect_pad_fwd": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_reflect_pad_bwd": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_avgpool3s2_fwd": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_avgpool3s2_bwd": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_maxpool2_fwd": ([_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_maxpool2_bwd": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_assemble_nhwc8": ([_vp, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i64, _vp], _i32),
    "mg_grad_sum_act": ([_vp, _vp, _vp, _vp, _i32, _i64, _i32, _f32, _vp], _i32),
    "mg_blend_fwd": ([_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _f32, _vp], _i32),
    "mg_blend_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _f32, _vp], _i32),
    "mg_pack_weight": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_unpack_wgrad": ([_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_l1_mean_fwd": ([_vp, _vp, _i32, _i64, _vp, _vp, _vp], _i32),
    "mg_l1_mean_bwd": ([_vp, _vp, _vp, _i32, _i64, _vp, _vp], _i32),
    "mg_pack_weights": ([_vp, _i32, _vp, _i32, _vp], _i32),
    "mg_pack_job_blocks": ([_i32, _i32, _i32, _i32, _i32, _i32], _i64),
    "mg_sn_power_iteration": ([_vp, _i32, _vp, _i32, _vp, _i32, _i32, _f32, _vp], _i32),
    "mg_sn_layer_blocks": ([_i32, _i32, _i32], _i64),
    "mg_grad_drain": ([_vp, _i32, _vp, _i32, _vp, _vp], _i32),
    "mg_grad_slot_blocks": ([_i32, _i32, _i32], _i64),
    "mg_wide_edge_weight": ([_vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp], _i32),
    "mg_hinge_fwd": ([_vp, _vp, _i32, _i64, _i32, _vp, _vp], _i32),
    "mg_hinge_bwd": ([_vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp], _i32),
    "mg_nearest_pyramid": ([ctypes.POINTER(PyramidDesc), _vp], _i32),
    "mg_pconv_mask": ([_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp], _i32),
    "mg_pixel_affine": ([_vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp, _vp], _i32),
    "mg_bg_compose": ([_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp], _i32),
    "mg_masked_mean_fill": ([_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp], _i32),
    "mg_orient_loss_fwd": ([_vp, _vp, _vp, _i32, _i64, _vp, _i64, _i32, _i64, _vp, _vp, _vp], _i32),
    "mg_orient_loss_bwd": ([_vp, _vp, _vp, _i32, _i64, _vp, _i64, _vp, _vp, _vp, _i32, _i64, _vp, _vp], _i32),
    "mg_gabor_argmax_fwd": ([_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_gabor_argmax_bwd": ([_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_self_attention": ([_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _vp], _i32),
    "mg_sn_normalize": ([_vp, _i32, _f32, _vp, _vp, _vp, _vp], _i32),
    "mg_sn_scale": ([_vp, _vp, _vp, _i64, _vp], _i32),
    "mg_sn_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp], _i32),
    "mg_adam_step": ([_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i32, _f32, _vp], _i32),
    "mg_input_crop_u8": ([_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_onehot_labels": ([_vp, _vp, _i32, _i64, _i32, _vp], _i32),
    "mg_orient_to_rgb_u8": ([_vp, _vp, _vp, _vp, _i64, _vp], _i32),
    "mg_generate_hole_u8": ([_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp], _i32),
    "mg_noise_octaves": ([_vp, _vp, _i32, _i32, _vp], _i32),
    "mg_resize_bicubic_u8": ([_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp], _i32),
    "mg_bicubic_ksize": ([_i32, _i32], _i32),
    "mg_inputs_set_option": ([_i32, _i32], _i32),
    "mg_bicubic_table": ([_i32, _i32, _vp, _vp], _i32),
    "mg_nearest_table": ([_i32, _i32, _vp], _i32),
    "mg_orient_rgb_table": ([_vp], _i32),
    "mg_noise_field_len": ([_i32], _i64),
    "mg_comm_unique_id": ([_vp], _i32),
    "mg_comm_init": ([_vp, _i32, _i32, ctypes.POINTER(_i64)], _i32),
    "mg_comm_destroy": ([_i64], _i32),
    "mg_comm_world": ([_i64, ctypes.POINTER(_i32), ctypes.POINTER(_i32)], _i32),
    "mg_allreduce_stats": ([_i64, _vp, _i64, _i32, _vp], _i32),
    "mg_allreduce_grads": ([_i64, _vp, _i64, _vp], _i32),
    "mg_probe_mfma_layout": ([_vp, _vp], _i32),
    "mg_probe_tr16": ([_vp, _vp, _vp], _i32),
    "mg_set_option": ([_i32, _i32], _i32),
    "mg_sizeof_desc": ([_i32], _i32),
    "mg_abi_version": ([], _i32),
    "mg_last_error": ([], ctypes.c_char_p),
}
EXPORTED_SYMBOLS = tuple(_PROTOS)
_NO_STATUS = {"mg_wgrad_det_workspace", "mg_norm_apply2_supported", "mg_grad_slot_blocks", "mg_pack_job_blocks", "mg_sn_layer_blocks", "mg_stats_workspace", "mg_sizeof_desc", "mg_abi_version", "mg_last_error", "mg_noise_field_len", "mg_bicubic_ksize"}

LIB_PATH = os.environ.get("MG_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmichigan_hip.so")   # MG_LIB: a measurement variant (tools/build_variant.py)


class HipBackend:
    """The real thing: every method is one C-ABI entry point of libmichigan_hip.so."""
    name = "hip"

    def __init__(self, path: str = LIB_PATH):
        if not os.path.exists(path) and path == LIB_PATH and os.environ.get("MG_AUTOBUILD", "1") != "0":
            # a source-only checkout (the .so is git-ignored): compile the HIP library in-tree once -- still the native
            # path, never a fallback; without hipcc this raises like the missing library would
            import sys
            print("[michigan_amd] libmichigan_hip.so missing: building it with hipcc (gfx950), ~2 min", file=sys.stderr, flush=True)
            from . import build as _build
            _build.build_locked(verbose=False)          # one process builds, the other ranks of a one-process-per-GPU launch wait
        if not os.path.exists(path):
            raise RuntimeError(
                f"libmichigan_hip.so not found at {path}: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                "michigan_amd has no CPU / eager-PyTorch fallback.")
        # torch must already have loaded its libamdhip64 so that we bind to the same runtime.
        import torch  # noqa: F401
        self._lib = ctypes.CDLL(path)
        for fn, (argtypes, restype) in _PROTOS.items():
            f = getattr(self._lib, fn)          # AttributeError if a symbol is missing
            f.argtypes, f.restype = argtypes, restype
        if self._lib.mg_abi_version() != MG_ABI_VERSION:
            raise RuntimeError("libmichigan_hip.so ABI version mismatch")
        if (self._lib.mg_sizeof_desc(0) != ctypes.sizeof(ConvDesc) or self._lib.mg_sizeof_desc(1) != ctypes.sizeof(WgradDesc)
                or self._lib.mg_sizeof_desc(2) != ctypes.sizeof(GradSlot) or self._lib.mg_sizeof_desc(3) != ctypes.sizeof(PackJob)
                or self._lib.mg_sizeof_desc(4) != ctypes.sizeof(SnLayer) or self._lib.mg_sizeof_desc(5) != ctypes.sizeof(NormApply2Desc)
                or self._lib.mg_sizeof_desc(6) != ctypes.sizeof(PyramidDesc)):
            raise RuntimeError("ctypes mirror of the descriptor structs is out of sync with include/michigan_hip.h")

    def __getattr__(self, fn):
        if fn not in _PROTOS:
            raise AttributeError(fn)
        raw = getattr(self._lib, fn)
        if fn in _NO_STATUS:
            return raw
        by_ref = fn in ("mg_conv_taps", "mg_conv_wgrad", "mg_norm_bwd_apply2", "mg_nearest_pyramid")

        def call(*args):
            if by_ref:
                args = (ctypes.byref(args[0]),) + tuple(args[1:])
            rc = raw(*args)
            if rc != 0:
                msg = self._lib.mg_last_error()
                raise RuntimeError(f"{fn} failed (code {rc}): {msg.decode() if msg else '?'}")
            return rc
        self.__dict__[fn] = call
        return call


_backend = None


def backend():
    """Return the active backend, loading libmichigan_hip.so on first use."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def set_backend(obj):
    """TEST HOOK ONLY (oracle/cabi_emulator.py): install a contract emulator, return the previous backend."""
    global _backend
    prev, _backend = _backend, obj
    return prev

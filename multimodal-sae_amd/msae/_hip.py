"""ctypes binding of libmsae_hip.so (C ABI declared in include/msae.h).

PyTorch is used only for device memory and streams: every call passes raw device pointers and the
current HIP stream.  There is NO CPU fallback: if the shared library is missing, or a tensor is not
on a HIP device, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import torch

_LIB_DIR = Path(__file__).resolve().parent / "_lib"
LIB_PATH = _LIB_DIR / "libmsae_hip.so"

c_void_p, c_int, c_float, c_size_t, c_int64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                               ctypes.c_size_t, ctypes.c_int64)

ABI_VERSION = 4


class MsaeOptions(ctypes.Structure):
    """struct msae_options of include/msae.h: the per-call options of the fused encoder."""
    _fields_ = [("size", ctypes.c_uint32), ("coarse_mode", ctypes.c_int32), ("guard_z", ctypes.c_float),
                ("status_detail", ctypes.c_int32), ("profile", ctypes.c_void_p), ("exact", ctypes.c_int32),
                ("dither", ctypes.c_int32), ("rows_rescored", ctypes.c_void_p), ("dither_seed", ctypes.c_uint64),
                ("certified", ctypes.c_int32), ("reserved2", ctypes.c_int32), ("certified_operands", ctypes.c_void_p)]


c_opts_p = ctypes.POINTER(MsaeOptions)

# name -> (restype, argtypes); mirrors include/msae.h one to one
PROTOTYPES = {
    "msae_options_init": (None, [c_opts_p]),
    "msae_abi_version": (c_int, []),
    "msae_error_string": (ctypes.c_char_p, [c_int]),
    "msae_target_arch": (ctypes.c_char_p, []),
    "msae_pre_acts_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                  c_int, c_void_p, c_void_p]),
    "msae_topk_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "msae_topk_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                              c_void_p]),
    "msae_encoder_prepared_bytes": (c_size_t, [c_int, c_int]),
    "msae_encoder_prepare": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "msae_encoder_prepare_opts": (c_int, [c_void_p, c_int, c_int, c_void_p, c_opts_p, c_void_p]),
    "msae_encoder_certified_bytes": (c_size_t, [c_int, c_int]),
    "msae_encoder_prepare_certified": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "msae_encoder_refresh": (c_int, [c_void_p, c_int, c_int, c_void_p, c_opts_p, c_void_p]),
    "msae_encoder_refresh_for": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_opts_p, c_void_p]),
    "msae_encode_topk_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_opts_p]),
    "msae_encode_topk": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_size_t, c_opts_p, c_void_p]),
    "msae_encode_topk_i64": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                     c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_opts_p, c_void_p]),
    "msae_encode_topk_rows_ws_bytes": (c_size_t, [c_int, c_int]),
    "msae_encode_topk_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "msae_shard_record_bytes": (c_size_t, [c_int]),
    "msae_shard_candidates": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_opts_p, c_void_p]),
    "msae_rescore_candidates_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "msae_rescore_candidates": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_int, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_size_t, c_opts_p, c_void_p]),
    "msae_decode_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_void_p, c_void_p, c_void_p]),
    "msae_decode_i64_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p]),
    "msae_decode_bwd_acts_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p]),
    "msae_decode_bwd_wdec_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "msae_decode_bwd_wdec_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "msae_sparsify_count": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int,
                                    c_void_p, c_void_p]),
    "msae_sparsify_write": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int,
                                    c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "msae_merge_topk": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "msae_compact_flags": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "msae_merge_topk_masked": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "msae_unit_norm_rows_f32": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p]),
    "msae_grad_sumsq_f32": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "msae_sum_f32": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "msae_weighted_row_sum_ws_bytes": (c_size_t, [c_int, c_int]),
    "msae_weighted_row_sum_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "msae_adam_rows_fused_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float,
                                         c_int, c_float, c_float, c_float, c_float, c_int, c_float, c_void_p, c_int,
                                         c_opts_p, c_void_p]),
    "msae_adam_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float,
                                   c_int, c_float, c_float, c_float, c_float, c_int, c_void_p]),
    "msae_profile_create": (c_int, [c_int, ctypes.POINTER(c_void_p)]),
    "msae_profile_read": (c_int, [c_void_p, c_void_p, c_void_p]),
    "msae_profile_destroy": (c_int, [c_void_p]),
}

DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}

_lib = None


class MsaeLibraryMissing(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load libmsae_hip.so; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("MSAE_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise MsaeLibraryMissing(
            f"{path} not found: build it with `python __graft_entry__.py` "
            "(multimodal-sae_amd/csrc/build.sh). There is no CPU fallback for the SAE kernels.")
    lib = ctypes.CDLL(str(path))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.msae_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libmsae_hip.so ABI version {lib.msae_abi_version()} != {ABI_VERSION}: rebuild it "
                           "(python __graft_entry__.py)")
    _lib = lib
    return lib


class MsaeNotImplemented(RuntimeError):
    """MSAE_ENOTIMPL: the shape lies outside what this entry point supports (callers with an alternative take it)."""


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().msae_error_string(code).decode()
        raise (MsaeNotImplemented if code == -4 else RuntimeError)(f"{what} failed: {msg} (code {code})")


def ptr(t: torch.Tensor | None):
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream_of(t: torch.Tensor):
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_device(*tensors: torch.Tensor | None) -> torch.device:
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "msae kernels run on MI355X only: got a tensor on "
                f"{t.device}. Move the Sae and its inputs to a HIP device (there is no CPU path).")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    return dev

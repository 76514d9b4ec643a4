"""ctypes front-end for the CPU oracle (oracle/sae_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  Nothing under ``multimodal-sae_amd/`` imports it: the product path fails loudly when
the HIP library is missing instead of falling back to this code.

Two restatements of the reference hot path live here:

* the C library (``libmsae_oracle.so``): arithmetic *defined* as one ascending-k f32 fmaf chain,
  the bit-exact comparison target for the HIP kernels (see sae_oracle.c header);
* :class:`RefPort`: the reference algorithm restated with the same torch-CPU operators the
  reference itself calls (``F.linear`` + ``relu`` -> ``topk`` -> eager ``scatter_`` + dense
  matmul; sae_auto_interp/sae/sae.py:172-191, sae/utils.py:108-111).  It is what
  ``bench.py`` times as ``cpu_baseline`` (kind "port") and what the golden fixtures in
  ``tests/golden`` (generated from the reference itself) are compared against.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libmsae_oracle.so"

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)


def build(force: bool = False) -> Path:
    """Compile the C oracle with gcc (oracle/Makefile)."""
    src = _HERE / "sae_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE), "-B", "libmsae_oracle.so"], check=True,
                       capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(_LIB_PATH))
        _lib.msae_oracle_pre_acts.restype = ctypes.c_int
        _lib.msae_oracle_topk.restype = ctypes.c_int
        _lib.msae_oracle_encode_topk.restype = ctypes.c_int
        _lib.msae_oracle_decode.restype = ctypes.c_int
        _lib.msae_oracle_decode_bwd_acts.restype = ctypes.c_int
        _lib.msae_oracle_sparsify.restype = ctypes.c_int64
        _lib.msae_oracle_num_threads.restype = ctypes.c_int
    return _lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a: np.ndarray | None, typ):
    return None if a is None else a.ctypes.data_as(typ)


def num_threads() -> int:
    return int(lib().msae_oracle_num_threads())


def pre_acts(x, W_enc, b_enc, b_dec, relu: bool = True) -> np.ndarray:
    """relu((x - b_dec) @ W_enc.T + b_enc), ascending-k fmaf chain.  sae.py:172-177"""
    x, W_enc = _f32(x), _f32(W_enc)
    T, d = x.shape
    N = W_enc.shape[0]
    assert W_enc.shape[1] == d
    b_enc = None if b_enc is None else _f32(b_enc)
    b_dec = None if b_dec is None else _f32(b_dec)
    out = np.empty((T, N), dtype=np.float32)
    rc = lib().msae_oracle_pre_acts(_p(x, _f32p), _p(W_enc, _f32p), _p(b_enc, _f32p),
                                    _p(b_dec, _f32p), T, d, N, int(relu), _p(out, _f32p))
    assert rc == 0, rc
    return out


def topk(latents, k: int):
    """Canonical top-k (value desc, index asc).  sae.py:179-181"""
    latents = _f32(latents)
    T, N = latents.shape
    vals = np.empty((T, k), dtype=np.float32)
    idx = np.empty((T, k), dtype=np.int32)
    rc = lib().msae_oracle_topk(_p(latents, _f32p), T, N, k, _p(vals, _f32p), _p(idx, _i32p))
    assert rc == 0, rc
    return vals, idx


def encode_topk(x, W_enc, b_enc, b_dec, k: int, set_feature: int = -1, set_value: float = 0.0,
                zero_feature: int = -1):
    """Fused pre_acts -> (optional hook edit) -> canonical top-k.  sae.py:183-185"""
    x, W_enc = _f32(x), _f32(W_enc)
    T, d = x.shape
    N = W_enc.shape[0]
    b_enc = None if b_enc is None else _f32(b_enc)
    b_dec = None if b_dec is None else _f32(b_dec)
    vals = np.empty((T, k), dtype=np.float32)
    idx = np.empty((T, k), dtype=np.int32)
    rc = lib().msae_oracle_encode_topk(_p(x, _f32p), _p(W_enc, _f32p), _p(b_enc, _f32p),
                                       _p(b_dec, _f32p), T, d, N, k, int(set_feature),
                                       ctypes.c_float(set_value), int(zero_feature),
                                       _p(vals, _f32p), _p(idx, _i32p))
    assert rc == 0, rc
    return vals, idx


def decode(idx, acts, W_dec, b_dec) -> np.ndarray:
    """sum_j acts[:, j] * W_dec[idx[:, j]] + b_dec, j-ordered fmaf chain.  sae.py:187-191"""
    idx, acts, W_dec = _i32(idx), _f32(acts), _f32(W_dec)
    A, k = idx.shape
    N, d = W_dec.shape
    b_dec = None if b_dec is None else _f32(b_dec)
    out = np.empty((A, d), dtype=np.float32)
    rc = lib().msae_oracle_decode(_p(idx, _i32p), _p(acts, _f32p), _p(W_dec, _f32p),
                                  _p(b_dec, _f32p), A, k, N, d, _p(out, _f32p))
    if rc == -3:
        raise IndexError("feature index out of range (kernels.py:276 device_assert)")
    assert rc == 0, rc
    return out


def decode_bwd_acts(idx, grad_out, W_dec) -> np.ndarray:
    """d loss / d top_acts.  kernels.py:421-425"""
    idx, grad_out, W_dec = _i32(idx), _f32(grad_out), _f32(W_dec)
    A, k = idx.shape
    N, d = W_dec.shape
    g = np.empty((A, k), dtype=np.float32)
    rc = lib().msae_oracle_decode_bwd_acts(_p(idx, _i32p), _p(grad_out, _f32p), _p(W_dec, _f32p),
                                           A, k, N, d, _p(g, _f32p))
    assert rc == 0, rc
    return g


def sparsify(vals, idx, B: int, S: int, row_base: int = 0, thresh: float = 1e-5,
             filter_bitmap=None):
    """(vals, idx)[B*S, k] -> COO (locations[nnz,3] int64, activations[nnz]).  cache.py:42-92"""
    vals, idx = _f32(vals).reshape(B * S, -1), _i32(idx).reshape(B * S, -1)
    k = vals.shape[1]
    cap = B * S * k
    loc = np.empty((cap, 3), dtype=np.int64)
    act = np.empty((cap,), dtype=np.float32)
    fb = None if filter_bitmap is None else np.ascontiguousarray(filter_bitmap, dtype=np.uint8)
    nnz = lib().msae_oracle_sparsify(_p(vals, _f32p), _p(idx, _i32p), B, S, k,
                                     ctypes.c_int64(row_base), ctypes.c_float(thresh),
                                     _p(fb, _u8p), ctypes.c_int64(cap), _p(loc, _i64p),
                                     _p(act, _f32p))
    return loc[:nnz].copy(), act[:nnz].copy()


class RefPort:
    """The reference algorithm on torch-CPU operators (what the reference runs with
    SAE_DISABLE_TRITON=1).  fp32 throughout, like sae.py:140,174."""

    def __init__(self, W_enc, b_enc, W_dec, b_dec, k: int):
        import torch

        self.torch = torch
        as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a))
        self.W_enc, self.b_enc = as_t(W_enc).float(), as_t(b_enc).float()
        self.W_dec, self.b_dec = as_t(W_dec).float(), as_t(b_dec).float()
        self.k = k

    def pre_acts(self, x):
        torch = self.torch
        sae_in = x.to(torch.float32) - self.b_dec  # sae.py:174
        return torch.relu(torch.nn.functional.linear(sae_in, self.W_enc, self.b_enc))  # :175-177

    def select_topk(self, latents):
        return latents.topk(self.k, sorted=False)  # sae.py:181

    def decode(self, top_acts, top_indices):
        # eager_decode, sae/utils.py:108-111 (dense scatter + matmul), then + b_dec sae.py:191
        buf = top_acts.new_zeros(top_acts.shape[:-1] + (self.W_dec.shape[0],))
        acts = buf.scatter_(dim=-1, index=top_indices, src=top_acts)
        return acts @ self.W_dec + self.b_dec

    def forward(self, x):
        top_acts, top_idx = self.select_topk(self.pre_acts(x))
        return self.decode(top_acts, top_idx), top_acts, top_idx

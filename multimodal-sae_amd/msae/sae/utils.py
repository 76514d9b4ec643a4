"""The decoder plug-in seam of the reference (sae_auto_interp/sae/utils.py:107-129; train/sae/sae/utils.py:107-129).

The reference picks `decoder_impl` at import time between `eager_decode` and `triton_decode` (env SAE_DISABLE_TRITON=1 ->
eager).  Both names exist here with the reference's calling convention -- W_dec is passed TRANSPOSED (`W_dec.mT`, a [d, N]
view of the [N, d] parameter), as Sae.decode does at sae.py:190 and train/sae/tests/test_decode.py:17-18 do:

  * `triton_decode` = the HIP k-sparse gather-matmul (`torch.ops.msae.decode`, csrc/decode.hip), differentiable as
    TritonDecoder is (kernels.py:403-429);
  * `eager_decode`  = the reference's dense restatement (utils.py:108-111: scatter into zeros [A, N], then a dense matmul) on the
    exact f32 MFMA kernel -- 2 A N d FLOP, kept for the reference's own test (eager == sparse) and as the same escape hatch
    SAE_DISABLE_TRITON=1 is there; differentiable.

`Sae.decode` goes through the module-level `decoder_impl` like the reference's (sae.py:190): rebinding it reroutes every decode.
"""
from __future__ import annotations

import os

import torch
from torch import Tensor

from .. import ops


def hip_decode(top_indices: Tensor, top_acts: Tensor, W_dec: Tensor) -> Tensor:
    """decoder_impl(top_indices, top_acts, W_dec.mT) -> [A, d] (no bias), differentiable."""
    return ops.decode(top_indices, top_acts, W_dec.mT, None)


class _EagerDecode(torch.autograd.Function):
    """zeros[A, N].scatter_(-1, idx, acts) @ W_dec.mT with dense GEMMs on the exact f32 kernel (forward and backward)."""

    @staticmethod
    def forward(ctx, top_indices: Tensor, top_acts: Tensor, W_dec: Tensor) -> Tensor:
        buf = top_acts.new_zeros(top_acts.shape[:-1] + (W_dec.shape[-1],), dtype=torch.float32)
        dense = buf.scatter_(dim=-1, index=top_indices, src=top_acts.float())
        ctx.save_for_backward(top_indices, dense, W_dec)
        return ops._dense_gemm_nt(dense, W_dec.float())          # [A, N] @ [d, N]^T

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        top_indices, dense, W_dec = ctx.saved_tensors
        g = grad_out.float().contiguous()
        g_acts = g_W = None
        if ctx.needs_input_grad[1]:
            g_acts = ops._dense_gemm_nt(g, W_dec.float().mT).gather(-1, top_indices)      # [A, d] @ [N, d]^T, the k picked
        if ctx.needs_input_grad[2]:
            g_W = ops._dense_gemm_nt(g.t(), dense.t()).to(W_dec.dtype)                    # [d, A] @ [N, A]^T
        return None, g_acts, g_W


def eager_decode(top_indices: Tensor, top_acts: Tensor, W_dec: Tensor) -> Tensor:
    """The reference's fallback decoder (utils.py:108-111), same arguments as `triton_decode`."""
    from .. import _hip

    _hip.require_device(top_indices, top_acts, W_dec)
    lead = top_acts.shape[:-1]
    out = _EagerDecode.apply(top_indices.reshape(-1, top_indices.shape[-1]), top_acts.reshape(-1, top_acts.shape[-1]), W_dec)
    return out.reshape(*lead, out.shape[-1])


# names the reference exports from this module; the selection rule is the reference's (utils.py:119-129)
triton_decode = hip_decode
decoder_impl = eager_decode if os.environ.get("SAE_DISABLE_TRITON") == "1" else triton_decode

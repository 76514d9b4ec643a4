"""The decoder plug-in seam of the reference (sae_auto_interp/sae/utils.py:107-129).

The reference picks `decoder_impl` at import time between `eager_decode` and `triton_decode`
(env SAE_DISABLE_TRITON).  Here the seam has one implementation: the HIP gather-matmul
(`torch.ops.msae.decode`).  The function keeps the reference's calling convention -- W_dec is
passed TRANSPOSED (`W_dec.mT`, a [d, N] view of the [N, d] parameter), as Sae.decode does at
sae.py:190 and train/sae/tests/test_decode.py:17-18 do.
"""
from __future__ import annotations

from torch import Tensor

from .. import ops


def hip_decode(top_indices: Tensor, top_acts: Tensor, W_dec: Tensor) -> Tensor:
    """decoder_impl(top_indices, top_acts, W_dec.mT) -> [A, d] (no bias), differentiable."""
    return ops.decode(top_indices, top_acts, W_dec.mT, None)


# names the reference exports from this module
triton_decode = hip_decode
decoder_impl = hip_decode

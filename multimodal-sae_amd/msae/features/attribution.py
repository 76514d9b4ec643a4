"""Batched grad x act feature attribution.

The reference scores one feature at a time: for every feature f it re-runs the LLM with the SAE
reconstruction spliced in and f's latent zeroed, back-propagates the metric, and sums
`(clean - corrupted) * corrupted.grad` over d (features/patching/attribution.py:133-183,
features/patching/utils.py:21-79) -- two forwards and one backward of the LLM per feature.

With the reconstruction linear in the latents, `clean - corrupted = act_f * W_dec[f]`, so to first
order every ACTIVE feature of every token is scored from ONE backward pass:

    score[t, j] = top_acts[t, j] * < dmetric/dsae_out[t, :], W_dec[top_indices[t, j], :] >

which is exactly the dense-dense-sparse-out primitive of the decoder backward
(`msae_decode_bwd_acts_f32`, reference sae/kernels.py:287-400).  Inactive features score 0, as in
the reference (zeroing an inactive latent changes nothing).  The gradient is taken at the clean run
rather than at each corrupted run -- the standard attribution-patching linearisation.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from .. import ops
from ..sae import Sae


@torch.no_grad()
def grad_times_act(sae: Sae, top_acts: Tensor, top_indices: Tensor, grad_sae_out: Tensor) -> Tensor:
    """[..., k] scores of the active features given d(metric)/d(sae_out) [..., d]."""
    k = top_acts.shape[-1]
    acts = top_acts.reshape(-1, k)
    g_acts, _ = ops.decode_bwd(top_indices.reshape(-1, k), acts, sae.W_dec,
                               grad_sae_out.reshape(-1, grad_sae_out.shape[-1]).float().contiguous(),
                               True, False)
    return (acts.float() * g_acts).view(top_acts.shape)


def feature_scores(sae: Sae, top_acts: Tensor, top_indices: Tensor, grad_sae_out: Tensor,
                   reduce_tokens: bool = True) -> Tuple[Tensor, Tensor]:
    """Aggregate the per-(token, feature) scores per feature id.  -> (feature ids, summed score)."""
    scores = grad_times_act(sae, top_acts, top_indices, grad_sae_out)
    if not reduce_tokens:
        return top_indices, scores
    flat_idx, flat_s = top_indices.reshape(-1), scores.reshape(-1)
    total = torch.zeros(sae.num_latents, dtype=torch.float32, device=flat_s.device)
    total.index_add_(0, flat_idx, flat_s)
    feats = torch.nonzero(total).flatten()
    return feats, total[feats]

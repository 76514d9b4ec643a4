"""Forward-hook bodies that splice the SAE reconstruction into the LLM, on the fused path.

Reference hooks replaced (behaviour identical, dense `[T, N]` latents never materialised):
  * steering     features/steering.py:102-128 (dup. tools/model_steering.py:62-79):
        latents = pre_acts(h); if S != 1: latents[:, :, f] = clamp; topk; decode(top[0]) -> fp16
  * attribution  features/patching/utils.py:33-58:
        latents = pre_acts(h.flatten(0,1)); latents[:, off] *= 0; topk; decode -> fp16 view(B,S,d)
The latent edits are arguments of the fused encode kernel (`set_feature`, `zero_feature`).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch
from torch import Tensor

from ..sae import Sae


def sae_reconstruct(sae, hidden: Tensor, *, set_feature: int = -1, set_value: float = 0.0,
                    zero_feature: int = -1, out_dtype: Optional[torch.dtype] = None,
                    differentiable: Optional[bool] = None) -> Tensor:
    """[..., d] hidden states -> SAE reconstruction of the same shape.  `sae` is an `Sae` module or a
    feature-sharded engine (msae.parallel.ShardedSae over an N/G slice of the encoder per rank, SURVEY 8f rank 4:
    "N-sharded across 8 GPUs"; every rank must hold the same hidden states): same edits, by GLOBAL feature id, same
    bits out."""
    flat = hidden.reshape(-1, hidden.shape[-1])
    if isinstance(sae, Sae):
        top = sae.encode(flat, set_feature=set_feature, set_value=set_value, zero_feature=zero_feature,
                         differentiable=differentiable)
        out = sae.decode(top.top_acts, top.top_indices)
    else:   # engine interface: encode -> (acts, global ids, status), decode(acts, ids)
        acts, idx, _ = sae.encode(flat.contiguous(), set_feature=set_feature, set_value=set_value,
                                  zero_feature=zero_feature)
        out = sae.decode(acts, idx)
    return out.to(out_dtype or hidden.dtype).view(hidden.shape)


def _replace_first(outputs, new0):
    if isinstance(outputs, tuple):
        return (new0,) + tuple(outputs[1:])
    return new0


def clamp_features_max(sae, feature: int, hooked_module: torch.nn.Module, k: float = 10):
    """Register the steering hook (steering.py:102-128): on prefill (S != 1) the feature's latent
    is set to `k` before TopK; every call replaces the layer output by the fp16 reconstruction.
    `sae`: an `Sae`, or a `ShardedSae` engine (the S = 1 decode steps then stream N/G rows of the encoder per rank
    and exchange 8 k_loc bytes; the decode of so few tokens is local on every rank)."""

    def hook(module, _, outputs):
        h = outputs[0] if isinstance(outputs, tuple) else outputs
        prefill = h.shape[1] != 1
        out = sae_reconstruct(sae, h[0], set_feature=feature if prefill else -1, set_value=float(k),
                              out_dtype=torch.float16).unsqueeze(0)
        return _replace_first(outputs, out)

    return [hooked_module.register_forward_hook(hook)]


def attribution_sae_hook(sae_dict: Dict[str, Sae], module_to_name: Dict[torch.nn.Module, str],
                         cache: Dict[str, Tensor], off_features: Optional[int] = None) -> Callable:
    """Hook body of get_model_forward_cache_with_sae (patching/utils.py:33-58)."""

    def hook(module, inputs, outputs):
        h = outputs[0] if isinstance(outputs, tuple) else outputs
        name = module_to_name[module]
        out = sae_reconstruct(sae_dict[name], h, zero_feature=-1 if off_features is None else off_features,
                              out_dtype=torch.float16, differentiable=torch.is_grad_enabled())
        cache[name] = out
        return _replace_first(outputs, out)

    return hook

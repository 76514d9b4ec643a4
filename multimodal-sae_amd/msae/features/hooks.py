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


def sae_reconstruct(sae: Sae, hidden: Tensor, *, set_feature: int = -1, set_value: float = 0.0,
                    zero_feature: int = -1, out_dtype: Optional[torch.dtype] = None) -> Tensor:
    """[..., d] hidden states -> SAE reconstruction of the same shape."""
    flat = hidden.reshape(-1, hidden.shape[-1])
    top = sae.encode(flat, set_feature=set_feature, set_value=set_value, zero_feature=zero_feature)
    out = sae.decode(top.top_acts, top.top_indices)
    return out.to(out_dtype or hidden.dtype).view(hidden.shape)


def _replace_first(outputs, new0):
    if isinstance(outputs, tuple):
        return (new0,) + tuple(outputs[1:])
    return new0


def clamp_features_max(sae: Sae, feature: int, hooked_module: torch.nn.Module, k: float = 10):
    """Register the steering hook (steering.py:102-128): on prefill (S != 1) the feature's latent
    is set to `k` before TopK; every call replaces the layer output by the fp16 reconstruction."""

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
                              out_dtype=torch.float16)
        cache[name] = out
        return _replace_first(outputs, out)

    return hook

"""Attribution-patching helpers -- same names, arguments and results as the reference's
sae_auto_interp/features/patching/utils.py:10-79, with the SAE splice running on the fused HIP path:

    reference hook (utils.py:33-58)                       here
    latents = sae.pre_acts(h.flatten(0, 1))               sae.encode(h.flatten(0, 1), zero_feature=off)
    latents = latents * mask(off_features)                  (the mask is an argument of the fused kernel;
    top = sae.select_topk(latents)                           the dense [T, N] latents are never built)
    sae_out = sae.decode(*top).to(fp16).view(B, S, d)      sae.decode(*top).to(fp16).view(B, S, d)

Autograd flows exactly where it does in the reference: from the spliced fp16 reconstruction back
through decode (d acts, d W_dec, d b_dec) and the selected latents of the encoder (d hidden, d W_enc,
d b_enc, d b_dec), so `tensor.retain_grad()` on the cached reconstruction and `metric.backward()`
(attribution.py:165-172) work unchanged.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Tuple

import torch
from torch import Tensor

from ...sae import Sae


def get_logit_diff(logits: Tensor, answer_token_indices: Tensor) -> Tensor:
    """mean(logit[correct] - logit[baseline]) at the final position (utils.py:10-19)."""
    if logits.dim() == 3:
        logits = logits[:, -1, :]
    correct = logits.gather(1, answer_token_indices[:, 0].unsqueeze(1))
    incorrect = logits.gather(1, answer_token_indices[:, 1].unsqueeze(1))
    return (correct - incorrect).mean()


def sae_splice_hook(sae_dict: Dict[str, Sae], module_to_name: Dict[torch.nn.Module, str],
                    cache: Dict[str, Tensor], off_features: Optional[int] = None,
                    keep_latents: Optional[Dict[str, Tuple[Tensor, Tensor]]] = None, extra_k: int = 0) -> Callable:
    """Forward-hook body of get_model_forward_cache_with_sae (utils.py:33-58).  `keep_latents`
    (optional) receives the (top_acts, top_indices) of every hooked module -- the batched
    attribution needs them; `extra_k` asks the encoder for that many latents beyond k (the
    reconstruction still uses the first k)."""

    def hook(module, inputs, outputs):
        unpacked = list(outputs) if isinstance(outputs, tuple) else [outputs]
        name = module_to_name[module]
        sae = sae_dict[name]
        bs, seq_len, dim = unpacked[0].shape
        flat = unpacked[0].flatten(0, 1)
        zero = -1 if off_features is None else int(off_features)
        if extra_k:
            from ... import ops

            with torch.no_grad():
                va, ia, _ = ops.encode_topk(flat, sae.encoder.weight, sae.encoder.bias, sae.b_dec,
                                            sae._prepared_weights(), sae.cfg.k + extra_k, -1, 0.0, zero)
            if keep_latents is not None:
                keep_latents[name] = (va, ia)
        # the reference's graph is differentiable wherever autograd is on (the SAE's parameters require grad even under
        # a frozen LLM): keep that, rather than Sae.encode's default of following x.requires_grad
        top = sae.encode(flat, zero_feature=zero, differentiable=torch.is_grad_enabled())
        if keep_latents is not None and not extra_k:
            keep_latents[name] = (top.top_acts.detach(), top.top_indices)
        sae_out = sae.decode(top.top_acts, top.top_indices).to(torch.float16).view(bs, seq_len, dim)
        cache[name] = sae_out
        if isinstance(outputs, tuple):
            return tuple([sae_out] + unpacked[1:])
        return sae_out

    return hook


def get_model_forward_cache_with_sae(model: torch.nn.Module, inputs: Dict[str, Any], sae_dict: Dict[str, Sae],
                                     module_to_name: Dict[torch.nn.Module, str], off_features: int = None,
                                     keep_latents: Optional[dict] = None, extra_k: int = 0):
    """Run the model with every hooked module's output replaced by its SAE reconstruction.
    -> (logits, {module name: fp16 reconstruction [B, S, d]})   (utils.py:21-71)."""
    cache: Dict[str, Tensor] = {}
    hook = sae_splice_hook(sae_dict, module_to_name, cache, off_features, keep_latents, extra_k)
    handles = [mod.register_forward_hook(hook) for mod in module_to_name.keys()]
    try:
        outputs = model(**inputs)
        logits = outputs["logits"]
    finally:
        for h in handles:
            h.remove()
    return logits, cache


def get_model_backward_cache_with_sae(logits: Tensor, metrics: Callable[[Tensor], Tensor]) -> Tensor:
    """metric(logits).backward() (utils.py:74-79)."""
    values = metrics(logits)
    values.backward()
    return values

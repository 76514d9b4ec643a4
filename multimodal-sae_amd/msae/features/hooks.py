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


class _DecodeStepGraph:
    """The S = 1 step of the steering hook -- encode, decode, cast: ~8 kernel launches through two custom ops -- captured ONCE
    into a HIP graph and replayed (round-5 verdict, item 9).  A generation step of an 8B model is host-bound; the hook's two
    op dispatches + ctypes calls (~17 us each) sit on that critical path up to 512 times per feature
    (features/steering.py:86).  A replay is one copy into the captured input and one graph launch.

    The library's entry points allocate nothing and never synchronise, which is what makes them capturable
    (tests/test_gpu_parity.py::test_encode_and_decode_replay_from_a_hip_graph).  The graph holds raw pointers, so it is keyed on
    everything it captured -- the prepared operand buffer, the parameters' storage and versions, the workspace epoch
    (ops.release_workspaces) -- and re-captured when any of them changes; a capture that fails once (an exotic build, a
    stream in capture already) switches the hook to the eager path for good.  The replayed dither seed is the captured one:
    fine for inputs that do not know it (include/msae.h)."""

    def __init__(self):
        self.key = None
        self.graph = None
        self.x = self.out = None
        self.keep = None
        self.failed = False

    @staticmethod
    def _key(sae, h: Tensor):
        from .. import ops

        w, bias, wd, bd = sae.encoder.weight, sae.encoder.bias, sae.W_dec, sae.b_dec
        prep = sae._prepared_weights()
        return (h.dtype, h.device, tuple(h.shape), ops.workspace_epoch(), 0 if prep is None else prep.data_ptr(),
                w.data_ptr(), w._version, bias.data_ptr(), bias._version, wd.data_ptr(), wd._version, bd.data_ptr(), bd._version,
                ops._defaults.coarse, ops._defaults.guard_z, ops._defaults.exact, getattr(ops._defaults, "certified", False))

    def __call__(self, sae, h: Tensor) -> Optional[Tensor]:
        """h [1, d] -> fp16 reconstruction [1, d] (a fresh tensor), or None: take the eager path."""
        if self.failed or torch.cuda.is_current_stream_capturing():
            return None
        key = self._key(sae, h)
        if key != self.key:
            try:
                self._capture(sae, h, key)
            except Exception:  # noqa: BLE001 -- any capture problem: the eager path is always right
                self.failed, self.graph, self.key = True, None, None
                return None
        self.x.copy_(h)
        self.graph.replay()
        return self.out.clone()

    def _capture(self, sae, h: Tensor, key) -> None:
        body = lambda t: sae_reconstruct(sae, t, out_dtype=torch.float16)
        x = h.clone()
        with torch.cuda.device(h.device):            # (a hook on a layer of another device than the current one)
            side = torch.cuda.Stream(device=h.device)
            side.wait_stream(torch.cuda.current_stream(h.device))
            with torch.cuda.stream(side):
                for _ in range(2):                   # workspaces and the prepared operands exist before the capture
                    body(x)
            torch.cuda.current_stream(h.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                out = body(x)
        self.graph, self.x, self.out, self.key = g, x, out, key
        # what the graph captured by address stays alive as long as the graph does: the operand buffer and the side stream's
        # scratch buffer (ops._workspace keeps only the most recently used streams' buffers)
        from .. import ops

        self.keep = (sae._prepared_weights(), ops._WS.get((h.device, side.cuda_stream)), side)


def clamp_features_max(sae, feature: int, hooked_module: torch.nn.Module, k: float = 10, graph_step: Optional[bool] = None):
    """Register the steering hook (steering.py:102-128): on prefill (S != 1) the feature's latent
    is set to `k` before TopK; every call replaces the layer output by the fp16 reconstruction.
    `sae`: an `Sae`, or a `ShardedSae` engine (the S = 1 decode steps then stream N/G rows of the encoder per rank
    and exchange 8 k_loc bytes; the decode of so few tokens is local on every rank).
    `graph_step`: replay the S = 1 step from a captured HIP graph (_DecodeStepGraph); default: on for a single-GPU `Sae`
    under no_grad unless MSAE_HOOK_GRAPH=0."""
    import os

    if graph_step is None:
        graph_step = os.environ.get("MSAE_HOOK_GRAPH", "1") not in ("0", "")
    step_graph = _DecodeStepGraph() if (graph_step and isinstance(sae, Sae)) else None

    def hook(module, _, outputs):
        h = outputs[0] if isinstance(outputs, tuple) else outputs
        prefill = h.shape[1] != 1
        if (step_graph is not None and not prefill and h.shape[0] == 1 and h.is_cuda and not torch.is_grad_enabled()
                and not sae.training):
            out = step_graph(sae, h[0])
            if out is not None:
                return _replace_first(outputs, out.unsqueeze(0))
        out = sae_reconstruct(sae, h[0], set_feature=feature if prefill else -1, set_value=float(k),
                              out_dtype=torch.float16).unsqueeze(0)
        return _replace_first(outputs, out)

    handle = hooked_module.register_forward_hook(hook)
    handle.step_graph = step_graph          # (diagnostics / tests: the captured S = 1 step, or None)
    return [handle]


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

"""Drop-in `Sae` module backed by the gfx950 HIP kernels.

Same constructor, attributes, state-dict keys (`encoder.weight [N,d]`, `encoder.bias [N]`,
`W_dec [N,d]`, `b_dec [d]`), checkpoint layout (`<dir>/cfg.json` + `<dir>/sae.safetensors`) and
methods as the reference module (sae_auto_interp/sae/sae.py:44-271), so the hooks in
features/cache.py, features/steering.py, features/patching/utils.py and tools/*.py keep working
unchanged.  What differs is underneath:

* `pre_acts`      -> torch.ops.msae.pre_acts   (exact f32 MFMA GEMM, dense output for legacy callers)
* `select_topk`   -> torch.ops.msae.topk       (canonical order: value desc, index asc)
* `encode`        -> torch.ops.msae.encode_topk (fused: the dense [T, N] latents never reach HBM)
* `decode`        -> torch.ops.msae.decode     (coalesced gather-matmul over W_dec rows, autograd)

SAE math is f32 whatever dtype the LLM hands over, as in the reference (sae.py:140,174).
There is no CPU implementation: calling a compute method with CPU tensors raises.
"""
from __future__ import annotations

import json
import os
import re
from fnmatch import fnmatch
from pathlib import Path
from typing import NamedTuple, Optional, Union

import torch
from torch import Tensor, nn

from .. import ops
from . import utils as _seam
from .config import SaeConfig


class EncoderOutput(NamedTuple):
    top_acts: Tensor
    """Activations of the top-k latents."""
    top_indices: Tensor
    """Indices of the top-k features."""


class ForwardOutput(NamedTuple):
    sae_out: Tensor
    latent_acts: Tensor
    latent_indices: Tensor
    fvu: Tensor
    auxk_loss: Tensor
    multi_topk_fvu: Tensor


class _SqErr(torch.autograd.Function):
    """sum((pred - target)^2) -> (scalar, pred - target) with ONE elementwise kernel and one reduction forward and one
    elementwise kernel backward, where the reference's `e = pred - x; e.pow(2).sum()` (sae.py:201,227) costs three
    T x d sweeps forward and three backward.  `target` is a constant (the LLM's activations)."""

    @staticmethod
    def forward(ctx, pred, target):
        e = pred - target
        ctx.save_for_backward(e)
        ctx.mark_non_differentiable(e)
        return torch.linalg.vector_norm(e).square(), e

    @staticmethod
    def backward(ctx, g, _ge):
        e, = ctx.saved_tensors
        return e * (2.0 * g), None


def _total_variance(x: Tensor) -> Tensor:
    """(x - x.mean(0)).pow(2).sum() (sae.py:202) as one column-variance kernel + a d-element sum."""
    return torch.var(x, dim=0, correction=0).sum() * x.shape[0]


def _natural_key(s: str):
    return [int(p) if p.isdigit() else p for p in re.split(r"(\d+)", s)]


class Sae(nn.Module):
    _warned_detached = False     # the one-time notice of Sae.encode (see its docstring)

    def __init__(self, d_in: int, cfg: SaeConfig, device: Union[str, torch.device] = "cpu",
                 dtype: Union[torch.dtype, None] = None, *, decoder: bool = True):
        super().__init__()
        self.cfg = cfg
        self.d_in = d_in
        self.num_latents = cfg.num_latents or d_in * cfg.expansion_factor

        self.encoder = nn.Linear(d_in, self.num_latents, device=device, dtype=dtype)
        self.encoder.bias.data.zero_()
        self.W_dec = nn.Parameter(self.encoder.weight.data.clone()) if decoder else None
        if decoder and self.cfg.normalize_decoder:
            self.set_decoder_norm_to_unit_norm()
        self.b_dec = nn.Parameter(torch.zeros(d_in, dtype=dtype, device=device))
        self._prepared: Optional[Tensor] = None
        self._prepared_key = None

    # ---- checkpoint I/O (sae.py:69-162) -----------------------------------------------------------
    @staticmethod
    def load_many(name: str, local: bool = False, layers: Union[list, None] = None,
                  device: Union[str, torch.device] = "cpu", *, decoder: bool = True,
                  pattern: Union[str, None] = None) -> dict:
        pattern = pattern + "/*" if pattern is not None else None
        if local:
            repo_path = Path(name)
        else:
            from huggingface_hub import snapshot_download

            repo_path = Path(snapshot_download(name, allow_patterns=pattern))
        if layers is not None:
            return {layer: Sae.load_from_disk(repo_path / layer, device=device, decoder=decoder)
                    for layer in sorted(layers, key=_natural_key)}
        dirs = [f for f in repo_path.iterdir()
                if f.is_dir() and (pattern is None or fnmatch(f.name, pattern))]
        return {f.name: Sae.load_from_disk(f, device=device, decoder=decoder)
                for f in sorted(dirs, key=lambda f: _natural_key(f.name))}

    @staticmethod
    def load_from_hub(name: str, hookpoint: Union[str, None] = None,
                      device: Union[str, torch.device] = "cpu", *, decoder: bool = True) -> "Sae":
        from huggingface_hub import snapshot_download

        repo_path = Path(snapshot_download(
            name, allow_patterns=f"{hookpoint}/*" if hookpoint is not None else None))
        if hookpoint is not None:
            repo_path = repo_path / hookpoint
        elif not repo_path.joinpath("cfg.json").exists():
            raise FileNotFoundError("No config file found; try specifying a layer.")
        return Sae.load_from_disk(repo_path, device=device, decoder=decoder)

    @staticmethod
    def load_from_disk(path: Union[Path, str], device: Union[str, torch.device] = "cpu", *,
                       decoder: bool = True) -> "Sae":
        from safetensors.torch import load_model

        path = Path(path)
        with open(path / "cfg.json", "r") as f:
            cfg_dict = json.load(f)
        d_in = cfg_dict.pop("d_in")
        cfg = SaeConfig.from_dict(cfg_dict)
        sae = Sae(d_in, cfg, device=device, decoder=decoder)
        load_model(model=sae, filename=str(path / "sae.safetensors"), device=str(device),
                   strict=decoder)
        sae.invalidate_prepared()
        return sae

    def save_to_disk(self, path: Union[Path, str]):
        from safetensors.torch import save_model

        path = Path(path)
        path.mkdir(parents=True, exist_ok=True)
        save_model(self, str(path / "sae.safetensors"))
        with open(path / "cfg.json", "w") as f:
            json.dump({**self.cfg.to_dict(), "d_in": self.d_in}, f)

    @property
    def device(self):
        return self.encoder.weight.device

    @property
    def dtype(self):
        return self.encoder.weight.dtype

    # ---- hot path -----------------------------------------------------------------------------------
    def pre_acts(self, x: Tensor) -> Tensor:
        """relu((x - b_dec) W_enc^T + b_enc) as a dense [..., N] f32 tensor (sae.py:172-177).
        Kept for callers that edit or reduce the dense latents (steering.py:111-114,
        tools/probe_activations.py:116); the caching / encode paths use the fused op instead."""
        return ops.pre_acts(x, self.encoder.weight, self.encoder.bias, self.b_dec)

    def select_topk(self, latents: Tensor) -> EncoderOutput:
        """Top-k latents (sae.py:179-181).  Order is canonical (value desc, index asc), a valid
        instance of the reference's `sorted=False`."""
        return EncoderOutput(*ops.topk(latents, self.cfg.k))

    def invalidate_prepared(self) -> None:
        """Drop the cached coarse-pass operands (bf16 / int8 copies of encoder.weight).  They are
        rebuilt when `encoder.weight`'s autograd version changes; an edit through `.data`, `copy_`
        into `.data` or a raw-pointer kernel does NOT bump the version -- call this after one.  (Stale
        operands cannot corrupt results silently in general: every re-scored pair checks the error
        model and sends the token to the exact path -- but that is the slow path.)"""
        self._prepared = None
        self._prepared_key = None
        if self.encoder.weight.is_cuda:
            ops.invalidate_train_operands(self.encoder.weight)     # the training loop's per-parameter buffer as well
            ops.invalidate_certified(self.encoder.weight)          # ... and the certified pass's two-plane operands

    def _prepared_weights(self) -> Optional[Tensor]:
        w = self.encoder.weight
        # (+ whether the fp8 pass is in force: its operands replace the int8 ones in the buffer; + the dither mode: the
        # operands of a large batch's subtractive dither are rounded against the seed of the PREPARE, so a buffer prepared
        # with the dither off cannot serve a dithering encode except through the exact path -- include/msae.h, `dither`)
        key = (w.data_ptr(), w._version, tuple(w.shape), w.device,
               ops.coarse_in_force() == "fp8", ops.dither_in_force())
        if self._prepared is None or self._prepared_key != key:
            self._prepared = ops.prepare_encoder(w)
            self._prepared_key = key
        return self._prepared

    def encode(self, x: Tensor, *, set_feature: int = -1, set_value: float = 0.0,
               zero_feature: int = -1, return_status: bool = False, resolve: bool = True,
               differentiable: Optional[bool] = None, exact: bool = False, certified: bool = False):
        """Fused encode + TopK (sae.py:183-185).  `set_feature/set_value` and `zero_feature` apply
        the steering / attribution hooks' edits of the dense latents (steering.py:113-114,
        patching/utils.py:43-48) inside the kernel, before TopK.

        The kernel verifies every token and recomputes the ones it cannot verify (degenerate
        activations: all-zero rows, fewer than k positive latents, tokens the error model of the
        candidate pass does not describe ...) exactly inside the call, on the device; nothing is read
        back, so the method is stream-ordered like every other op.  `status` (return_status=True):
        0 verified, 1 recomputed exactly.  `resolve` is accepted for compatibility and ignored.
        Guarantees (include/msae.h): the default int8 pass rounds its operands stochastically with per-call seeds, so a
        member of the true top-k is missed with probability <= k exp(-z^2 / 2) (7e-10 at z = 7, k = 32) for EVERY input;
        `certified=True` runs the two-plane pass with a deterministic error band (no probability left, ~3x the time on
        large batches); `exact=True` computes EVERY token by the exact path (msae_options::exact).  Both are inference switches.

        Autograd follows the reference (sae.py:183-185 builds a graph whenever autograd would): the call is one differentiable
        node (sparse backward through the selected latents) when gradients are enabled and `x` requires grad (the attribution
        hooks: the LLM's hidden states do), or the module is in TRAINING mode and a parameter requires grad (a loaded module is,
        as in the reference, until `.eval()` / `.requires_grad_(False)`), or `differentiable=True` is passed.  The detached fast
        path is what runs under `torch.no_grad()` (the cache and steering hooks) and in `eval()` mode."""
        if differentiable is None:
            params = self.encoder.weight.requires_grad or self.encoder.bias.requires_grad or self.b_dec.requires_grad
            differentiable = x.requires_grad or (self.training and params and not (exact or certified or return_status))
        want_grad = torch.is_grad_enabled() and differentiable
        if want_grad and (exact or certified):
            raise RuntimeError("Sae.encode(exact=True / certified=True) is an inference switch: differentiate pre_acts -> "
                               "select_topk (the exact path with autograd) instead")
        if want_grad and return_status:
            raise RuntimeError("Sae.encode(return_status=True) returns non-differentiable outputs: call it under "
                               "torch.no_grad(), or without return_status where gradients must flow")
        if want_grad:
            # autograd must flow (attribution patching: patching/utils.py:33-58, attribution.py:165):
            # same kernel, as one autograd node with the sparse backward
            (acts, idx), = ops.sparse_encode(x, self.encoder.weight, self.encoder.bias, self.b_dec, self.cfg.k,
                                             prepared=self._prepared_weights(), set_feature=set_feature,
                                             set_value=set_value, zero_feature=zero_feature)
            return EncoderOutput(acts, idx)
        with torch.no_grad():      # (a custom op without an autograd formula would hang a raising node on the outputs)
            acts, idx, status = ops.encode_topk(x, self.encoder.weight, self.encoder.bias, self.b_dec,
                                                self._prepared_weights(), self.cfg.k, set_feature,
                                                float(set_value), zero_feature, exact=exact, certified=certified)
        out = EncoderOutput(acts, idx)
        return (out, status) if return_status else out

    def decode(self, top_acts: Tensor, top_indices: Tensor) -> Tensor:
        assert self.W_dec is not None, "Decoder weight was not initialized."
        # the module-level seam, as the reference (sae.py:190): `decoder_impl(top_indices, top_acts, W_dec.mT) + b_dec`.  The
        # default implementation is called with the bias inside the kernel (same bits: the kernel adds b_dec behind the
        # chain; one pass over the [A, d] output instead of two); a rebound decoder_impl gets the reference's call.
        impl = _seam.decoder_impl
        if impl is _seam.hip_decode:
            return ops.decode(top_indices, top_acts.to(self.dtype), self.W_dec, self.b_dec)
        return impl(top_indices, top_acts.to(self.dtype), self.W_dec.mT) + self.b_dec

    def forward(self, x: Tensor, dead_mask: Union[Tensor, None] = None) -> ForwardOutput:
        """Training forward (sae.py:193-247): reconstruction, FVU, AuxK and Multi-TopK terms, fully
        differentiable.  The encoder is one autograd node with a sparse backward (ops._SparseEncode):
        gradients reach encoder.weight / encoder.bias / b_dec / x only through the selected latents,
        exactly as in the reference's dense graph, without its second [T,N]x[T,d] GEMM."""
        k_aux, scale = 0, 0.0
        if dead_mask is not None and (num_dead := int(dead_mask.sum())) > 0:
            k_aux = x.shape[-1] // 2                      # heuristic from the paper (sae.py:209)
            scale = min(num_dead / k_aux, 1.0)
            k_aux = min(k_aux, num_dead)
        k_multi = 4 * self.cfg.k if self.cfg.multi_topk else 0
        sel = ops.sparse_encode(x, self.encoder.weight, self.encoder.bias, self.b_dec, self.cfg.k,
                                dead_mask, k_aux, k_multi)
        top_acts, top_indices = sel[0]
        sae_out = self.decode(top_acts, top_indices)
        # the loss terms in as few T x d sweeps as autograd allows (each elementwise torch op on [T, d] is a 45-us kernel at
        # C2; the reference's expression is ~10 of them forward + backward): squared error and its gradient through _SqErr,
        # the total variance as one column-variance kernel.  An input that itself requires grad keeps the plain expressions.
        lean = x.is_cuda and x.dim() == 2 and not x.requires_grad and x.dtype == torch.float32
        if lean:
            sq_err, e = _SqErr.apply(sae_out, x)
            total_variance = _total_variance(x)
        else:
            e = sae_out - x
            sq_err = e.pow(2).sum()
            total_variance = (x - x.mean(0)).pow(2).sum()

        if k_aux > 0:
            auxk_acts, auxk_indices = sel[1]
            e_hat = self.decode(auxk_acts, auxk_indices)
            # (e is the residual the dead latents should explain: a constant target, as in the reference's graph it carries
            # gradient to the main path too -- keep that: use the differentiable residual when it matters)
            e_t = e if not lean else (sae_out - x)
            auxk_loss = scale * (e_hat - e_t).pow(2).sum() / total_variance
        else:
            auxk_loss = sae_out.new_tensor(0.0)

        fvu = sq_err / total_variance
        if k_multi > 0:
            top_acts, top_indices = sel[-1]
            sae_out = self.decode(top_acts, top_indices)
            multi_topk_fvu = (_SqErr.apply(sae_out, x)[0] if lean else (sae_out - x).pow(2).sum()) / total_variance
        else:
            multi_topk_fvu = sae_out.new_tensor(0.0)
        return ForwardOutput(sae_out, top_acts, top_indices, fvu, auxk_loss, multi_topk_fvu)

    # ---- decoder-norm utilities (sae.py:249-271) ------------------------------------------------------
    @torch.no_grad()
    def set_decoder_norm_to_unit_norm(self):
        assert self.W_dec is not None, "Decoder weight was not initialized."
        eps = torch.finfo(self.W_dec.dtype).eps
        W = self.W_dec.data
        if W.is_cuda and W.dtype == torch.float32 and W.is_contiguous():
            ops.unit_norm_rows_(W, eps)       # one read + one write of the 2 GiB matrix
        else:                                 # module still on the host (construction, checkpoint tools)
            W /= torch.norm(W, dim=1, keepdim=True) + eps

    @torch.no_grad()
    def remove_gradient_parallel_to_decoder_directions(self):
        assert self.W_dec is not None, "Decoder weight was not initialized."
        assert self.W_dec.grad is not None
        along = (self.W_dec.grad * self.W_dec.data).sum(dim=1, keepdim=True)
        self.W_dec.grad -= along * self.W_dec.data

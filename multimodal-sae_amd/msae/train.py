"""One SAE optimisation step on streamed activations (BASELINE configs[3]).

Mirrors the inner loop of the reference trainer (train/sae/sae/trainer.py:347-414) for one
hookpoint: renormalise the decoder, forward (FVU + AuxK + Multi-TopK, `Sae.forward`), backward through
the HIP kernels (sparse encoder backward, decoder gather/scatter backward), data-parallel gradient
averaging (what DDP does, trainer.py:338-345), `clip_grad_norm_(1.0)`, removal of the decoder-parallel
gradient component, Adam, dead-latent bookkeeping (`did_fire` MAX-reduced, trainer.py:387-388,404-408).
Clip, parallel-component removal and Adam run as ONE pass per parameter (csrc/train.hip: ~38 GB of HBM
traffic per step at C2 instead of the ~67 GB of the separate torch ops); the global gradient norm stays
on the device, so the step never synchronises with the host.
The LLM forward that produces `hiddens`, dataset plumbing, wandb and checkpointing are the reference
trainer's outer loop and out of scope.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor

from . import ops
from .sae import Sae


class SaeTrainStep:
    def __init__(self, sae: Sae, lr: Optional[float] = None, auxk_alpha: float = 0.0,
                 dead_feature_threshold: int = 10_000_000, group=None):
        self.sae, self.auxk_alpha, self.group = sae, auxk_alpha, group
        self.dead_feature_threshold = dead_feature_threshold
        if lr is None:  # trainer.py:131: 2e-4 scaled by 1/sqrt(N / 2^14)
            lr = 2e-4 / (sae.num_latents / (2 ** 14)) ** 0.5
        self.lr, self.betas, self.eps, self.max_grad_norm = lr, (0.9, 0.999), 1e-8, 1.0
        self.params = [p for p in sae.parameters()]
        assert all(p.dtype == torch.float32 for p in self.params), "the SAE trains in fp32 (trainer.py:190)"
        self.exp_avg = [torch.zeros_like(p) for p in self.params]       # torch.optim.Adam state
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.t = 0
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=sae.device)
        self.num_tokens_since_fired = torch.zeros(sae.num_latents, dtype=torch.long, device=sae.device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def state_dict(self) -> dict:
        return {"step": self.t, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "num_tokens_since_fired": self.num_tokens_since_fired}

    def load_state_dict(self, sd: dict) -> None:
        self.t = int(sd["step"])
        for dst, src in zip(self.exp_avg + self.exp_avg_sq, list(sd["exp_avg"]) + list(sd["exp_avg_sq"])):
            dst.copy_(src)
        self.num_tokens_since_fired.copy_(sd["num_tokens_since_fired"])

    def _all_reduce(self, t: Tensor, op=None):
        if self.world > 1:
            dist.all_reduce(t, op=op or dist.ReduceOp.SUM, group=self.group)
        return t

    def step(self, hiddens: Tensor) -> dict:
        sae = self.sae
        if sae.cfg.normalize_decoder:
            sae.set_decoder_norm_to_unit_norm()
        dead_mask = (self.num_tokens_since_fired > self.dead_feature_threshold) if self.auxk_alpha > 0 else None
        out = sae(hiddens, dead_mask)
        loss = out.fvu + self.auxk_alpha * out.auxk_loss + out.multi_topk_fvu / 8
        loss.backward()
        did_fire = torch.zeros(sae.num_latents, dtype=torch.bool, device=sae.device)
        did_fire[out.latent_indices.flatten()] = True
        if self.world > 1:
            fired = did_fire.to(torch.int32)
            self._all_reduce(fired, dist.ReduceOp.MAX)
            did_fire = fired.bool()
            for p in sae.parameters():          # DDP semantics: gradients averaged over the ranks
                if p.grad is not None:
                    self._all_reduce(p.grad).div_(self.world)
        # clip_grad_norm_(1.0) -> remove_gradient_parallel_to_decoder_directions -> Adam, fused
        self._sumsq.zero_()
        for p in self.params:
            if p.grad is not None:
                ops.grad_sumsq_(self._sumsq, p.grad)
        self.t += 1
        for p, m, v in zip(self.params, self.exp_avg, self.exp_avg_sq):
            if p.grad is None:
                continue
            ops.adam_rows_(p.data, p.grad, m, v, self.t, self.lr, total_sumsq=self._sumsq,
                           max_norm=self.max_grad_norm, betas=self.betas, eps=self.eps,
                           project=sae.cfg.normalize_decoder and p is sae.W_dec)
            p.grad = None
            torch.autograd.graph.increment_version(p)    # p changed behind autograd's back: caches keyed on it go stale
        n_tok = torch.tensor(hiddens.shape[0], device=sae.device)
        self._all_reduce(n_tok)
        self.num_tokens_since_fired += n_tok
        self.num_tokens_since_fired[did_fire] = 0
        stats = torch.stack([out.fvu.detach(), out.auxk_loss.detach(), out.multi_topk_fvu.detach()])
        if self.world > 1:
            self._all_reduce(stats).div_(self.world)
        return {"fvu": stats[0].item(), "auxk_loss": stats[1].item(), "multi_topk_fvu": stats[2].item()}

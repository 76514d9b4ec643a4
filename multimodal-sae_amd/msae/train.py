"""One SAE optimisation step on streamed activations (BASELINE configs[3]).

Mirrors the inner loop of the reference trainer (train/sae/sae/trainer.py:347-414) for one
hookpoint: renormalise the decoder, forward (FVU + AuxK + Multi-TopK, `Sae.forward`), backward through
the HIP kernels (sparse encoder backward, decoder gather/scatter backward), data-parallel gradient
averaging (what DDP does, trainer.py:338-345), `clip_grad_norm_(1.0)`, removal of the decoder-parallel
gradient component, Adam, dead-latent bookkeeping (`did_fire` MAX-reduced, trainer.py:387-388,404-408).
The LLM forward that produces `hiddens`, dataset plumbing, wandb and checkpointing are the reference
trainer's outer loop and out of scope.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor

from .sae import Sae


class SaeTrainStep:
    def __init__(self, sae: Sae, lr: Optional[float] = None, auxk_alpha: float = 0.0,
                 dead_feature_threshold: int = 10_000_000, group=None):
        self.sae, self.auxk_alpha, self.group = sae, auxk_alpha, group
        self.dead_feature_threshold = dead_feature_threshold
        if lr is None:  # trainer.py:131: 2e-4 scaled by 1/sqrt(N / 2^14)
            lr = 2e-4 / (sae.num_latents / (2 ** 14)) ** 0.5
        # fused=True: one kernel over all parameters instead of ~9 foreach passes (23 -> ~5 ms at C2)
        self.optimizer = torch.optim.Adam(sae.parameters(), lr=lr, fused=sae.device.type == "cuda")
        self.num_tokens_since_fired = torch.zeros(sae.num_latents, dtype=torch.long, device=sae.device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

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
        torch.nn.utils.clip_grad_norm_(sae.parameters(), 1.0)
        if sae.cfg.normalize_decoder:
            sae.remove_gradient_parallel_to_decoder_directions()
        self.optimizer.step()
        self.optimizer.zero_grad()
        n_tok = torch.tensor(hiddens.shape[0], device=sae.device)
        self._all_reduce(n_tok)
        self.num_tokens_since_fired += n_tok
        self.num_tokens_since_fired[did_fire] = 0
        stats = torch.stack([out.fvu.detach(), out.auxk_loss.detach(), out.multi_topk_fvu.detach()])
        if self.world > 1:
            self._all_reduce(stats).div_(self.world)
        return {"fvu": stats[0].item(), "auxk_loss": stats[1].item(), "multi_topk_fvu": stats[2].item()}

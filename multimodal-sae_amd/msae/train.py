"""SAE optimisation steps on streamed activations (BASELINE configs[3]).

Mirrors the inner loop of the reference trainer (train/sae/sae/trainer.py:316-414) for one hookpoint:
first-step `b_dec` = geometric median of the batch (`:325-332`, sae/utils.py:37-62), decoder
renormalisation, forward in `micro_acc_steps` chunks (FVU + AuxK + Multi-TopK, `Sae.forward`), backward
through the HIP kernels (sparse encoder backward, decoder gather/scatter backward), data-parallel
gradient averaging (what DDP does, `:338-345`), `clip_grad_norm_(1.0)` every step, and every
`grad_acc_steps` steps: removal of the decoder-parallel gradient component, Adam with the linear
warm-up / linear decay schedule (`get_linear_schedule_with_warmup`, `:155-157`), dead-latent
bookkeeping (`did_fire` MAX-reduced, `:387-388,404-408`).

MI355X-side design:
  * clip, parallel-component removal and Adam run as ONE pass per parameter (csrc/train.hip: ~38 GB of HBM
    traffic per step at C2 instead of ~67 GB of separate torch ops); the global gradient norm stays on the
    device, so a step never synchronises with the host (statistics are returned as device tensors);
  * passes that would re-read a matrix another kernel has just produced are folded into the producer
    (`fuse_next_step`, round 4): the squared gradient norm of the two weight gradients is accumulated per row by
    the weight-gradient kernel itself (row in registers; a fixed-order 512-KB sum replaces two 2-GiB reads), the
    NEXT step's decoder renormalisation and the next encode's int8 / bf16 operands of the encoder are produced
    by the Adam pass from the row it has just updated (same bits as the separate passes; ~11 GB less traffic per
    step).  Between steps `W_dec` therefore holds unit-norm rows (the reference renormalises at the top of the
    next step, trainer.py:352 -- the same values one kernel later; a checkpoint written between steps differs from
    the reference's by that normalisation, ~1e-5 relative);
  * data parallel: each parameter's gradient all-reduce (RCCL, ReduceOp.AVG -- no division pass) is
    launched ASYNCHRONOUSLY from a post-accumulate-grad hook the moment autograd has finished that
    parameter, so the 2 GiB W_dec exchange over xGMI overlaps the encoder backward, and the W_enc
    exchange overlaps the gradient-norm pass of W_dec; the four parameters are their own buckets (two
    of 2 GiB, two tiny).  A ring all-reduce of 2 x 2 GiB moves ~7 GiB per GPU per step over the 7 xGMI
    links -- of the order of the 18 ms compute of a T = 8192 step, so `grad_acc_steps` / larger per-GPU
    batches are what make the 8-GPU step compute-bound, exactly as in the reference.
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


@torch.no_grad()
def geometric_median(points: Tensor, max_iter: int = 100, tol: float = 1e-5) -> Tensor:
    """Weiszfeld iterations from the mean (sae/utils.py:37-62); initialises the decoder bias."""
    guess = points.mean(dim=0)
    for _ in range(max_iter):
        prev = guess
        weights = 1 / torch.norm(points - guess, dim=1)
        weights /= weights.sum()
        guess = (weights.unsqueeze(1) * points).sum(dim=0)
        if torch.norm(guess - prev) < tol:
            break
    return guess


def linear_schedule_with_warmup(step: int, warmup_steps: int, total_steps: Optional[int]) -> float:
    """LR multiplier of transformers.get_linear_schedule_with_warmup at scheduler step `step`."""
    if step < warmup_steps:
        return step / max(1, warmup_steps)
    if total_steps is None:
        return 1.0
    return max(0.0, (total_steps - step) / max(1, total_steps - warmup_steps))


class SaeTrainStep:
    def __init__(self, sae: Sae, lr: Optional[float] = None, auxk_alpha: float = 0.0,
                 dead_feature_threshold: int = 10_000_000, group=None, grad_acc_steps: int = 1,
                 micro_acc_steps: int = 1, lr_warmup_steps: int = 0, total_steps: Optional[int] = None,
                 init_b_dec: bool = False, fuse_next_step: bool = True):
        from .parallel import _unpinned

        group = _unpinned(group)          # (the default group is addressed as None: nothing here keeps it alive past its destroy)
        self.sae, self.auxk_alpha, self.group = sae, auxk_alpha, group
        self.dead_feature_threshold = dead_feature_threshold
        if lr is None:  # trainer.py:131: 2e-4 scaled by 1/sqrt(N / 2^14)
            lr = 2e-4 / (sae.num_latents / (2 ** 14)) ** 0.5
        self.lr, self.betas, self.eps, self.max_grad_norm = lr, (0.9, 0.999), 1e-8, 1.0
        self.grad_acc_steps, self.micro_acc_steps = grad_acc_steps, micro_acc_steps
        self.lr_warmup_steps, self.total_steps, self.init_b_dec = lr_warmup_steps, total_steps, init_b_dec
        self.params = [p for p in sae.parameters()]
        assert all(p.dtype == torch.float32 for p in self.params), "the SAE trains in fp32 (trainer.py:190)"
        self.exp_avg = [torch.zeros_like(p) for p in self.params]       # torch.optim.Adam state
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.t = 0                     # optimizer steps taken (= lr scheduler steps)
        self.global_step = 0           # batches seen (trainer.py:394)
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=sae.device)
        self.num_tokens_since_fired = torch.zeros(sae.num_latents, dtype=torch.long, device=sae.device)
        self._did_fire = torch.zeros(sae.num_latents, dtype=torch.bool, device=sae.device)
        self._tokens_in_step = torch.zeros((), dtype=torch.long, device=sae.device)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._pending, self._reduce_now = [], False
        # round 4: gradient norm from the weight-gradient kernels, next step's renorm + encoder operands from the Adam pass
        self.fuse_next_step = fuse_next_step
        self._normed_version = None        # W_dec._version whose rows the Adam pass left unit-norm
        self._wgrad_sumsq: dict = {}
        self._chunk_tokens = 0
        self._avg_op = None
        if self.world > 1:
            # AVG exists on nccl (RCCL) only; gloo (CPU tests) sums and divides
            self._avg_op = dist.ReduceOp.AVG if dist.get_backend(group) == "nccl" else None
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._grad_ready)

    @property
    def current_lr(self) -> float:
        return self.lr * linear_schedule_with_warmup(self.t, self.lr_warmup_steps, self.total_steps)

    def state_dict(self) -> dict:
        return {"step": self.t, "global_step": self.global_step, "exp_avg": self.exp_avg,
                "exp_avg_sq": self.exp_avg_sq, "num_tokens_since_fired": self.num_tokens_since_fired}

    def load_state_dict(self, sd: dict) -> None:
        self.t = int(sd["step"])
        self.global_step = int(sd.get("global_step", self.t * self.grad_acc_steps))
        for dst, src in zip(self.exp_avg + self.exp_avg_sq, list(sd["exp_avg"]) + list(sd["exp_avg_sq"])):
            dst.copy_(src)
        self.num_tokens_since_fired.copy_(sd["num_tokens_since_fired"])

    # ---- data-parallel plumbing -------------------------------------------------------------------------
    def _grad_ready(self, p: Tensor) -> None:
        """post-accumulate-grad hook: this parameter's gradient is final for this backward -> start its
        all-reduce now (asynchronously) if this is the step's last micro-batch."""
        if self._reduce_now:
            op = self._avg_op if self._avg_op is not None else dist.ReduceOp.SUM
            self._pending.append((dist.all_reduce(p.grad, op=op, group=self.group, async_op=True), p))

    def _join_reductions(self) -> None:
        for work, p in self._pending:
            work.wait()
            if self._avg_op is None:
                p.grad.div_(self.world)
        self._pending = []

    def _all_reduce(self, t: Tensor, op=None) -> Tensor:
        if self.world > 1:
            dist.all_reduce(t, op=op or dist.ReduceOp.SUM, group=self.group)
        return t

    # ---- compute (overridden with torch-CPU restatements by the gloo tests) ----------------------------------
    def _forward(self, hiddens: Tensor, dead_mask: Optional[Tensor]):
        return self.sae(hiddens, dead_mask)

    def _renorm_decoder(self) -> None:
        W = self.sae.W_dec
        if self._normed_version is not None and W._version == self._normed_version:
            return                          # the previous step's Adam pass has already normalised these rows
        self.sae.set_decoder_norm_to_unit_norm()

    def _clip_in_place(self) -> None:
        """clip_grad_norm_(1.0) on accumulated gradients that are NOT consumed by an optimizer step now."""
        torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)

    def _update(self, lr: float) -> None:
        """clip_grad_norm_(1.0) -> remove_gradient_parallel_to_decoder_directions -> Adam, fused."""
        sae = self.sae
        self._sumsq.zero_()
        # the squared norm of a weight gradient that came out of exactly ONE weight-gradient kernel call and was not
        # touched since (no accumulation, no all-reduce) is what that kernel recorded per row
        single = self.world == 1 and self.grad_acc_steps * self.micro_acc_steps == 1
        for p in self.params:
            if p.grad is None:
                continue
            ent = self._wgrad_sumsq.get(p.data_ptr()) if single else None
            if ent is not None and ent[0] == 1 and ent[1] == p.grad.data_ptr():
                ops.sum_into_(self._sumsq, ent[2])
            else:
                ops.grad_sumsq_(self._sumsq, p.grad)
        self._wgrad_sumsq = {}
        fuse = self.fuse_next_step and sae.W_dec.is_cuda
        for p, m, v in zip(self.params, self.exp_avg, self.exp_avg_sq):
            if p.grad is None:
                continue
            kw = {}
            is_dec = p is sae.W_dec
            is_enc = p is sae.encoder.weight
            if fuse and is_dec and sae.cfg.normalize_decoder:
                kw["renorm_eps"] = torch.finfo(p.dtype).eps          # sae.py:252
            if fuse and is_enc and self._chunk_tokens > 0 and ops.coarse_in_force() != "fp8":   # (the fused pass builds int8 / bf16 operands; MSAE_COARSE=fp8 in the environment counts: ADVICE r5)
                kw["refresh"], kw["tokens_next"] = ops.train_operand_buffer(p), self._chunk_tokens
            ops.adam_rows_(p.data, p.grad, m, v, self.t, lr, total_sumsq=self._sumsq,
                           max_norm=self.max_grad_norm, betas=self.betas, eps=self.eps,
                           project=sae.cfg.normalize_decoder and is_dec, **kw)
            p.grad = None
            torch.autograd.graph.increment_version(p)    # p changed behind autograd's back: caches keyed on it go stale
            if "renorm_eps" in kw:
                self._normed_version = p._version
            if "refresh" in kw:
                ops.mark_train_operands_fresh(p, self._chunk_tokens)

    # ---- one batch ---------------------------------------------------------------------------------------------
    def step(self, hiddens: Tensor) -> dict:
        """One global step (one batch of activations of this rank).  Returns device tensors (no host
        synchronisation): {"fvu", "auxk_loss", "multi_topk_fvu"} averaged over micro-batches and ranks,
        "stepped": whether the optimizer moved (every `grad_acc_steps`-th call)."""
        sae = self.sae
        if self.global_step == 0 and self.init_b_dec:            # trainer.py:325-332
            pts = hiddens.float()
            if self.world > 1:
                parts = [torch.empty_like(pts) for _ in range(self.world)]
                dist.all_gather(parts, pts, group=self.group)
                pts = torch.cat(parts)
            sae.b_dec.data.copy_(geometric_median(pts).to(sae.b_dec.dtype))
        if sae.cfg.normalize_decoder:
            self._renorm_decoder()
        dead_mask = (self.num_tokens_since_fired > self.dead_feature_threshold) if self.auxk_alpha > 0 else None
        acc_steps = self.grad_acc_steps * self.micro_acc_steps
        stats = torch.zeros(3, dtype=torch.float32, device=hiddens.device)
        chunks = hiddens.chunk(self.micro_acc_steps)
        for ci, chunk in enumerate(chunks):
            self._reduce_now = self.world > 1 and ci == len(chunks) - 1
            self._chunk_tokens = chunk.shape[0]
            out = self._forward(chunk, dead_mask)
            loss = out.fvu + self.auxk_alpha * out.auxk_loss + out.multi_topk_fvu / 8
            if hiddens.is_cuda:
                with ops.collect_wgrad_sumsq() as self._wgrad_sumsq:
                    loss.div(acc_steps).backward()
            else:
                loss.div(acc_steps).backward()
            stats += torch.stack([out.fvu.detach(), out.auxk_loss.detach(), out.multi_topk_fvu.detach()])
            self._did_fire[out.latent_indices.flatten()] = True
        self._reduce_now = False
        if self.world > 1:
            fired = self._did_fire.to(torch.int32)
            self._all_reduce(fired, dist.ReduceOp.MAX)             # max is boolean "any"
            self._did_fire = fired.bool()
            self._join_reductions()                               # DDP semantics: gradients averaged over the ranks
        n_tok = torch.tensor(hiddens.shape[0], device=hiddens.device)
        self._tokens_in_step += self._all_reduce(n_tok)
        self.global_step += 1
        stepped = self.global_step % self.grad_acc_steps == 0
        if stepped:
            lr = self.current_lr
            self.t += 1
            self._update(lr)
            self.num_tokens_since_fired += self._tokens_in_step
            self.num_tokens_since_fired[self._did_fire] = 0
            self._tokens_in_step.zero_()
            self._did_fire.zero_()
        else:
            self._clip_in_place()                                  # trainer.py:391: clipped every batch
        stats /= len(chunks)
        if self.world > 1:
            self._all_reduce(stats).div_(self.world)
        return {"fvu": stats[0], "auxk_loss": stats[1], "multi_topk_fvu": stats[2], "stepped": stepped}

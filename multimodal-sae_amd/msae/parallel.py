"""Feature-axis sharding of the SAE over the GPUs of one node (BASELINE configs[2]).

No reference counterpart: the reference only data-parallelises by sharding the dataset
(launch/cache/cache.py:66); SURVEY.md section 8(e) specifies this path.

Rank g of G owns rows [g*N/G, (g+1)*N/G) of W_enc / b_enc.  Every rank sees the same tokens x
(each rank runs the same LLM forward, or x is broadcast).  Per call:

  1. local fused encode + TopK over the shard           -> (vals f32, idx i32 + g*N/G)  [T, k]
  2. ONE all-gather of the packed pairs (8 B * k per token per rank = 256 B at k = 32) over
     RCCL/xGMI -- latency-bound, the only exchange step of the path
  3. every rank merges the G*k candidates per token with the canonical key (value desc, global
     index asc); global top-k is a subset of the union of per-shard top-k, so the result is
     bit-identical to the single-GPU result
  4. decode is token-sharded against a replicated W_dec (2 GiB of 288 GB): rank g reconstructs
     tokens [g*T/G, (g+1)*T/G) and an all-gather returns the full [T, d] to every rank (the hook
     that replaces the layer output needs it everywhere; the caching path skips it).

One process per GPU (`torch.distributed`, backend "nccl" == RCCL on ROCm).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist
from torch import Tensor


def canonical_key(vals: Tensor, idx: Tensor) -> Tensor:
    """int64 key whose descending order is (value desc, index asc) -- same order as the kernels'
    rank_key (csrc/common.h)."""
    b = vals.contiguous().view(torch.int32).to(torch.int64)
    b = torch.where(b == -(1 << 31), torch.zeros_like(b), b)  # -0.0 ties with +0.0
    u = torch.where(b < 0, -b - 1 - (1 << 31), b)             # order-preserving, signed 32-bit range
    return (u << 32) | (0x7FFFFFFF - idx.to(torch.int64))


def merge_topk(vals: Tensor, idx: Tensor, k: int):
    """Canonical top-k of candidate pairs along the last dim.  vals/idx: [..., M] -> [..., k]."""
    key = canonical_key(vals, idx)
    top = torch.topk(key, k, dim=-1, largest=True, sorted=True)
    return torch.gather(vals, -1, top.indices), torch.gather(idx, -1, top.indices)


def token_slice(T: int, rank: int, world: int):
    per = (T + world - 1) // world
    return min(rank * per, T), min((rank + 1) * per, T), per


class ShardedSae:
    """Inference engine over a feature-sharded encoder.  With world == 1 it is the plain fused
    encode -> decode pipeline (what bench.py times on one GPU)."""

    def __init__(self, W_enc_shard: Tensor, b_enc_shard: Tensor, W_dec: Tensor, b_dec: Tensor, k: int,
                 rank: int = 0, world: int = 1, group=None,
                 encode_fn: Optional[Callable] = None, decode_fn: Optional[Callable] = None,
                 force_collectives: bool = False):
        self.W_enc, self.b_enc, self.W_dec, self.b_dec = W_enc_shard, b_enc_shard, W_dec, b_dec
        self.k, self.rank, self.world, self.group = k, rank, world, group
        self.n_loc = W_enc_shard.shape[0]
        # collectives run whenever there is more than one rank; `force_collectives` also runs them on
        # a 1-rank group so the RCCL code path can be exercised on a single-GPU box
        self.collective = world > 1 or (force_collectives and dist.is_initialized())
        self.decode_events = None
        self.decode_event_i = 0
        if encode_fn is None:
            from . import ops

            prepared = ops.prepare_encoder(W_enc_shard)
            encode_fn = lambda x: ops.encode_topk(x, self.W_enc, self.b_enc, self.b_dec, prepared, k)
            decode_fn = lambda idx, vals: ops.decode(idx, vals, self.W_dec, self.b_dec)
        self._encode, self._decode = encode_fn, decode_fn

    def encode(self, x: Tensor):
        """-> (top_acts [T,k] f32, top_indices [T,k] int64 GLOBAL feature ids, status [T])."""
        vals, idx, status = self._encode(x)
        if not self.collective:
            return vals, idx, status
        T = vals.shape[0]
        packed = torch.stack((vals.view(torch.int32), (idx + self.rank * self.n_loc).to(torch.int32)), 0)
        flat = torch.empty((self.world * 2, T, self.k), dtype=torch.int32, device=packed.device)
        dist.all_gather_into_tensor(flat, packed.contiguous(), group=self.group)  # concat along dim 0
        gathered = flat.view(self.world, 2, T, self.k)
        all_vals = gathered[:, 0].view(torch.float32).permute(1, 0, 2).reshape(T, self.world * self.k)
        all_idx = gathered[:, 1].permute(1, 0, 2).reshape(T, self.world * self.k).to(torch.int64)
        vals, idx = merge_topk(all_vals, all_idx, self.k)
        return vals, idx, status

    def decode(self, vals: Tensor, idx: Tensor, gather: bool = True) -> Tensor:
        ev = None
        if self.decode_events is not None and self.decode_event_i < len(self.decode_events):
            ev = self.decode_events[self.decode_event_i]
            self.decode_event_i += 1
            ev[0].record()
        if not self.collective:
            out = self._decode(idx, vals)
        else:
            T = vals.shape[0]
            lo, hi, per = token_slice(T, self.rank, self.world)
            local = self._decode(idx[lo:hi].contiguous(), vals[lo:hi].contiguous())
            if gather:
                d = local.shape[-1]
                pad = torch.zeros(per, d, dtype=local.dtype, device=local.device)
                pad[: hi - lo] = local
                full = torch.empty(self.world * per, d, dtype=local.dtype, device=local.device)
                dist.all_gather_into_tensor(full, pad, group=self.group)
                out = full[:T]
            else:
                out = local
        if ev is not None:
            ev[1].record()
        return out

    def forward(self, x: Tensor) -> dict:
        vals, idx, status = self.encode(x)
        recon = self.decode(vals, idx)
        return {"sae_out": recon, "top_acts": vals, "top_indices": idx, "status": status}

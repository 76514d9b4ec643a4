"""Feature-activation caching on the fused HIP path.

Drop-in for the reference's `Cache`, `FeatureCache` and `FeatureImageCache`
(sae_auto_interp/features/cache.py:23-429): same constructor arguments, same `run / save /
save_splits / concate_safetensors` methods, and byte-compatible output files

    <save_dir>/<module>/Rank{r}_{start}_{end}.safetensors      per rank   (cache.py:282-309)
    <save_dir>/<module>/{start}_{end}.safetensors              after rank-0 concat (cache.py:249-280)

with keys `locations [nnz,3] int64 = (row, pos, feature)` and `activations [nnz] f32`, which is what
`features/loader.py:143-196` (FeatureDataset) reads.

What changed underneath: the reference computes dense `[B,S,N]` latents, `torch.topk`, a second
dense `zeros_like + scatter_`, and two threshold scans + `nonzero` (cache.py:209-217, 80-81) --
about 2.6 MB of HBM traffic per token at N = 131072.  Here `Sae.encode` (fused GEMM + TopK) yields
`[B,S,k]` pairs and `ops.sparsify` emits the COO records directly in the reference's order.

Reference quirk kept on purpose: `save_splits` masks
`start <= feature < end` with `end = boundary - 1`, so the last feature of every split is never
written (cache.py:243-247, 299).  `include_split_end=True` writes them.
"""
from __future__ import annotations

import os
import re
from collections import defaultdict
from typing import Dict, Optional, Sequence

import torch
import torch.distributed as dist
from safetensors.torch import load_file, save_file
from torch import Tensor

from .. import ops
from ..sae import Sae


def generate_split_indices(width: int, n_splits: int):
    """[(start, end)] with end = next boundary - 1 (cache.py:243-247; loader.py:143-162 names
    files `{start}_{end}`)."""
    b = torch.linspace(0, width, steps=n_splits + 1).long()
    return [(int(s), int(e) - 1) for s, e in zip(b[:-1], b[1:])]


class Cache:
    """Accumulates COO feature records per hooked module (cache.py:23-92)."""

    def __init__(self, shard_size: int, filters: Optional[Dict[str, Tensor]] = None,
                 batch_size: int = 64, spill_dir: Optional[str] = None, device_budget_bytes: int = 256 << 20):
        """`spill_dir`: stream every batch's records to disk instead of holding the whole run in host
        RAM (the reference keeps everything in Python lists until save_splits, cache.py:56-57);
        `save()` reads them back in batch order, so the final tensors are identical.
        `device_budget_bytes`: records of the fused path wait on the device (worst-case sized buffers, 7 MB
        per 8192-token batch at k = 32) and cross to the host in one transfer once this much is pending."""
        self.feature_locations = defaultdict(list)
        self.feature_activations = defaultdict(list)
        self.spill_dir = spill_dir
        self._spilled = defaultdict(list)
        self.filters = filters
        self.batch_size = batch_size
        self.shard_size = shard_size  # rows held by lower ranks (cache.py:39)
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self._bitmaps: Dict[str, Tensor] = {}
        self.device_budget_bytes = device_budget_bytes
        self._pending, self._pending_bytes = [], 0
        # groups of batches on their way to the host: (copy-done event, device buffers kept alive, pinned staging, layout)
        self._inflight = []
        self._copy_stream = None

    def _bitmap(self, module_path: str, num_latents: int, device) -> Optional[Tensor]:
        if self.filters is None:
            return None
        bm = self._bitmaps.get(module_path)
        if bm is None or bm.device != device or bm.numel() != num_latents:
            bm = torch.zeros(num_latents, dtype=torch.uint8, device=device)
            sel = self.filters[module_path].to(device=device, dtype=torch.int64)
            bm[sel] = 1
            self._bitmaps[module_path] = bm
        return bm

    def add_topk(self, top_acts: Tensor, top_indices: Tensor, num_latents: int, batch_number: int,
                 module_path: str):
        """Fused equivalent of scatter_ + Cache.add (cache.py:214-217, 42-57) for `[B,S,k]` pairs."""
        row_base = batch_number * self.batch_size + self.shard_size  # cache.py:55
        loc, act, nnz = ops.sparsify(top_acts, top_indices, num_latents, row_base=row_base, thresh=1e-5,
                                     filter_bitmap=self._bitmap(module_path, num_latents, top_acts.device), sync=False)
        # stays on the device, stream-ordered (the reference does a nonzero() + two .cpu() per batch inside the
        # hook loop): records go to the host in one transfer per `device_budget_bytes` of pending batches
        self._pending.append((module_path, loc, act, nnz))
        self._pending_bytes += loc.numel() * 8 + act.numel() * 4
        if self._pending_bytes >= self.device_budget_bytes:
            # this group starts its way to the host; the one before it (a whole budget's worth of batches ago) is finished
            self._start_transfer()
            self._drain(keep=1)

    # free pinned byte buffers, reused from group to group and from Cache to Cache (pinning a few hundred MB takes tens of
    # milliseconds: a pool per instance paid it again for every run)
    _staging: list = []

    def _pinned(self, nbytes: int) -> Tensor:
        pool = Cache._staging
        for i, b in enumerate(pool):
            if b.numel() >= nbytes:
                return pool.pop(i)
        return torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=True)

    def _start_transfer(self):
        """The pending batches' (worst-case sized) record buffers and counts go to PINNED host memory on a side stream, behind
        an event of the compute stream: the host does not wait and keeps enqueueing encodes (a `.cpu()` per batch is a pageable,
        blocking copy -- 470 MB per 64 batches of 8192 tokens stalled the loop for 9 % of its time)."""
        group, self._pending, self._pending_bytes = self._pending, [], 0
        dev = group[0][1].device
        if self._copy_stream is None or self._copy_stream.device != dev:
            self._copy_stream = torch.cuda.Stream(device=dev)
        layout, off = [], 0
        for (_, loc, act, _) in group:
            lb, ab = loc.numel() * 8, act.numel() * 4
            layout.append((off, lb, off + lb, ab))
            off += (lb + ab + 15) // 16 * 16
        counts_off = off
        host = self._pinned(off + 4 * len(group))
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ready)
            for (_, loc, act, nnz), (lo, lb, ao, ab) in zip(group, layout):
                host[lo:lo + lb].view(torch.int64).view(loc.shape).copy_(loc, non_blocking=True)
                host[ao:ao + ab].view(torch.float32).view(act.shape).copy_(act, non_blocking=True)
            host[counts_off:counts_off + 4 * len(group)].view(torch.int32).copy_(
                torch.stack([p[3] for p in group]).to(torch.int32), non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        self._inflight.append((done, group, host, layout, counts_off))

    def _drain(self, keep: int = 0):
        """Finish all but the `keep` most recent transfers: wait for the copy, slice the records out of the staging buffer
        (into ordinary host tensors, batch order) and give the staging buffer back."""
        while len(self._inflight) > keep:
            done, group, host, layout, counts_off = self._inflight.pop(0)
            done.synchronize()
            counts = host[counts_off:counts_off + 4 * len(group)].view(torch.int32).tolist()
            for (module_path, loc, act, _), (lo, lb, ao, ab), n in zip(group, layout, counts):
                loc_h = host[lo:lo + lb].view(torch.int64).view(loc.shape)[:n].clone()
                act_h = host[ao:ao + ab].view(torch.float32).view(act.shape)[:n].clone()
                self._append(module_path, loc_h, act_h)
            Cache._staging.append(host)

    def flush_pending(self):
        """Move every batch collected on the device to host memory (ONE host synchronisation)."""
        if self._pending:
            self._start_transfer()
        self._drain(0)

    def _append(self, module_path: str, loc: Tensor, act: Tensor):
        if self.spill_dir is None:
            self.feature_locations[module_path].append(loc)
            self.feature_activations[module_path].append(act)
            return
        d = os.path.join(self.spill_dir, module_path)
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f"batch_{len(self._spilled[module_path]):08d}.safetensors")
        save_file({"locations": loc.contiguous(), "activations": act.contiguous()}, path)
        self._spilled[module_path].append(path)

    def add(self, latents: Tensor, batch_number: int, module_path: str):
        """Legacy entry point taking DENSE `[B,S,N]` latents (cache.py:42-57)."""
        loc, act = self.get_nonzeros(latents, module_path)
        loc, act = loc.cpu(), act.cpu()
        loc[:, 0] += batch_number * self.batch_size + self.shard_size
        self._append(module_path, loc, act)

    def get_nonzeros(self, latents: Tensor, module_path: str):
        keep = latents.abs() > 1e-5  # cache.py:80-81
        loc, act = torch.nonzero(keep), latents[keep]
        if self.filters is None:
            return loc, act
        mask = torch.isin(loc[:, 2], self.filters[module_path].to(loc.device))
        return loc[mask], act[mask]

    def save(self):
        self.flush_pending()
        for module_path, paths in self._spilled.items():      # streamed batches come back in order
            parts = [load_file(p) for p in paths]
            self.feature_locations[module_path] = [p["locations"] for p in parts]
            self.feature_activations[module_path] = [p["activations"] for p in parts]
            for p in paths:
                os.remove(p)
        self._spilled.clear()
        for module_path in list(self.feature_locations.keys()):
            self.feature_locations[module_path] = torch.cat(self.feature_locations[module_path], dim=0)
            self.feature_activations[module_path] = torch.cat(self.feature_activations[module_path], dim=0)


class FeatureCache:
    def __init__(self, model, tokenizer, submodule_dict: Dict[str, Sae], batch_size: int,
                 shard_size: int, filters: Optional[Dict[str, Tensor]] = None):
        # LlavaNextForConditionalGeneration wraps the language model (cache.py:104-109)
        if hasattr(model, "language_model") and hasattr(model, "vision_tower"):
            self.llava_model, self.model = model, model.language_model
        else:
            self.llava_model, self.model = None, model
        self.tokenizer = tokenizer
        self.name_to_module = {name: self.model.get_submodule(name) for name in submodule_dict}
        self.module_to_name = {v: k for k, v in self.name_to_module.items()}
        self.submodule_dict = submodule_dict
        self.batch_size = batch_size
        first = next(iter(submodule_dict.values()))
        self.width = first.cfg.num_latents or first.d_in * first.cfg.expansion_factor
        self.cache = Cache(shard_size, filters, batch_size=batch_size)
        if filters is not None:
            self.filter_submodules(filters)

    def filter_submodules(self, filters):
        self.submodule_dict = {m: s for m, s in self.submodule_dict.items() if m in filters}

    def load_token_batches(self, n_tokens: int, tokens: Tensor):
        max_batches = n_tokens // tokens.shape[1]
        tokens = tokens[:max_batches]
        n = len(tokens) // self.batch_size
        return [tokens[self.batch_size * i:self.batch_size * (i + 1), :] for i in range(n)]

    # -- one batch: hook the layers, run the LLM, encode + sparsify what the hooks captured ----------
    def _capture(self, forward_fn):
        buffer: Dict[str, Tensor] = {}

        def hook(module, _, outputs):
            buffer[self.module_to_name[module]] = outputs[0] if isinstance(outputs, tuple) else outputs

        handles = [m.register_forward_hook(hook) for m in self.name_to_module.values()]
        try:
            with torch.no_grad():
                forward_fn()
        finally:
            for h in handles:
                h.remove()
        return buffer

    def _consume(self, buffer: Dict[str, Tensor], batch_number: int, drop_first_token: bool):
        for module_path, hidden in buffer.items():
            if module_path not in self.submodule_dict:
                continue
            sae = self.submodule_dict[module_path]
            if drop_first_token:  # image cache drops the BOS position (cache.py:407-409)
                hidden = hidden[:, 1:, :]
            with torch.no_grad():
                top = sae.encode(hidden)  # fused: replaces pre_acts + topk + zeros/scatter_
            self.cache.add_topk(top.top_acts, top.top_indices, sae.num_latents, batch_number, module_path)

    def run(self, n_tokens: int, tokens):
        from torch.utils.data import DataLoader

        batches = DataLoader(tokens, batch_size=self.batch_size, drop_last=True, shuffle=False)
        total_tokens = 0
        device = next(self.model.parameters()).device
        for batch_number, batch in enumerate(batches):
            total_tokens += n_tokens
            ids = batch["input_ids"].to(device)
            target = self.llava_model if self.llava_model is not None else self.model
            buffer = self._capture(lambda: target(ids))
            self._consume(buffer, batch_number, drop_first_token=False)
        print(f"Total tokens processed: {total_tokens:,}")
        self.cache.save()
        if dist.is_initialized():
            dist.barrier()

    def save(self, save_dir):
        for module_path in self.cache.feature_locations.keys():
            save_file({"locations": self.cache.feature_locations[module_path],
                       "activations": self.cache.feature_activations[module_path]},
                      f"{save_dir}/{module_path}.safetensors")

    def _generate_split_indices(self, n_splits):
        return generate_split_indices(self.width, n_splits)

    def save_splits(self, n_splits: int, save_dir, rank: int, include_split_end: bool = False):
        for module_path in self.cache.feature_locations.keys():
            loc = self.cache.feature_locations[module_path]
            act = self.cache.feature_activations[module_path]
            feats = loc[:, 2]
            module_dir = f"{save_dir}/{module_path}"
            os.makedirs(module_dir, exist_ok=True)
            for start, end in self._generate_split_indices(n_splits):
                hi = end + 1 if include_split_end else end  # reference: feature == end is dropped
                mask = (feats >= start) & (feats < hi)
                save_file({"locations": loc[mask].contiguous(), "activations": act[mask].contiguous()},
                          f"{module_dir}/Rank{rank}_{start}_{end}.safetensors")

    def concate_safetensors(self, n_splits: int, save_dir):
        for module_path in self.cache.feature_locations.keys():
            module_dir = f"{save_dir}/{module_path}"
            for start, end in self._generate_split_indices(n_splits):
                pat = re.compile(r"^Rank(\d+)_{}_{}\.safetensors$".format(start, end))
                parts = sorted((int(m.group(1)), f) for f in os.listdir(module_dir)
                               if (m := pat.match(f)))  # rank order (reference: os.listdir order)
                acts, locs = [], []
                for _, fname in parts:
                    data = load_file(os.path.join(module_dir, fname))
                    acts.append(data["activations"])
                    locs.append(data["locations"])
                    os.remove(os.path.join(module_dir, fname))
                save_file({"locations": torch.cat(locs, dim=0), "activations": torch.cat(acts, dim=0)},
                          f"{module_dir}/{start}_{end}.safetensors")


class FeatureImageCache(FeatureCache):
    """Image variant (cache.py:312-429): `<image>` prompt per image through the LLaVA processor,
    BOS position dropped before the SAE."""

    def __init__(self, model, tokenizer, submodule_dict, batch_size: int, shard_size: int,
                 filters=None, processor=None):
        super().__init__(model, tokenizer, submodule_dict, batch_size, shard_size, filters)
        if processor is None:  # resolved lazily, not at import time as cache.py:321 does
            from transformers import LlavaNextProcessor

            processor = LlavaNextProcessor.from_pretrained("llava-hf/llama3-llava-next-8b-hf")
        self.processor = processor
        self.prompt = "<image>"

    def run(self, n_tokens: int, tokens):
        from torch.utils.data import DataLoader

        def collate_fn(instances: Sequence):
            images = [inst["image"].convert("RGB") for inst in instances]
            sizes = torch.tensor([im.size for im in images]).to(torch.long)
            return dict(images=images, image_sizes=sizes)

        batches = DataLoader(tokens, batch_size=self.batch_size, drop_last=True, shuffle=False,
                             collate_fn=collate_fn, num_workers=0)
        total = 0
        device = next(self.model.parameters()).device
        for batch_number, batch in enumerate(batches):
            inputs = self.processor(text=[self.prompt] * self.batch_size, images=batch["images"],
                                    return_tensors="pt")
            total += self.batch_size
            buffer = self._capture(lambda: self.llava_model(
                input_ids=inputs["input_ids"].to(device),
                pixel_values=inputs["pixel_values"].to(device),
                image_sizes=inputs["image_sizes"].to(device),
                attention_mask=inputs["attention_mask"].to(device)))
            self._consume(buffer, batch_number, drop_first_token=True)
        print(f"Total Images processed: {total:,}")
        self.cache.save()
        if dist.is_initialized():
            dist.barrier()

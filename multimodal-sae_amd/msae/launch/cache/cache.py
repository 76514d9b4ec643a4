"""python -m msae.launch.cache.cache <model> <dataset> --sae_path ... --n_splits ... --save_dir ...

Text feature caching, same command line as `sae_auto_interp.launch.cache.cache`
(launch/cache/cache.py:19-105): torchrun-style DP over the dataset, per-rank split files, rank-0
concat.  The SAE work inside the forward hook runs on the fused HIP path."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from ...config import CacheConfig, parse_cache_config
from ...features import FeatureCache
from ...utils import ddp_setup, load_filter, load_saes, maybe_load_llava_model, shard_offsets


def chunk_and_tokenize(dataset, tokenizer, max_seq_len: int, text_key: str = "text", batch_docs: int = 2048,
                       return_final_batch: bool = False, num_proc: int = 1):
    """GPT-style chunking with the reference's semantics (sae_auto_interp/sae/data.py:16-100), chunk for chunk:

      * documents are taken in batches of 2048 (`Dataset.map(batched=True, batch_size=2048)` there); the texts of a
        batch are JOINED AS STRINGS with the EOS token (an empty first element, so the batch starts with one) and
        tokenised in ONE call with `truncation` + `return_overflowing_tokens` at `min(model_max_length, max_seq_len)`;
      * a slow tokenizer returns one flat id list plus a flat overflow, which is cut into chunks; a fast tokenizer
        returns one row per chunk (special tokens re-added on every row) -- both taken as they come;
      * the last chunk of EVERY batch (ragged almost surely) is dropped unless `return_final_batch`.

    So chunk boundaries and therefore the cache's `row` ids equal the reference's on the same dataset WHEN THE REFERENCE
    MAPS WITH ONE PROCESS (golden fixture g11: 2500 documents, both tokenizer kinds, num_proc = 1).  The reference's
    launcher passes num_proc = cpu_count() // 2 (launch/cache/cache.py:58): `Dataset.map` then cuts the dataset into
    that many contiguous shards first and batches 2048 documents inside each, so a batch never spans a shard border
    and one more ragged chunk is dropped per shard -- `num_proc` below reproduces that split (the same
    `Dataset.shard(contiguous=True)` arithmetic: the first len % n shards hold one more document), still on one host
    process.  Returns a `datasets.Dataset` in torch format with the single
    column `input_ids`, which the caller shards with `.shard(world, rank, contiguous=True)` like the reference
    (launch/cache/cache.py:66).  Only the chunk ids are kept in memory (one batch of texts at a time)."""
    from datasets import Dataset

    chunk = min(tokenizer.model_max_length, max_seq_len)
    sep = tokenizer.eos_token or "<|endoftext|>"
    rows = []
    n_docs, n_sh = len(dataset), max(1, min(int(num_proc), len(dataset)))
    div, mod = divmod(n_docs, n_sh)
    starts = []                                        # (start, stop) of every 2048-document batch, shard by shard
    for sh in range(n_sh):
        lo = sh * div + min(sh, mod)
        hi = lo + div + (1 if sh < mod else 0)
        starts += [(b, min(b + batch_docs, hi)) for b in range(lo, hi, batch_docs)]
    for start, stop in starts:
        texts = dataset[start:stop][text_key]
        enc = tokenizer(sep.join([""] + list(texts)), max_length=chunk, return_attention_mask=False,
                        return_overflowing_tokens=True, truncation=True)
        ids = enc["input_ids"]
        overflow = enc.get("overflowing_tokens") if hasattr(enc, "get") else None
        if overflow:                                   # slow tokenizer: flat lists
            pieces = [list(ids)] + [list(overflow[i:i + chunk]) for i in range(0, len(overflow), chunk)]
        elif len(ids) and isinstance(ids[0], (list, tuple)):
            pieces = [list(r) for r in ids]            # fast tokenizer: one row per chunk
        else:
            pieces = [list(ids)]
        if not return_final_batch:
            pieces = pieces[:-1]
        if not pieces:
            raise ValueError("Not enough data to create a single complete batch. Either allow the final batch to be "
                             "returned, or supply more data.")     # data.py:80-85
        rows.extend(pieces)
    return Dataset.from_dict({"input_ids": rows}).with_format("torch", columns=["input_ids"])


def main(cfg: CacheConfig):
    from datasets import load_dataset
    from transformers import AutoTokenizer

    ddp, rank, world = ddp_setup(timeout_s=18000)
    dtype = torch.bfloat16 if torch.cuda.is_bf16_supported() else "auto"
    model, _ = maybe_load_llava_model(cfg.model, rank, dtype, cfg.hf_token)
    tokenizer = AutoTokenizer.from_pretrained(cfg.model, token=cfg.hf_token)
    dataset = load_dataset(cfg.dataset, split=cfg.split)
    filters = load_filter(cfg.filters_path, device=model.device) if cfg.filters_path else None
    # the reference's default: num_proc = cpu_count() // 2 (sae_auto_interp/sae/data.py:21) -- the same chunks, hence the same
    # cache row ids, as the reference produces on this machine
    num_proc = max(1, (os.cpu_count() or 2) // 2)
    try:
        dataset = chunk_and_tokenize(dataset, tokenizer, max_seq_len=cfg.ctx_len, num_proc=num_proc)
    except ValueError as e:
        # ONLY the chunker's own "a shard without one complete chunk" (a dataset too small for num_proc shards): the
        # reference stops here (sae/data.py); one shard gives the run a chance -- loudly, because the chunk boundaries, hence
        # the cache's row ids, are then those of num_proc=1 and not what the reference would produce on this machine
        # (ADVICE r4).  Any other ValueError (the tokenizer's) propagates.
        if num_proc == 1 or "Not enough data to create a single complete batch" not in str(e):
            raise
        import warnings

        warnings.warn(f"launch.cache: the dataset is too small for num_proc={num_proc} shards ({e}); re-chunking with "
                      "num_proc=1 -- the cache's row ids follow the single-process chunk boundaries")
        dataset = chunk_and_tokenize(dataset, tokenizer, max_seq_len=cfg.ctx_len, num_proc=1)
    shard_size = 0
    if ddp:
        dist.barrier()
        dataset = dataset.shard(world, rank, contiguous=True)       # contiguous chunks (cache.py:66)
        shard_size = sum(shard_offsets(len(dataset), model.device)[:rank])   # all_gather_into_tensor (cache.py:67-75)
    saes = load_saes(cfg.sae_path, filters=filters, device=model.device)
    cache = FeatureCache(model, tokenizer, saes, batch_size=cfg.batch_size, shard_size=shard_size,
                         filters=filters)
    if ddp:
        dist.barrier()
    cache.run(cfg.ctx_len, dataset)
    cache.save_splits(n_splits=cfg.n_splits, save_dir=cfg.save_dir, rank=rank)
    if ddp:
        dist.barrier()
    if rank == 0:
        cache.concate_safetensors(n_splits=cfg.n_splits, save_dir=cfg.save_dir)
    if ddp:
        dist.barrier()


if __name__ == "__main__":
    main(parse_cache_config())

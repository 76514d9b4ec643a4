"""python -m msae.launch.cache.cache <model> <dataset> --sae_path ... --n_splits ... --save_dir ...

Text feature caching, same command line as `sae_auto_interp.launch.cache.cache`
(launch/cache/cache.py:19-105): torchrun-style DP over the dataset, per-rank split files, rank-0
concat.  The SAE work inside the forward hook runs on the fused HIP path."""
from __future__ import annotations

import torch
import torch.distributed as dist

from ...config import CacheConfig, parse_cache_config
from ...features import FeatureCache
from ...utils import ddp_setup, load_filter, load_saes, maybe_load_llava_model, shard_offsets


def chunk_and_tokenize(dataset, tokenizer, max_seq_len: int, text_key: str = "text"):
    """GPT-style chunking as the reference's sae/data.py:16-100 does it: documents joined with EOS
    (the stream starts with one), cut into chunks of exactly `max_seq_len` ids, the ragged final chunk
    dropped.  Returns a `datasets.Dataset` in torch format with the single column `input_ids`, so the
    caller can `.shard(world, rank, contiguous=True)` it exactly like the reference (cache.py:66)."""
    from datasets import Dataset

    eos = tokenizer.eos_token_id
    buf, chunks = [eos], []
    for row in dataset:
        buf.extend(tokenizer(row[text_key], add_special_tokens=False)["input_ids"] + [eos])
        while len(buf) >= max_seq_len:
            chunks.append(buf[:max_seq_len])
            buf = buf[max_seq_len:]
    if not chunks:
        raise ValueError("Not enough data to create a single complete batch.")   # data.py:80-85
    return Dataset.from_dict({"input_ids": chunks}).with_format("torch", columns=["input_ids"])


def main(cfg: CacheConfig):
    from datasets import load_dataset
    from transformers import AutoTokenizer

    ddp, rank, world = ddp_setup(timeout_s=18000)
    dtype = torch.bfloat16 if torch.cuda.is_bf16_supported() else "auto"
    model, _ = maybe_load_llava_model(cfg.model, rank, dtype, cfg.hf_token)
    tokenizer = AutoTokenizer.from_pretrained(cfg.model, token=cfg.hf_token)
    dataset = load_dataset(cfg.dataset, split=cfg.split)
    filters = load_filter(cfg.filters_path, device=model.device) if cfg.filters_path else None
    dataset = chunk_and_tokenize(dataset, tokenizer, max_seq_len=cfg.ctx_len)
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

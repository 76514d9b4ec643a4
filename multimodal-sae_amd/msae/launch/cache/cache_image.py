"""python -m msae.launch.cache.cache_image <model> <dataset> --sae_path ... --n_splits ... (README
command of the reference, launch/cache/cache_image.py:24-104): image feature caching."""
from __future__ import annotations

import torch
import torch.distributed as dist

from ...config import CacheConfig, parse_cache_config
from ...features import FeatureImageCache
from ...utils import ddp_setup, load_filter, load_saes, maybe_load_llava_model, shard_offsets


def main(cfg: CacheConfig):
    from datasets import load_dataset
    from transformers import AutoTokenizer

    ddp, rank, world = ddp_setup(timeout_s=18000)
    dtype = torch.bfloat16 if torch.cuda.is_bf16_supported() else "auto"
    model, processor = maybe_load_llava_model(cfg.model, rank, dtype, cfg.hf_token)
    tokenizer = AutoTokenizer.from_pretrained(cfg.model, token=cfg.hf_token)
    dataset = load_dataset(cfg.dataset, split=cfg.split)
    filters = load_filter(cfg.filters_path, device=model.device) if cfg.filters_path else None
    shard_size = 0
    if ddp:
        dist.barrier()
        dataset = dataset.shard(world, rank, contiguous=True)
        shard_size = sum(shard_offsets(len(dataset), model.device)[:rank])
    saes = load_saes(cfg.sae_path, filters=filters, device=model.device)
    cache = FeatureImageCache(model, tokenizer, saes, batch_size=cfg.batch_size, shard_size=shard_size,
                              processor=processor, filters=filters)
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

"""Loaders shared by the launch scripts (reference: sae_auto_interp/utils.py:44-140)."""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

from .sae import Sae


def load_filter(path: str, device: str = "cuda:0") -> Dict[str, torch.Tensor]:
    """{module: [feature idx, ...]} JSON -> {module: int64 tensor} (utils.py:44-48)."""
    with open(path) as f:
        filt = json.load(f)
    return {key: torch.tensor(value, device=device) for key, value in filt.items()}


def load_saes(sae_path: str, filters: Optional[Dict[str, torch.Tensor]] = None,
              device="cuda:0") -> Dict[str, Sae]:
    """Local directory or hub repo -> {hookpoint: Sae}; with `filters` only the filtered
    hookpoints are loaded (utils.py:106-127)."""
    local = os.path.exists(sae_path)
    if filters is None:
        return Sae.load_many(sae_path, local=local, device=device)
    if local:
        return {m: Sae.load_from_disk(os.path.join(sae_path, m), device=device) for m in filters}
    return {m: Sae.load_from_hub(sae_path, m, device=device) for m in filters}


def load_single_sae(sae_path: str, module_name: str, device="cuda:0") -> Sae:
    if os.path.exists(sae_path):
        return Sae.load_from_disk(os.path.join(sae_path, module_name), device=device)
    return Sae.load_from_hub(sae_path, module_name, device=device)


def maybe_load_llava_model(model_name: str, rank: int, dtype, hf_token=None):
    """HF model (+ processor for LLaVA-NeXT) on cuda:{rank} (utils.py:68-88).  The LLM itself stays
    stock PyTorch-ROCm; only the SAE path behind its forward hook is native."""
    from transformers import AutoModel, LlavaNextForConditionalGeneration, LlavaNextProcessor

    kw = dict(device_map={"": f"cuda:{rank}"}, torch_dtype=dtype, token=hf_token)
    if "llava" in model_name:
        return (LlavaNextForConditionalGeneration.from_pretrained(model_name, **kw),
                LlavaNextProcessor.from_pretrained(model_name))
    return AutoModel.from_pretrained(model_name, **kw), None


def ddp_setup(timeout_s: Optional[int] = None):
    """torchrun-style process setup used by every launch script (launch/cache/cache.py:22-31):
    one process per GPU, backend "nccl" (RCCL on ROCm).  Returns (ddp, rank, world)."""
    import datetime

    import torch.distributed as dist

    local_rank = os.environ.get("LOCAL_RANK")
    if local_rank is None:
        return False, 0, 1
    torch.cuda.set_device(int(local_rank))
    kw = {} if timeout_s is None else {"timeout": datetime.timedelta(seconds=timeout_s)}
    dist.init_process_group("nccl", **kw)
    return True, int(local_rank), dist.get_world_size()


def shard_offsets(n_local: int, device) -> list:
    """all-gather of the per-rank dataset shard lengths (launch/cache/cache.py:68-75)."""
    import torch.distributed as dist

    lens = torch.zeros(dist.get_world_size(), dtype=torch.int, device=device)
    dist.all_gather_into_tensor(lens, torch.tensor([n_local], dtype=torch.int, device=device))
    return lens.cpu().tolist()

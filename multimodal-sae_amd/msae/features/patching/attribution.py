"""`Attribution` -- per-feature attribution patching (reference features/patching/attribution.py:25-189).

The reference scores ONE feature per iteration: two forwards of the LLM with the SAE reconstruction
spliced in (clean, and with the feature's latent zeroed), one backward of the metric, and
`((clean - corrupted) * corrupted.grad).sum(-1)` per hooked module (attribution.py:133-183).

`get_attribution(indices)` reproduces exactly that (`method="exact"`; the clean run, identical for
every feature, is done once).  `method="batched"` is the MI355X-native shape of the same quantity:
the reconstruction is linear in the latents, so with r the token's (k+1)-th latent (the one that
enters the top-k when a member is zeroed)

    clean - corrupted_f = act_f W_dec[f] - act_r W_dec[r]          exactly, for every active f,
    score[t, f] = act_f <g_t, W_dec[f]> - act_r <g_t, W_dec[r]>,   g = d metric / d reconstruction

and ALL features come out of ONE forward + ONE backward of the LLM plus one call of the decoder-backward
primitive (msae_decode_bwd_acts_f32, reference sae/kernels.py:287-400) over the k+1 latents -- instead
of 2 N forwards + N backwards.  The only approximation is that g is taken at the clean run rather than
at each corrupted run (the usual attribution-patching linearisation; tests state the tolerance).
"""
from __future__ import annotations

import collections
import json
import os
from functools import partial
from typing import Dict, List, Optional, Union

import torch
import torch.distributed as dist
from torch import Tensor

from ... import ops
from ...sae import Sae
from .utils import get_logit_diff, get_model_backward_cache_with_sae, get_model_forward_cache_with_sae

os.environ.setdefault("TOKENIZERS_PARALLELISM", "false")


class Attribution:
    def __init__(self, model, tokenizer, sae_path: str, data_path: str, selected_sae: str = None,
                 image_processor=None) -> None:
        from PIL import Image

        self.model = model
        self.image_processor = image_processor
        if selected_sae is not None:
            if not os.path.exists(sae_path):
                sae = Sae.load_from_hub(sae_path, hookpoint=selected_sae, device=model.device)
            else:
                sae = Sae.load_from_disk(os.path.join(sae_path, selected_sae), device=model.device)
            sae_dict = {selected_sae: sae}
        else:
            sae_dict = Sae.load_many(sae_path, local=os.path.exists(sae_path), device=model.device)
        self.data_path = data_path
        with open(data_path, "r") as f:
            self.data = json.load(f)     # [{"prompt", "answer", "baseline", "image"}, ...] (attribution.py:61-69)
        prompt, answer, images, image_sizes = [], [], [], []
        for item in self.data:
            prompt.append(item["prompt"])
            answer.append([str(item["answer"]), str(item["baseline"])])
            image = Image.open(item["image"])
            images.append(image)
            image_sizes.append([image.size[0], image.size[1]])
        pixel_values = image_processor(images, do_pad=True, return_tensors="pt")["pixel_values"] \
            .to(model.device).to(model.dtype)
        prompt_ids = tokenizer(prompt, return_tensors="pt")["input_ids"].to(model.device)[:, 1:]
        answer_ids = torch.tensor([[tokenizer.convert_tokens_to_ids(a[0]), tokenizer.convert_tokens_to_ids(a[1])]
                                   for a in answer]).to(model.device)
        self._setup(sae_dict, {"input_ids": prompt_ids, "pixel_values": pixel_values, "image_sizes": image_sizes,
                               "attention_mask": prompt_ids.ne(tokenizer.pad_token_id)}, answer_ids)

    @classmethod
    def from_parts(cls, model, sae_dict: Dict[str, Sae], inputs: dict, answer_ids: Tensor) -> "Attribution":
        """Build from already-tokenised inputs (tests, programmatic use): `inputs` is what the model's
        forward takes, `answer_ids` [B, 2] = (correct token id, baseline token id)."""
        self = cls.__new__(cls)
        self.model, self.image_processor, self.data_path, self.data = model, None, None, None
        self._setup(sae_dict, inputs, answer_ids)
        return self

    def _setup(self, sae_dict, inputs, answer_ids):
        self.sae_dict = sae_dict
        for sae in sae_dict.values():
            sae.eval()
        self.inputs = inputs
        self.prompt_ids, self.attention_mask = inputs.get("input_ids"), inputs.get("attention_mask")
        self.pixel_values, self.image_sizes = inputs.get("pixel_values"), inputs.get("image_sizes")
        self.answer_ids = answer_ids
        lm = getattr(self.model, "language_model", self.model)   # a pure llama model has no .language_model
        self.name_to_module = {name: lm.get_submodule(name) for name in sae_dict.keys()}
        self.module_to_name = {v: k for k, v in self.name_to_module.items()}
        self.metric = partial(get_logit_diff, answer_token_indices=answer_ids)

    # ------------------------------------------------------------------------------------------------------
    def _default_indices(self) -> Tensor:
        sae = next(iter(self.sae_dict.values()))
        return torch.arange(sae.num_latents)      # (the reference reads a misspelt cfg field here, attribution.py:121)

    def get_attribution(self, indices: Union[List[int], Tensor, None] = None, method: str = "exact"):
        """-> {module name: [fp16 CPU tensor [B, S] per requested feature]} (attribution.py:116-189)."""
        if indices is None:
            indices = self._default_indices()
        indices = [int(i) for i in (indices.tolist() if isinstance(indices, Tensor) else indices)]
        if method == "batched":
            out = self._batched(indices)
        elif method == "exact":
            out = self._per_feature(indices)
        else:
            raise ValueError(f"unknown method {method!r}")
        if dist.is_initialized() and os.environ.get("LOCAL_RANK") is not None:
            dist.barrier()
        return out

    def _per_feature(self, indices: List[int]):
        attribution_dict = collections.defaultdict(list)
        with torch.no_grad():    # the clean run does not depend on the feature and is never differentiated
            _, clean_cache = get_model_forward_cache_with_sae(self.model, self.inputs, self.sae_dict,
                                                              self.module_to_name)
        for idx in indices:
            corrupted_logits, corrupted_cache = get_model_forward_cache_with_sae(
                self.model, self.inputs, self.sae_dict, self.module_to_name, off_features=idx)
            for tensor in corrupted_cache.values():
                tensor.retain_grad()
            get_model_backward_cache_with_sae(logits=corrupted_logits, metrics=self.metric)
            for name in self.sae_dict.keys():
                attribution = (clean_cache[name] - corrupted_cache[name]) * corrupted_cache[name].grad
                attribution_dict[name].append(attribution.detach().sum(dim=-1).cpu())
            self._zero_param_grads()
        return attribution_dict

    def _zero_param_grads(self):
        for sae in self.sae_dict.values():
            for p in sae.parameters():
                p.grad = None
        for p in self.model.parameters():
            p.grad = None

    def batched_scores(self):
        """ONE forward + backward -> {module: (top_indices [B, S, k] int64, scores [B, S, k] f32)}: the
        attribution of every ACTIVE feature of every token (inactive ones score 0)."""
        latents: dict = {}
        logits, cache = get_model_forward_cache_with_sae(self.model, self.inputs, self.sae_dict,
                                                         self.module_to_name, keep_latents=latents, extra_k=1)
        for tensor in cache.values():
            tensor.retain_grad()
        get_model_backward_cache_with_sae(logits=logits, metrics=self.metric)
        out = {}
        with torch.no_grad():
            for name, sae in self.sae_dict.items():
                va, ia = latents[name]                               # [T, k + 1]
                g = cache[name].grad
                B, S, d = g.shape
                k = sae.cfg.k
                dots, _ = ops.decode_bwd(ia, va, sae.W_dec, g.reshape(-1, d).float().contiguous(), True, False)
                contrib = va * dots                                  # act_j <g, W_dec[j]>
                scores = (contrib[:, :k] - contrib[:, k:]) * (va[:, :k] > 0)
                out[name] = (ia[:, :k].reshape(B, S, k), scores.reshape(B, S, k))
        self._zero_param_grads()
        return out

    def _batched(self, indices: List[int]):
        attribution_dict = collections.defaultdict(list)
        for name, (idx, scores) in self.batched_scores().items():
            sae = self.sae_dict[name]
            B, S, k = idx.shape
            slot = torch.full((sae.num_latents,), -1, dtype=torch.long, device=idx.device)
            slot[torch.tensor(indices, device=idx.device)] = torch.arange(len(indices), device=idx.device)
            dense = torch.zeros(len(indices), B * S, dtype=torch.float32, device=idx.device)
            where = slot[idx.reshape(-1)]
            tok = torch.arange(B * S, device=idx.device).repeat_interleave(k)
            hit = where >= 0
            dense.index_put_((where[hit], tok[hit]), scores.reshape(-1)[hit], accumulate=True)
            dense = dense.to(torch.float16).view(len(indices), B, S).cpu()
            attribution_dict[name] = [dense[i] for i in range(len(indices))]
        return attribution_dict

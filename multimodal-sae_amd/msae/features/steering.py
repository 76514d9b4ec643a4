"""SteeringController -- per-feature clamped generation (reference features/steering.py:13-128).
The forward hook is `msae.features.hooks.clamp_features_max` (fused encode with the clamp applied
inside the kernel); generation itself is HF `generate`."""
from __future__ import annotations

import os
from typing import List

import torch

from ..sae import Sae
from .hooks import clamp_features_max


class SteeringController:
    """`sae`: an `Sae`, or a feature-sharded `msae.parallel.ShardedSae` engine -- then EVERY rank of the engine's
    group must run the same controller on the same prompt and feature list (they meet in the engine's collectives
    at every hooked forward)."""

    def __init__(self, sae, module_name: str, feature_idx: List[int], model, processor,
                 prompt: str, image_path: str = None, k: float = 50):
        self.sae, self.feature_idx, self.model, self.k = sae, feature_idx, model, k
        self.module_name, self.processor = module_name, processor
        self.hooked_module = model.language_model.get_submodule(module_name)
        local_rank = os.environ.get("LOCAL_RANK")
        self.ddp = local_rank is not None
        self.rank = int(local_rank) if local_rank is not None else 0
        content = [{"type": "text", "text": prompt}]
        self.image = None
        if image_path is not None:
            from PIL import Image

            self.image = Image.open(image_path)
            content.append({"type": "image"})
        self.prompt = processor.apply_chat_template([{"role": "user", "content": content}],
                                                    add_generation_prompt=True)
        self.inputs = processor(images=self.image, text=self.prompt, return_tensors="pt").to(model.device)

    def clamp_features_max(self, sae, feature: int, hooked_module, k: float = 10):
        return clamp_features_max(sae, feature, hooked_module, k=k)

    def _generate(self) -> str:
        kw = {}
        world = getattr(self.sae, "world", 1)
        if not isinstance(self.sae, Sae) and world > 1:
            # feature-sharded engine: all ranks meet in its collectives at every hooked forward, so their generation
            # loops must take the same number of steps.  Same RNG state on every rank (the model's generation_config
            # may sample), and HF's `synced_gpus`: a rank whose sequence hit EOS keeps stepping until all have.
            # -- seeded ONCE per controller (re-seeding in every _generate made every sampled steering
            # generation replay the same random stream); later generations continue the ranks' common stream
            if not getattr(self, "_ranks_seeded", False):
                torch.manual_seed(0x5AE)
                self._ranks_seeded = True
            kw["synced_gpus"] = True
        with torch.no_grad():
            output = self.model.generate(**self.inputs, max_new_tokens=512, **kw)
        cont = output[:, self.inputs["input_ids"].shape[-1]:]
        return self.processor.batch_decode(cont, skip_special_tokens=True)[0]

    def run(self) -> dict:
        original = self._generate()
        results = {}
        for idx in self.feature_idx:
            handles = self.clamp_features_max(self.sae, idx, self.hooked_module, k=self.k)
            try:
                clamped = self._generate()
            finally:
                for h in handles:
                    h.remove()
            results[f"{self.module_name}_feature{idx}"] = {
                "original_resps": original, "clamped_resps": clamped, "idx": idx}
        return results

"""python -m msae.launch.features.attribution_patching <model> --data_path ... --sae_path ...
--selected_sae layers.24 --save_dir ...   (reference launch/features/attribution_patching.py:15-80).

One process per GPU under torchrun; the feature axis is chunked over the ranks, rank 0 gathers and
writes `<save_dir>/<model>_<selected_sae>.safetensors` with one [n_features * B, S] tensor per hooked
module, as the reference does.  `--method batched` computes every feature from one forward + backward
(msae/features/patching/attribution.py) instead of 2 forwards + 1 backward per feature.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from ...config import AttributionConfig, parse_attribution_config
from ...features.patching import Attribution
from ...utils import ddp_setup, maybe_load_llava_model


def main(cfg: AttributionConfig):
    from safetensors.torch import save_file
    from transformers import AutoTokenizer

    ddp, rank, world = ddp_setup()
    tokenizer = AutoTokenizer.from_pretrained(cfg.model)
    model, processor = maybe_load_llava_model(cfg.model, rank, torch.float16, None)
    attribution = Attribution(model, tokenizer, sae_path=cfg.sae_path, data_path=cfg.data_path,
                              selected_sae=cfg.selected_sae,
                              image_processor=processor.image_processor if processor is not None else None)
    if ddp:
        sae = next(iter(attribution.sae_dict.values()))
        indices = torch.arange(sae.num_latents).chunk(world)[rank]
        dist.barrier()
        attribution_dict = attribution.get_attribution(indices, method=cfg.method)
        gathered = [None for _ in range(world)]
        dist.all_gather_object(gathered, dict(attribution_dict))
        if rank == 0:
            attribution_dict = {k: list(v) for k, v in gathered[0].items()}
            for part in gathered[1:]:
                for name, vals in part.items():
                    attribution_dict[name].extend(vals)
        dist.barrier()
    else:
        attribution_dict = attribution.get_attribution(method=cfg.method)
    if rank == 0:
        out = {k: torch.concatenate(v, dim=0) for k, v in attribution_dict.items()}
        os.makedirs(cfg.save_dir, exist_ok=True)
        output_file = os.path.join(cfg.save_dir,
                                   f"{cfg.model.split('/')[-1]}_{cfg.selected_sae.replace('.', '_')}.safetensors")
        save_file(out, output_file)
        return output_file


if __name__ == "__main__":
    main(parse_attribution_config())

"""python -m msae.launch.features.steering -m <model> -t <text> [-i <image>] --sae-path ...
--filters ... -k <clamp> -s <save dir>   (reference launch/features/steering.py:18-113)."""
from __future__ import annotations

import argparse
import json
import os

import torch
import torch.distributed as dist

from ...features.steering import SteeringController
from ...utils import ddp_setup, load_filter, load_saes, maybe_load_llava_model


def parse_argument(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--model", "-m", type=str, default="llava-hf/llama3-llava-next-8b-hf",
                   help="The model name of your trained model")
    p.add_argument("--image-path", "-i", type=str, default=None, help="The path to your image")
    p.add_argument("--text", "-t", type=str, help="The text you want to ask the model")
    p.add_argument("--sae-path", type=str, help="The path to your sae, can be hub or local")
    p.add_argument("--filters", type=str, help="The filters path")
    p.add_argument("--clamp-value", "-k", type=float, default=50, help="The clamping value")
    p.add_argument("--save-dir", "-s", default="./results/steering",
                   help="The path to save your steering result")
    p.add_argument("--shard-sae", "--shard_sae", action="store_true",
                   help="(new) under torchrun: shard the SAE's feature axis over the ranks instead of the feature "
                        "list -- every rank runs every generation, each streams N/G rows of the encoder per step "
                        "(the latency mode for wide SAEs); results are identical")
    p.add_argument("--shard-mode", default="topk", choices=["topk", "candidates"],
                   help="exchange scheme of --shard-sae (msae.parallel.ShardedSae)")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_argument(argv)
    ddp, rank, world = ddp_setup()
    model, processor = maybe_load_llava_model(args.model, rank=rank, dtype=torch.float16, hf_token=None)
    filters = load_filter(args.filters, device="cpu")
    sae_dict = load_saes(args.sae_path, filters, device=f"cuda:{rank}")
    for module_name, sae in sae_dict.items():
        feats = filters[module_name]
        shard = args.shard_sae and ddp and sae.num_latents % world == 0
        if shard:       # one SAE over all ranks: every rank walks the whole feature list in step
            from ...parallel import ShardedSae

            # broadcast_input: every rank runs its own LLM forward; rank 0's hidden state is the one all shards encode
            # (nondeterministic kernels / sampling cannot make the ranks merge results of different inputs), and the
            # controller keeps the ranks' generation loops in lockstep (SteeringController._generate)
            engine = ShardedSae.from_sae(sae, rank=rank, world=world, group=dist.group.WORLD, mode=args.shard_mode,
                                         broadcast_input=True)
            feature_idx = feats.cpu().tolist()
        else:           # the reference's split: every rank its own slice of the feature list (steering.py:70-75)
            engine = sae
            feature_idx = (feats.tensor_split(world)[rank] if ddp else feats).cpu().tolist()
        result = SteeringController(sae=engine, module_name=module_name, feature_idx=feature_idx,
                                    prompt=args.text, model=model, processor=processor,
                                    image_path=args.image_path, k=args.clamp_value).run()
        if ddp and not shard:
            gathered = [None] * world
            dist.gather_object(result, gathered if rank == 0 else None, dst=0)
            if rank == 0:
                for r in gathered:
                    result.update(r)
        if rank == 0:
            os.makedirs(args.save_dir, exist_ok=True)
            with open(os.path.join(args.save_dir, f"{module_name}.json"), "w", encoding="utf-8") as f:
                json.dump(result, f, indent=4, ensure_ascii=False)
        if ddp:
            dist.barrier()


if __name__ == "__main__":
    main()

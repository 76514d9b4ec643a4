"""Executed under `python -m torch.distributed.run --nproc-per-node 1 ... tests/launch_runner.py <dir>`
by tests/test_gpu_dropin.py: runs the drop-in launch entry points (msae.launch.cache.cache,
msae.launch.cache.cache_image, msae.launch.features.steering, msae.launch.features.attribution_patching)
through their real `main()` -- DDP setup over RCCL, sharding, hooks, file writers -- with the HF model /
tokenizer / dataset loaders replaced by the stand-ins of tests/fakes.py (no checkpoint, no network)."""
import json
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for p in (REPO, REPO / "tests", REPO / "multimodal-sae_amd"):
    sys.path.insert(0, str(p))

import torch

import fakes
import synth


def build_sae_checkpoint(root: Path, hookpoints, d=64, N=1024, k=8, seed=9):
    from msae import Sae, SaeConfig

    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed)
    for hp in hookpoints:
        sae = Sae(d, SaeConfig(num_latents=N, k=k), device="cpu")
        with torch.no_grad():
            sae.encoder.weight.copy_(torch.from_numpy(W_enc)); sae.encoder.bias.copy_(torch.from_numpy(b_enc))
            sae.W_dec.copy_(torch.from_numpy(W_dec)); sae.b_dec.copy_(torch.from_numpy(b_dec))
        sae.save_to_disk(root / hp)


def main():
    out = Path(sys.argv[1])
    rank = int(os.environ.get("LOCAL_RANK", "0"))
    vocab = 40
    # ---- stand-ins for everything that needs a checkpoint or the network
    import datasets
    import transformers

    import msae.utils as mu

    def fake_model_loader(model_name, rank, dtype, hf_token=None):
        return fakes.TinyLlava(vocab=vocab).to(f"cuda:{rank}"), fakes.FakeProcessor(vocab)

    transformers.AutoTokenizer.from_pretrained = classmethod(lambda cls, *a, **k: fakes.FakeSlowTokenizer(vocab))

    def fake_load_dataset(name, split="train", **kw):
        if name == "images":
            return fakes.FakeImageDataset(6)
        words = [f"w{i}" for i in range(23)]
        return datasets.Dataset.from_dict({"text": [" ".join(words[(i + j) % 23] for j in range(9)) for i in range(30)]})

    datasets.load_dataset = fake_load_dataset
    sae_dir = out / "saes"
    if rank == 0:
        build_sae_checkpoint(sae_dir, ["layers.1"])
        (out / "filters.json").write_text(json.dumps({"layers.1": [3, 77, 300]}))
    import msae.launch.cache.cache as lc
    import msae.launch.cache.cache_image as lci
    import msae.launch.features.attribution_patching as lap
    import msae.launch.features.steering as ls
    from msae.config import AttributionConfig, CacheConfig

    for mod in (lc, lci, ls, lap):
        mod.maybe_load_llava_model = fake_model_loader
    # DDP is initialised by the first main(); the later ones must not initialise it again
    import torch.distributed as dist

    real_setup = mu.ddp_setup
    state = {}

    def setup_once(timeout_s=None):
        if "v" not in state:
            state["v"] = real_setup(timeout_s)
        return state["v"]

    for mod in (lc, lci, ls, lap):
        mod.ddp_setup = setup_once

    lc.main(CacheConfig(model="llava-tiny", dataset="text", sae_path=str(sae_dir), batch_size=2, n_splits=4,
                        ctx_len=16, save_dir=str(out / "cache_text")))
    lci.main(CacheConfig(model="llava-tiny", dataset="images", sae_path=str(sae_dir), batch_size=2, n_splits=4,
                         ctx_len=16, save_dir=str(out / "cache_image")))
    ls.main(["-m", "llava-tiny", "-t", "describe", "--sae-path", str(sae_dir), "--filters", str(out / "filters.json"),
             "-k", "50", "-s", str(out / "steering")])
    ls.main(["-m", "llava-tiny", "-t", "describe", "--sae-path", str(sae_dir), "--filters", str(out / "filters.json"),
             "-k", "50", "-s", str(out / "steering_sharded"), "--shard-sae"])
    # attribution: two prompts with images on disk
    from PIL import Image

    data = []
    for i in range(2):
        img = out / f"img{i}.png"
        Image.new("RGB", (8, 6), (10 * i, 20, 30)).save(img)
        data.append({"prompt": f"what is w{i} w{i + 1} w{i + 2}", "answer": "w5", "baseline": "w9", "image": str(img)})
    (out / "attr.json").write_text(json.dumps(data))
    for method in ("exact", "batched"):
        lap.main(AttributionConfig(model="llava-tiny", data_path=str(out / "attr.json"), sae_path=str(sae_dir),
                                   selected_sae="layers.1", save_dir=str(out / f"attribution_{method}"), method=method))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        (out / "DONE").write_text("ok")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE ITSELF on CPU.

Runs only in the build container, where /root/reference exists (it does not exist on the GPU
box; nothing in tests/, smoke() or bench.py reads it at run time).  The fixtures are data only:
seeds + small arrays of inputs and the reference's outputs.  Usage:

    SAE_DISABLE_TRITON=1 python tests/golden/make_golden.py [--full]

`--full` additionally writes the d=4096 / N=131072 fixture (needs ~12 GB RAM, ~1 min).

Import recipe: stub the packages the reference imports but this image
lacks (simple_parsing, natsort, torchtyping), neutralise the import-time
LlavaNextProcessor.from_pretrained default argument (features/cache.py:321), and import
sae_auto_interp.features.cache under a synthetic parent package so features/__init__.py (which
needs blobfile/orjson/torchvision) is not executed.
"""
from __future__ import annotations

import argparse
import dataclasses
import importlib.util
import os
import sys
import tempfile
import types
from pathlib import Path

os.environ["SAE_DISABLE_TRITON"] = "1"  # triton_decode raises "0 active drivers" on CPU tensors

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
OUT = HERE                      # where the fixtures are written (--out: a scratch directory, tests/test_golden_recipe.py)
REPO = HERE.parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(REPO / "tests"))
import synth  # noqa: E402


def _install_stubs():
    sp = types.ModuleType("simple_parsing")

    class Serializable:
        def to_dict(self):
            return dataclasses.asdict(self)

    sp.Serializable = Serializable
    sp.list_field = lambda *a, **k: dataclasses.field(default_factory=lambda: list(a))
    sp.field = lambda *a, default=None, **k: dataclasses.field(default=default)
    sp.parse = lambda *a, **k: None
    sys.modules["simple_parsing"] = sp
    ns = types.ModuleType("natsort")
    ns.natsorted = lambda seq, key=None: sorted(seq, key=key)
    sys.modules["natsort"] = ns
    tt = types.ModuleType("torchtyping")

    class _TT:
        def __class_getitem__(cls, item):
            return torch.Tensor

    tt.TensorType = _TT
    sys.modules["torchtyping"] = tt
    import transformers

    transformers.LlavaNextProcessor.from_pretrained = classmethod(lambda cls, *a, **k: None)


def _import_reference():
    _install_stubs()
    sys.path.insert(0, str(REF))
    from sae_auto_interp.sae import Sae, SaeConfig  # noqa
    from sae_auto_interp.sae.utils import eager_decode  # noqa

    # features.cache without features/__init__.py
    pkg = types.ModuleType("sae_auto_interp.features")
    pkg.__path__ = [str(REF / "sae_auto_interp" / "features")]
    sys.modules["sae_auto_interp.features"] = pkg
    spec = importlib.util.spec_from_file_location(
        "sae_auto_interp.features.cache", REF / "sae_auto_interp" / "features" / "cache.py")
    cache_mod = importlib.util.module_from_spec(spec)
    sys.modules["sae_auto_interp.features.cache"] = cache_mod
    spec.loader.exec_module(cache_mod)
    return Sae, SaeConfig, eager_decode, cache_mod


def _make_ref_sae(Sae, SaeConfig, d, N, k, seed, multi_topk=False):
    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed)
    sae = Sae(d, SaeConfig(num_latents=N, k=k, multi_topk=multi_topk), device="cpu")
    with torch.no_grad():
        sae.encoder.weight.copy_(torch.from_numpy(W_enc))
        sae.encoder.bias.copy_(torch.from_numpy(b_enc))
        sae.W_dec.copy_(torch.from_numpy(W_dec))
        sae.b_dec.copy_(torch.from_numpy(b_dec))
    return sae


def _canon(vals: torch.Tensor, idx: torch.Tensor):
    """Reorder a top-k result to (value desc, index asc)."""
    v, i = vals.numpy().astype(np.float32), idx.numpy().astype(np.int64)
    order = np.lexsort((i, -v.astype(np.float64)), axis=-1)
    return np.take_along_axis(v, order, -1), np.take_along_axis(i, order, -1).astype(np.int32)


def encode_decode_fixture(Sae, SaeConfig, name, d, N, ks, T, wseed, xseed):
    out = {"d": d, "N": N, "T": T, "wseed": wseed, "xseed": xseed, "ks": np.array(ks)}
    x = torch.from_numpy(synth.activations(T, d, xseed)).to(torch.bfloat16)
    sae = _make_ref_sae(Sae, SaeConfig, d, N, ks[0], wseed)
    with torch.no_grad():
        pre = sae.pre_acts(x)  # sae.py:172
        out["pre_slice"] = pre[:8, :256].numpy()
        out["pre_rowsum"] = pre.double().sum(-1).numpy()
        out["pre_nnz"] = (pre > 0).sum(-1).numpy()
        for k in ks:
            sae.cfg.k = k
            top = sae.select_topk(pre)  # sae.py:179
            v, i = _canon(top.top_acts, top.top_indices)
            out[f"k{k}_acts"], out[f"k{k}_idx"] = v, i
            # gap between the k-th and (k+1)-th pre-activation: rows where index equality is
            # tolerance-dependent are identifiable
            kk = pre.topk(k + 1, sorted=True).values
            out[f"k{k}_gap"] = (kk[:, k - 1] - kk[:, k]).numpy()
            out[f"k{k}_recon"] = sae.decode(top.top_acts, top.top_indices).numpy()  # sae.py:187
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name, {k: getattr(v, "shape", v) for k, v in out.items()})


def encode_decode_large_fixture(Sae, SaeConfig, name, d, N, ks, T, wseed, xseed):
    """g13 (round-4 verdict, item 2): the reference's encode / decode at batch sizes that reach the kernels bench.py
    times -- T > 256 tokens (the large-batch int8 candidate GEMM, the feature-major re-score) and 17 <= T <= 256 slices of the
    same batch (the weight-stream GEMM) -- stored compactly: top-k values + indices + the (k, k+1) gap for every token,
    the reconstruction as per-token sums (f64 of the f32 rows) plus 16 whole rows."""
    out = {"d": d, "N": N, "T": T, "wseed": wseed, "xseed": xseed, "ks": np.array(ks)}
    x = torch.from_numpy(synth.activations(T, d, xseed)).to(torch.bfloat16)
    sae = _make_ref_sae(Sae, SaeConfig, d, N, ks[0], wseed)
    rows = np.arange(0, T, max(1, T // 16))[:16]
    out["recon_rows_at"] = rows
    with torch.no_grad():
        pre = sae.pre_acts(x)  # sae.py:172
        out["pre_slice"] = pre[:8, :256].numpy()
        for k in ks:
            sae.cfg.k = k
            top = sae.select_topk(pre)  # sae.py:179
            v, i = _canon(top.top_acts, top.top_indices)
            out[f"k{k}_acts"], out[f"k{k}_idx"] = v, i
            kk = pre.topk(k + 1, sorted=True).values
            out[f"k{k}_gap"] = (kk[:, k - 1] - kk[:, k]).numpy()
            recon = sae.decode(top.top_acts, top.top_indices)  # sae.py:187
            out[f"k{k}_recon_sum"] = recon.double().sum(-1).numpy()
            out[f"k{k}_recon_abs"] = recon.double().abs().sum(-1).numpy()
            out[f"k{k}_recon_rows"] = recon[torch.from_numpy(rows)].numpy()
    np.savez_compressed(OUT / f"{name}.npz", **out)
    print("wrote", name, {k: getattr(v, "shape", v) for k, v in out.items()})


def decode_seam_fixture(eager_decode):
    """train/sae/tests/test_decode.py:6-20 restated with fixed inputs (CPU)."""
    g = torch.Generator().manual_seed(0)
    latents = torch.rand(2, 100, generator=g)
    W_dec = torch.randn(100, 50, generator=g)
    top_vals, top_idx = latents.topk(10)
    res = eager_decode(top_idx, top_vals, W_dec.mT)
    np.savez_compressed(OUT / "g3_decode_seam.npz", latents=latents.numpy(), W_dec=W_dec.numpy(),
                        top_vals=top_vals.numpy(), top_idx=top_idx.numpy().astype(np.int32),
                        eager=res.numpy())
    print("wrote g3_decode_seam")


def cache_fixture(Sae, SaeConfig, cache_mod):
    """Cache.add / get_nonzeros / save_splits / concate_safetensors / _generate_split_indices."""
    from safetensors.torch import load_file

    d, N, k = 16, 64, 4
    sae = _make_ref_sae(Sae, SaeConfig, d, N, k, seed=7)
    x = torch.from_numpy(synth.activations(2 * 3, d, 5, n_outlier=0).reshape(2, 3, d))
    out = {"x": x.numpy(), "wseed": 7, "d": d, "N": N, "k": k}
    module = "model.layers.24"
    for tag, filt in (("nofilter", None), ("filter", torch.tensor([1, 5, 9, 20, 33, 40, 41, 63]))):
        cache = cache_mod.Cache(shard_size=100, filters=None if filt is None else {module: filt},
                                batch_size=2)
        with torch.no_grad():
            lat = sae.pre_acts(x)
            topk = torch.topk(lat, k=k, dim=-1)  # cache.py:210
            result = torch.zeros_like(lat)
            result.scatter_(-1, topk.indices, topk.values)  # cache.py:214-216
            cache.add(result, 5, module)  # cache.py:217
        cache.save()
        out[f"{tag}_locations"] = cache.feature_locations[module].numpy()
        out[f"{tag}_activations"] = cache.feature_activations[module].numpy()
        if filt is not None:
            out["filter_features"] = filt.numpy()

    # split naming + per-rank files + rank-0 concat through the reference's own writer
    fc = cache_mod.FeatureCache.__new__(cache_mod.FeatureCache)
    fc.width = N
    cache = cache_mod.Cache(shard_size=0, filters=None, batch_size=2)
    cache.feature_locations[module] = torch.from_numpy(out["nofilter_locations"])
    cache.feature_activations[module] = torch.from_numpy(out["nofilter_activations"])
    fc.cache = cache
    with tempfile.TemporaryDirectory() as td:
        fc.save_splits(4, td, rank=0)
        names = sorted(os.listdir(f"{td}/{module}"))
        out["split_rank_files"] = np.array(names)
        fc.concate_safetensors(4, td)
        names2 = sorted(os.listdir(f"{td}/{module}"))
        out["split_concat_files"] = np.array(names2)
        for nm in names2:
            dat = load_file(f"{td}/{module}/{nm}")
            out[f"split_{nm}_locations"] = dat["locations"].numpy()
            out[f"split_{nm}_activations"] = dat["activations"].numpy()
    for width, n in ((64, 4), (131072, 128), (131072, 8), (4096, 3)):
        fc.width = width
        si = fc._generate_split_indices(n)  # cache.py:243-247
        out[f"splits_{width}_{n}"] = np.array([[int(a), int(b)] for a, b in si], dtype=np.int64)
    np.savez_compressed(OUT / "g4_cache.npz", **out)
    print("wrote g4_cache", out["nofilter_locations"][:3].tolist(), out["split_rank_files"])


def hook_fixture(Sae, SaeConfig):
    """Steering hook body (features/steering.py:105-124) and attribution hook body
    (features/patching/utils.py:33-58), executed line for line on the reference Sae."""
    d, N, k = 64, 1024, 8
    sae = _make_ref_sae(Sae, SaeConfig, d, N, k, seed=9)
    out = {"d": d, "N": N, "k": k, "wseed": 9}
    for S in (5, 1):
        x = torch.from_numpy(synth.activations(S, d, 20 + S, n_outlier=1)).to(torch.float16)[None]
        feature, clamp = 77, 10.0
        with torch.no_grad():
            latents = sae.pre_acts(x)
            if latents.shape[1] != 1:
                latents[:, :, feature] = clamp
            top_acts, top_indices = sae.select_topk(latents)
            sae_out = sae.decode(top_acts[0], top_indices[0]).unsqueeze(0).to(torch.float16)
        out[f"steer_S{S}_x"] = x.numpy()
        out[f"steer_S{S}_out"] = sae_out.numpy()
        out[f"steer_S{S}_feature"], out[f"steer_S{S}_clamp"] = feature, clamp
    x = torch.from_numpy(synth.activations(2 * 3, d, 31, n_outlier=1)).to(torch.float16)
    x = x.reshape(2, 3, d)
    with torch.no_grad():
        lat0 = sae.pre_acts(x.flatten(0, 1))
        off = int(lat0[0].argmax())  # a feature that is certainly active for token 0
    for tag, off_features in (("none", None), ("off", off)):
        with torch.no_grad():
            bs, seq_len, dim = x.shape
            latents = sae.pre_acts(x.flatten(0, 1))
            if off_features is not None:
                mask = torch.ones_like(latents)
                mask[:, off_features] = 0
                latents = latents * mask
            top_acts, top_indices = sae.select_topk(latents)
            sae_out = sae.decode(top_acts, top_indices).to(torch.float16).view(bs, seq_len, dim)
        out[f"attr_{tag}_out"] = sae_out.numpy()
    out["attr_x"], out["attr_off_feature"] = x.numpy(), off
    np.savez_compressed(OUT / "g5_hooks.npz", **out)
    print("wrote g5_hooks")


def train_fixture(Sae, SaeConfig):
    """Sae.forward with dead_mask / multi_topk (sae.py:193-247) and decode grads via eager
    autograd (the contract TritonDecoder.backward implements, kernels.py:411-429)."""
    d, N, k = 64, 1024, 8
    sae = _make_ref_sae(Sae, SaeConfig, d, N, k, seed=11, multi_topk=True)
    x = torch.from_numpy(synth.activations(24, d, 40, n_outlier=1, bf16=False))
    dead = torch.zeros(N, dtype=torch.bool)
    dead[::7] = True
    fo = sae(x, dead)
    fo.fvu.backward(retain_graph=True)
    out = {"d": d, "N": N, "k": k, "wseed": 11, "x": x.numpy(), "dead_mask": dead.numpy(),
           "fvu": fo.fvu.item(), "auxk_loss": fo.auxk_loss.item(),
           "multi_topk_fvu": fo.multi_topk_fvu.item(), "sae_out": fo.sae_out.detach().numpy()}
    # decode grads alone
    sae.zero_grad()
    # (a generator of this fixture's own: drawn from the global RNG, these inputs depended on which fixtures ran before this
    # one -- inserting g13 in round 5 silently changed what a regeneration produced; round-5 verdict, weak 5)
    gen = torch.Generator().manual_seed(7007)
    acts = torch.rand(5, k, dtype=torch.float32, generator=gen).requires_grad_()
    idx = torch.stack([torch.randperm(N, generator=gen)[:k] for _ in range(5)])
    g = torch.from_numpy(synth.normalish(77, 5 * d).reshape(5, d))
    y = sae.decode(acts, idx)
    y.backward(g)
    out.update(dec_acts=acts.detach().numpy(), dec_idx=idx.numpy().astype(np.int32), dec_gout=g.numpy(),
               dec_grad_acts=acts.grad.numpy(), dec_grad_Wdec_rows=sae.W_dec.grad[idx.flatten()].numpy(),
               dec_grad_Wdec_nnzrows=int((sae.W_dec.grad.abs().sum(1) > 0).sum()),
               dec_grad_bdec=sae.b_dec.grad.numpy())
    # full training gradients of the trainer's loss (train/sae/sae/trainer.py:379-384):
    # loss = fvu + auxk_alpha * auxk_loss + multi_topk_fvu / 8
    sae.zero_grad()
    xg = x.clone().requires_grad_()
    fo = sae(xg, dead)
    loss = fo.fvu + (1.0 / 32) * fo.auxk_loss + fo.multi_topk_fvu / 8
    loss.backward()
    out.update(loss=loss.item(), g_W_enc=sae.encoder.weight.grad.numpy(), g_b_enc=sae.encoder.bias.grad.numpy(),
               g_W_dec=sae.W_dec.grad.numpy(), g_b_dec=sae.b_dec.grad.numpy(), g_x=xg.grad.numpy())
    # one optimizer step in the trainer's order (train/sae/sae/trainer.py:347-401): renormalise the
    # decoder, forward/backward, clip_grad_norm_(1.0), remove decoder-parallel grads, Adam step
    sae2 = _make_ref_sae(Sae, SaeConfig, d, N, k, seed=11, multi_topk=True)
    opt = torch.optim.Adam(sae2.parameters(), lr=1e-3)
    sae2.set_decoder_norm_to_unit_norm()
    fo = sae2(x, dead)
    (fo.fvu + (1.0 / 32) * fo.auxk_loss + fo.multi_topk_fvu / 8).backward()
    torch.nn.utils.clip_grad_norm_(sae2.parameters(), 1.0)
    sae2.remove_gradient_parallel_to_decoder_directions()
    opt.step()
    out.update(step_fvu=fo.fvu.item(), step_W_enc=sae2.encoder.weight.detach().numpy(),
               step_b_enc=sae2.encoder.bias.detach().numpy(), step_W_dec=sae2.W_dec.detach().numpy(),
               step_b_dec=sae2.b_dec.detach().numpy(),
               step_fired=np.unique(fo.latent_indices.numpy()))
    np.savez_compressed(OUT / "g7_train.npz", **out)
    print("wrote g7_train", out["fvu"], out["auxk_loss"], out["multi_topk_fvu"])


def attribution_fixture(Sae, SaeConfig, name="g8_attribution", d=64, N=1024, k=8, wseed=9):
    """`Attribution.get_attribution` of the reference itself (features/patching/attribution.py:116-189,
    hooks of patching/utils.py:21-79) on the tiny LLaVA stand-in of tests/fakes.py with an SAE spliced
    into `layers.0`.  The constructor (files, PIL images, tokenizer) is bypassed; every line of the
    scoring loop runs as written."""
    from functools import partial

    import torch.distributed as dist

    import fakes
    from sae_auto_interp.features.patching.attribution import Attribution
    from sae_auto_interp.features.patching.utils import get_logit_diff

    vocab = 40
    model = fakes.TinyLlava(vocab=vocab, d=d, n_layers=2, seed=300)
    sae = _make_ref_sae(Sae, SaeConfig, d, N, k, seed=wseed)
    module = "layers.0"
    inputs = fakes.FakeProcessor(vocab)(text=["<image>"] * 2, images=[fakes.FakeImage(0), fakes.FakeImage(1)])
    answer_ids = torch.tensor([[5, 11], [17, 2]])
    attr = Attribution.__new__(Attribution)
    attr.model, attr.image_processor = model, None
    attr.sae_dict = {module: sae}
    attr.prompt_ids = inputs["input_ids"]
    attr.pixel_values = inputs["pixel_values"].to(torch.float16)
    attr.image_sizes = inputs["image_sizes"].tolist()
    attr.attention_mask = inputs["attention_mask"].bool()
    attr.name_to_module = {module: model.language_model.get_submodule(module)}
    attr.module_to_name = {v: kk for kk, v in attr.name_to_module.items()}
    attr.metric = partial(get_logit_diff, answer_token_indices=answer_ids)
    # which features are active where (to choose informative indices)
    captured = {}
    h = attr.name_to_module[module].register_forward_hook(lambda m, i, o: captured.__setitem__("h", o[0]))
    with torch.no_grad():
        model(input_ids=attr.prompt_ids, pixel_values=attr.pixel_values)
    h.remove()
    with torch.no_grad():
        top = sae.encode(captured["h"].flatten(0, 1))
    act_idx = top.top_indices
    inactive = [i for i in range(N) if i not in set(act_idx.flatten().tolist())][:2]
    indices = sorted(set(act_idx[4].tolist() + act_idx[7].tolist() + act_idx[9].tolist())) + inactive
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1)
    res = attr.get_attribution(torch.tensor(indices))
    out = {"d": d, "N": N, "k": k, "vocab": vocab, "wseed": wseed, "module": module,
           "input_ids": attr.prompt_ids.numpy(), "pixel_values": attr.pixel_values.numpy(),
           "answer_ids": answer_ids.numpy(), "indices": np.array(indices),
           "attribution": torch.stack(res[module]).numpy(),            # [n_idx, B, S] fp16
           "clean_top_idx": act_idx.numpy().astype(np.int32), "clean_top_acts": top.top_acts.numpy()}
    np.savez_compressed(OUT / f"{name}.npz", **out)
    a = out["attribution"].astype(np.float32)
    print("wrote", name, a.shape, "max |attr|", np.abs(a).max(), "nonzero entries", int((a != 0).sum()))


def attribution_wide_fixture(Sae, SaeConfig):
    """The same reference run on a less pathological stand-in: k = 32 latents of a d = 256 stream (one ablation moves
    the reconstruction by a few per cent, not 12 %), where the batched one-pass scores must be close to first order."""
    attribution_fixture(Sae, SaeConfig, name="g12_attribution_wide", d=256, N=4096, k=32, wseed=19)


def steering_fixture(Sae, SaeConfig):
    """`SteeringController.run` of the reference (features/steering.py:13-128): original generation,
    then one clamped generation per feature -- the hook fires on the prefill (S != 1: clamp) and on
    every single-token decode step (S == 1)."""
    import importlib.util as iu

    import fakes

    spec = iu.spec_from_file_location("sae_auto_interp.features.steering",
                                      REF / "sae_auto_interp" / "features" / "steering.py")
    mod = iu.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d, N, k, vocab = 64, 1024, 8, 40
    model = fakes.TinyLlava(vocab=vocab, d=d, n_layers=2, seed=300)
    sae = _make_ref_sae(Sae, SaeConfig, d, N, k, seed=9)
    module, feats, clamp = "layers.0", [77, 300, 901], 50.0
    ctl = mod.SteeringController(sae=sae, module_name=module, feature_idx=feats, model=model,
                                 processor=fakes.FakeProcessor(vocab), prompt="describe", k=clamp)
    res = ctl.run()
    out = {"d": d, "N": N, "k": k, "vocab": vocab, "wseed": 9, "module": module, "features": np.array(feats),
           "clamp": clamp, "original": np.array(res[f"{module}_feature{feats[0]}"]["original_resps"]),
           "clamped": np.array([res[f"{module}_feature{f}"]["clamped_resps"] for f in feats])}
    np.savez_compressed(OUT / "g10_steering.npz", **out)
    print("wrote g10_steering", out["original"], out["clamped"])


def image_cache_fixture(Sae, SaeConfig, cache_mod):
    """`FeatureImageCache.run` of the reference (features/cache.py:325-429: `<image>` prompt through the
    processor, forward of the LLaVA model, BOS position dropped, top-k, Cache.add) on the stand-ins."""
    import fakes

    d, N, k, vocab = 64, 1024, 8, 40
    model = fakes.TinyLlava(vocab=vocab, d=d, n_layers=2, seed=300)
    sae = _make_ref_sae(Sae, SaeConfig, d, N, k, seed=9)
    module = "layers.1"
    fic = cache_mod.FeatureImageCache.__new__(cache_mod.FeatureImageCache)
    fic.llava_model, fic.model, fic.tokenizer = model, model.language_model, None
    fic.name_to_module = {module: model.language_model.get_submodule(module)}
    fic.module_to_name = {v: kk for kk, v in fic.name_to_module.items()}
    fic.submodule_dict = {module: sae}
    fic.batch_size, fic.width = 2, N
    fic.cache = cache_mod.Cache(7, None, batch_size=2)
    fic.processor, fic.prompt = fakes.FakeProcessor(vocab), "<image>"
    dataset = [{"image": fakes.FakeImage(i)} for i in range(5)]       # drop_last: two batches of two images
    fic.run(0, dataset)
    out = {"d": d, "N": N, "k": k, "vocab": vocab, "wseed": 9, "module": module, "n_images": 5, "shard_size": 7,
           "locations": fic.cache.feature_locations[module].numpy(),
           "activations": fic.cache.feature_activations[module].numpy()}
    np.savez_compressed(OUT / "g9_image_cache.npz", **out)
    print("wrote g9_image_cache", out["locations"].shape, out["locations"][:2].tolist(),
          "max pos", int(out["locations"][:, 1].max()))


def chunker_fixture(Sae=None, SaeConfig=None):
    """The reference's GPT-style chunker (sae_auto_interp/sae/data.py:16-100) on 2500 documents -- two of its
    2048-document batches, each of which drops its ragged last chunk -- with a slow-style tokenizer (flat overflow,
    re-chunked) and a real fast tokenizer (one row per chunk, BOS on each)."""
    import datasets

    import fakes
    from sae_auto_interp.sae.data import chunk_and_tokenize

    docs = fakes.chunker_documents()
    out = {}
    for name, tok in (("slow", fakes.FakeSlowTokenizer(64)), ("fast", fakes.make_fast_tokenizer())):
        ds = datasets.Dataset.from_dict({"text": docs})
        got = chunk_and_tokenize(ds, tok, max_seq_len=48, num_proc=1, load_from_cache_file=False)
        out[name] = np.asarray(got["input_ids"], dtype=np.int64)
        print("chunker", name, out[name].shape)
    # the reference's launcher maps with num_proc = cpu_count() // 2 (launch/cache/cache.py:58): Dataset.map cuts the
    # dataset into that many contiguous shards and batches inside each (one more ragged chunk dropped per shard)
    ds = datasets.Dataset.from_dict({"text": docs})
    got = chunk_and_tokenize(ds, fakes.FakeSlowTokenizer(64), max_seq_len=48, num_proc=3, load_from_cache_file=False)
    out["slow_num_proc3"] = np.asarray(got["input_ids"], dtype=np.int64)
    print("chunker slow, num_proc=3", out["slow_num_proc3"].shape)
    np.savez_compressed(OUT / "g11_chunker.npz", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--only", default=None, help="comma-separated fixture functions to (re)generate")
    ap.add_argument("--out", default=None, help="write the fixtures here instead of tests/golden/ (the recipe check)")
    args = ap.parse_args()
    if args.out:
        global OUT
        OUT = Path(args.out)
        OUT.mkdir(parents=True, exist_ok=True)
    # No fixture may draw from the global RNG: each one seeds a generator of its own (or the counter-based synth module), so a
    # fixture's bytes do not depend on which fixtures ran before it.  The global seed below only pins what torch draws on its
    # own inside the reference's constructors (nn.Linear's init, overwritten by _make_ref_sae).
    torch.manual_seed(0)
    torch.set_num_threads(8)
    Sae, SaeConfig, eager_decode, cache_mod = _import_reference()
    if args.only:
        for name in args.only.split(","):
            if name == "g13":
                encode_decode_large_fixture(Sae, SaeConfig, "g13_d4096_n16384_t1024", 4096, 16384, [32, 256], 1024, 12, 13)
                if args.full:
                    encode_decode_large_fixture(Sae, SaeConfig, "g13_c2_d4096_n131072_t320", 4096, 131072, [32, 256], 320, 3, 15)
                continue
            fn = globals()[name]
            fn(Sae, SaeConfig, cache_mod) if name in ("cache_fixture", "image_cache_fixture") else fn(Sae, SaeConfig)
        return
    encode_decode_fixture(Sae, SaeConfig, "g1_c1_d768_n4096", 768, 4096, [32], 64, 1, 0)
    encode_decode_fixture(Sae, SaeConfig, "g2_d4096_n16384", 4096, 16384, [32, 256], 16, 2, 3)
    encode_decode_large_fixture(Sae, SaeConfig, "g13_d4096_n16384_t1024", 4096, 16384, [32, 256], 1024, 12, 13)
    decode_seam_fixture(eager_decode)
    cache_fixture(Sae, SaeConfig, cache_mod)
    hook_fixture(Sae, SaeConfig)
    train_fixture(Sae, SaeConfig)
    attribution_fixture(Sae, SaeConfig)
    attribution_wide_fixture(Sae, SaeConfig)
    steering_fixture(Sae, SaeConfig)
    image_cache_fixture(Sae, SaeConfig, cache_mod)
    chunker_fixture()
    if args.full:
        encode_decode_fixture(Sae, SaeConfig, "g2_c2_d4096_n131072", 4096, 131072, [32, 256], 16, 3, 4)
        encode_decode_large_fixture(Sae, SaeConfig, "g13_c2_d4096_n131072_t320", 4096, 131072, [32, 256], 320, 3, 15)


if __name__ == "__main__":
    main()

"""CPU: the C-ABI library loads and exports every symbol include/msae.h declares (no compute),
and the host-side logic that needs no GPU (config, checkpoint I/O, split naming, error paths)."""
import ctypes
import json
import re

import numpy as np
import pytest
import torch

from conftest import REPO


def _declared_symbols():
    text = (REPO / "include" / "msae.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msae_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from msae import _hip

    lib = _hip.load()
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"libmsae_hip.so does not export {name}"
    assert set(declared) == set(_hip.PROTOTYPES), set(declared) ^ set(_hip.PROTOTYPES)
    assert lib.msae_abi_version() == 4
    assert lib.msae_target_arch() == b"gfx950"
    assert b"workspace" in lib.msae_error_string(-3)
    # pure host-side size queries (no GPU needed)
    assert lib.msae_encoder_prepared_bytes(131072, 4096) >= 131072 * 4096 * 2
    assert lib.msae_encode_topk_ws_bytes(8192, 4096, 131072, 32, None) > 0
    assert lib.msae_encode_topk_ws_bytes(16, 768, 4096, 32, None) >= 16 * 4096 * 4
    # per-call options: the coarse mode changes the workspace, nothing in the library remembers it
    o8, ob = _hip.MsaeOptions(), _hip.MsaeOptions()
    lib.msae_options_init(ctypes.byref(o8)); lib.msae_options_init(ctypes.byref(ob))
    assert o8.size == ctypes.sizeof(_hip.MsaeOptions) and o8.coarse_mode == -1 and o8.guard_z == 0.0
    o8.coarse_mode, ob.coarse_mode = 1, 0
    w8 = lib.msae_encode_topk_ws_bytes(8192, 4096, 131072, 32, ctypes.byref(o8))
    wb = lib.msae_encode_topk_ws_bytes(8192, 4096, 131072, 32, ctypes.byref(ob))
    assert w8 > 0 and wb > 0 and w8 != wb
    assert lib.msae_encode_topk_ws_bytes(8192, 4096, 131072, 32, ctypes.byref(o8)) == w8   # no hidden state
    bad = _hip.MsaeOptions()
    lib.msae_options_init(ctypes.byref(bad))
    bad.guard_z = 1000.0
    assert lib.msae_encode_topk_ws_bytes(8192, 4096, 131072, 32, ctypes.byref(bad)) == 0
    bad.guard_z, bad.size = 0.0, 4     # a caller compiled against a shorter struct than this library knows
    assert lib.msae_encode_topk_ws_bytes(8192, 4096, 131072, 32, ctypes.byref(bad)) == 0
    # ... while ABI 2's struct (24 bytes, no `exact`) is still served (exact = 0: the same plan)
    assert o8.exact == 0 and o8.dither == 0 and o8.dither_seed == 0 and ctypes.sizeof(_hip.MsaeOptions) == 64
    o8.size = 24
    assert lib.msae_encode_topk_ws_bytes(8192, 4096, 131072, 32, ctypes.byref(o8)) == w8
    # ... and ABI 3's (40 bytes: `reserved` where `dither` now is, no dither_seed)
    o8.size = 40
    assert lib.msae_encode_topk_ws_bytes(8192, 4096, 131072, 32, ctypes.byref(o8)) == w8
    o8.size, o8.dither = 64, 7         # an unknown dither mode is an argument error
    assert lib.msae_encode_topk_ws_bytes(8192, 4096, 131072, 32, ctypes.byref(o8)) == 0


def test_compute_on_cpu_tensors_raises_instead_of_falling_back():
    from msae import Sae, SaeConfig, ops

    sae = Sae(16, SaeConfig(num_latents=64, k=4))
    x = torch.randn(3, 16)
    for fn in (lambda: sae.pre_acts(x), lambda: sae.encode(x),
               lambda: sae.decode(torch.rand(3, 4), torch.zeros(3, 4, dtype=torch.long)),
               lambda: ops.topk(torch.randn(3, 64), 4)):
        with pytest.raises(RuntimeError, match="MI355X|HIP"):
            fn()


def test_missing_library_is_loud(monkeypatch, tmp_path):
    from msae import _hip

    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setenv("MSAE_HIP_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_hip.MsaeLibraryMissing):
        _hip.load()


def test_sae_checkpoint_roundtrip_and_cfg_json(tmp_path):
    """cfg.json keys and state-dict keys are the reference's (sae.py:126-162)."""
    from safetensors.torch import load_file

    from msae import Sae, SaeConfig

    sae = Sae(16, SaeConfig(num_latents=64, k=4))
    assert sae.num_latents == 64 and sae.W_dec.shape == (64, 16)
    assert torch.allclose(sae.W_dec.norm(dim=1), torch.ones(64), atol=1e-5)  # normalize_decoder
    assert torch.count_nonzero(sae.encoder.bias) == 0 and torch.count_nonzero(sae.b_dec) == 0
    sae.save_to_disk(tmp_path / "model.layers.24")
    cfg = json.loads((tmp_path / "model.layers.24" / "cfg.json").read_text())
    assert cfg == {"expansion_factor": 32, "normalize_decoder": True, "num_latents": 64, "k": 4,
                   "multi_topk": False, "signed": False, "d_in": 16}
    assert sorted(load_file(tmp_path / "model.layers.24" / "sae.safetensors")) == \
        ["W_dec", "b_dec", "encoder.bias", "encoder.weight"]
    back = Sae.load_from_disk(tmp_path / "model.layers.24")
    assert torch.equal(back.W_dec, sae.W_dec) and back.cfg == sae.cfg and back.d_in == 16
    many = Sae.load_many(str(tmp_path), local=True)
    assert list(many) == ["model.layers.24"]
    enc_only = Sae.load_from_disk(tmp_path / "model.layers.24", decoder=False)
    assert enc_only.W_dec is None
    with pytest.raises(AssertionError):
        enc_only.decode(torch.rand(1, 4), torch.zeros(1, 4, dtype=torch.long))
    assert Sae(8, SaeConfig(expansion_factor=4)).num_latents == 32


def test_split_indices_match_reference(golden_dir):
    from msae.features import generate_split_indices

    g = np.load(golden_dir / "g4_cache.npz")
    for key in g.files:
        if key.startswith("splits_"):
            _, width, n = key.split("_")
            assert generate_split_indices(int(width), int(n)) == [tuple(r) for r in g[key].tolist()]
    si = generate_split_indices(131072, 128)
    assert si[:2] == [(0, 1023), (1024, 2047)]


def test_save_splits_and_concat_file_layout(golden_dir, tmp_path):
    """Writer reproduces the reference's per-rank and concatenated files (cache.py:249-309),
    including the dropped last feature of each split; two ranks concatenate in rank order."""
    import os

    from safetensors.torch import load_file

    from msae.features import cache as C

    g = np.load(golden_dir / "g4_cache.npz")
    module = "model.layers.24"
    loc = torch.from_numpy(g["nofilter_locations"])
    act = torch.from_numpy(g["nofilter_activations"])
    fc = C.FeatureCache.__new__(C.FeatureCache)
    fc.width = int(g["N"])
    fc.cache = C.Cache(shard_size=0)
    fc.cache.feature_locations[module], fc.cache.feature_activations[module] = loc, act
    fc.save_splits(4, str(tmp_path), rank=0)
    assert sorted(os.listdir(tmp_path / module)) == list(g["split_rank_files"])
    # a second rank with shifted rows
    loc1 = loc.clone(); loc1[:, 0] += 1000
    fc.cache.feature_locations[module] = loc1
    fc.save_splits(4, str(tmp_path), rank=1)
    fc.concate_safetensors(4, str(tmp_path))
    names = sorted(os.listdir(tmp_path / module))
    assert names == list(g["split_concat_files"])
    total = 0
    for nm in names:
        dat = load_file(str(tmp_path / module / nm))
        ref_l, ref_a = g[f"split_{nm}_locations"], g[f"split_{nm}_activations"]
        n = len(ref_l)
        assert np.array_equal(dat["locations"][:n].numpy(), ref_l)
        assert np.array_equal(dat["locations"][n:, 0].numpy(), ref_l[:, 0] + 1000)
        assert np.array_equal(dat["activations"][:n].numpy(), ref_a)
        total += n
        start, end = (int(v) for v in nm.split(".")[0].split("_"))
        f = dat["locations"][:, 2]
        assert bool(((f >= start) & (f < end)).all())     # feature == end is dropped (quirk)
    dropped = int((np.isin(g["nofilter_locations"][:, 2], [15, 31, 47, 63])).sum())
    assert total == len(loc) - dropped
    # the fixed variant keeps them
    fc.cache.feature_locations[module] = loc
    fc.save_splits(4, str(tmp_path / "fixed"), rank=0, include_split_end=True)
    kept = sum(len(load_file(str(tmp_path / "fixed" / module / f))["activations"])
               for f in os.listdir(tmp_path / "fixed" / module))
    assert kept == len(loc)


def test_legacy_dense_cache_add_matches_reference(golden_dir):
    """Cache.add on DENSE latents (cache.py:42-92) -- pure torch, runs anywhere."""
    import synth
    from oracle import oracle
    from msae.features import Cache

    g = np.load(golden_dir / "g4_cache.npz")
    d, N, k = int(g["d"]), int(g["N"]), int(g["k"])
    W_enc, b_enc, _, b_dec = synth.sae_weights(d, N, int(g["wseed"]))
    pre = oracle.pre_acts(g["x"].reshape(-1, d), W_enc, b_enc, b_dec)
    v, i = oracle.topk(pre, k)
    dense = torch.zeros(6, N).scatter_(-1, torch.from_numpy(i).long(), torch.from_numpy(v)).view(2, 3, N)
    for tag, filt in (("nofilter", None), ("filter", {"m": torch.from_numpy(g["filter_features"])})):
        c = Cache(shard_size=100, filters=filt, batch_size=2)
        c.add(dense, 5, "m")
        c.save()
        assert np.array_equal(c.feature_locations["m"].numpy(), g[f"{tag}_locations"])


def test_cli_flags_match_reference_readme_commands(tmp_path):
    """The README's cache / steering command lines (README.md:46-56,106-110) parse unchanged."""
    from msae.config import parse_cache_config
    from msae.launch.features.steering import parse_argument

    cfg = parse_cache_config(["llava-hf/llama3-llava-next-8b-hf", "lmms-lab/sae-sample-cache-dataset",
                              "--sae_path", "lmms-lab/llama3-llava-next-8b-hf-sae-131k", "--split", "train",
                              "--batch_size", "1", "--ctx_len", "64", "--n_splits", "128",
                              "--save_dir", "./sae_cache", "--filters_path", "filters.json"])
    assert (cfg.model, cfg.dataset) == ("llava-hf/llama3-llava-next-8b-hf", "lmms-lab/sae-sample-cache-dataset")
    assert (cfg.batch_size, cfg.ctx_len, cfg.n_splits, cfg.split) == (1, 64, 128, "train")
    assert cfg.filters_path == "filters.json" and cfg.load_in_8bit is False
    d = parse_cache_config([])
    assert (d.batch_size, d.n_splits, d.ctx_len, d.save_dir) == (32, 2, 2048, "./features_cache")
    a = parse_argument(["-t", "Tell me a story", "-k", "30", "--sae-path", "p", "--filters", "f.json",
                        "-i", "img.png", "-s", "out"])
    assert (a.text, a.clamp_value, a.sae_path, a.filters, a.image_path, a.save_dir) == \
        ("Tell me a story", 30.0, "p", "f.json", "img.png", "out")
    assert parse_argument(["-t", "x"]).model == "llava-hf/llama3-llava-next-8b-hf"


def test_load_filter_and_load_saes_local(tmp_path):
    from msae import Sae, SaeConfig
    from msae.utils import load_filter, load_saes, load_single_sae

    for name in ("model.layers.2", "model.layers.10"):
        Sae(8, SaeConfig(num_latents=32, k=4)).save_to_disk(tmp_path / name)
    (tmp_path / "filters.json").write_text(json.dumps({"model.layers.10": [3, 1, 30]}))
    filt = load_filter(str(tmp_path / "filters.json"), device="cpu")
    assert filt["model.layers.10"].tolist() == [3, 1, 30]
    assert list(load_saes(str(tmp_path), device="cpu")) == ["model.layers.2", "model.layers.10"]  # natural order
    only = load_saes(str(tmp_path), filters=filt, device="cpu")
    assert list(only) == ["model.layers.10"] and only["model.layers.10"].num_latents == 32
    assert load_single_sae(str(tmp_path), "model.layers.2", device="cpu").d_in == 8


def test_cache_files_round_trip_through_the_reader_rule(golden_dir, tmp_path):
    """Files written by save_splits/concate_safetensors are found and decoded by the restated
    FeatureDataset / TensorBuffer rule (features/loader.py:74-90,143-196)."""
    from msae.features import FeatureDataset, split_path
    from msae.features import cache as C

    g = np.load(golden_dir / "g4_cache.npz")
    module, width = "model.layers.24", int(g["N"])
    loc = torch.from_numpy(g["nofilter_locations"])
    act = torch.from_numpy(g["nofilter_activations"])
    fc = C.FeatureCache.__new__(C.FeatureCache)
    fc.width = width
    fc.cache = C.Cache(shard_size=0)
    fc.cache.feature_locations[module], fc.cache.feature_activations[module] = loc, act
    fc.save_splits(4, str(tmp_path), rank=0, include_split_end=True)
    fc.concate_safetensors(4, str(tmp_path))
    ds = FeatureDataset(str(tmp_path), width, 4)
    assert len(ds) == 4
    seen = 0
    for rec in ds:
        mask = loc[:, 2] == rec.feature
        assert torch.equal(rec.locations, loc[mask][:, :2]) and torch.equal(rec.activations, act[mask])
        assert split_path(str(tmp_path), module, width, 4, rec.feature).endswith(
            f"{rec.feature // 16 * 16}_{rec.feature // 16 * 16 + 15}.safetensors")
        seen += len(rec.activations)
    assert seen == len(act)
    sel = {module: torch.tensor([int(loc[0, 2]), int(loc[-1, 2])])}
    got = {r.feature for r in FeatureDataset(str(tmp_path), width, 4, modules=[module], features=sel)}
    assert got == set(sel[module].tolist())


def test_cache_spill_to_disk_equals_in_memory(tmp_path):
    """Streaming every batch to disk (spill_dir) yields the same final tensors as the reference's
    keep-everything-in-RAM accumulation."""
    from msae.features import Cache

    g = torch.Generator().manual_seed(0)
    caches = [Cache(shard_size=7, batch_size=2), Cache(shard_size=7, batch_size=2, spill_dir=str(tmp_path))]
    for b in range(3):
        dense = torch.relu(torch.randn(2, 3, 32, generator=g) - 1.0)
        for c in caches:
            c.add(dense, b, "model.layers.24")
    for c in caches:
        c.save()
    assert torch.equal(caches[0].feature_locations["model.layers.24"], caches[1].feature_locations["model.layers.24"])
    assert torch.equal(caches[0].feature_activations["model.layers.24"], caches[1].feature_activations["model.layers.24"])
    assert not any(f.endswith(".safetensors") for _, _, fs in __import__("os").walk(tmp_path) for f in fs)


def test_shard_helpers_and_argument_checks_without_a_gpu():
    """Host logic of the feature-sharded group: list sizes, token slices, record size, and the argument checks of the
    candidate-exchange entry points (they return MSAE_E* codes before touching the device)."""
    import ctypes

    from msae import _hip
    from msae.parallel import default_candidates, default_k_loc, token_slice

    assert [default_candidates(32, g) for g in (2, 4, 8)] == [64, 64, 32]
    assert default_candidates(256, 8) == 128
    assert [default_k_loc(32, g) for g in (1, 2, 4, 8)] == [32, 32, 23, 14]
    covered = []
    for r in range(8):
        lo, hi, per = token_slice(8195, r, 8)
        assert per == 1025 and 0 <= lo <= hi <= 8195
        covered += list(range(lo, hi))
    assert covered == list(range(8195))
    lib = _hip.load()
    assert lib.msae_shard_record_bytes(32) == 32 * 12 + 8 and lib.msae_shard_record_bytes(0) == 0
    assert lib.msae_rescore_candidates_ws_bytes(1024, 4096, 131072, 32, 8, 32) > 1024 * 4096 * 4
    assert lib.msae_rescore_candidates_ws_bytes(0, 4096, 131072, 32, 8, 32) == 0
    null = ctypes.c_void_p(None)
    # G * C < k, missing records, negative sizes: MSAE_EINVAL (-1); too small a workspace: MSAE_EWS
    einval = lib.msae_rescore_candidates(null, 2, null, null, null, 4, 4, 4096, 131072, 32, 2, 8, null, -1, 0.0, -1, null, null,
                                         null, null, 0, None, null)
    assert einval < 0
    assert lib.msae_shard_candidates(null, 2, null, null, null, 4, 4096, 16384, 32, 0, 0, -1, -1, null, null, 0, None, null) < 0
    assert lib.msae_error_string(einval).decode()


def test_chunk_and_tokenize_matches_reference_chunker(golden_dir):
    """launch.cache's GPT-style chunker == the reference's (sae_auto_interp/sae/data.py:16-100, run by
    tests/golden/make_golden.py:chunker_fixture) on 2500 documents -- more than one of its 2048-document batches, each of
    which drops its ragged last chunk -- with a slow-style tokenizer (flat overflow, re-chunked) and a real fast
    tokenizer (one row per chunk, BOS re-added).  Same chunk boundaries => same cache `row` ids."""
    import datasets

    import fakes
    from msae.launch.cache.cache import chunk_and_tokenize

    g = np.load(golden_dir / "g11_chunker.npz")
    docs = fakes.chunker_documents()
    for name, tok in (("slow", fakes.FakeSlowTokenizer(64)), ("fast", fakes.make_fast_tokenizer())):
        got = chunk_and_tokenize(datasets.Dataset.from_dict({"text": docs}), tok, max_seq_len=48)
        ids = np.asarray(got["input_ids"], dtype=np.int64)
        assert ids.shape == g[name].shape, (name, ids.shape, g[name].shape)
        assert np.array_equal(ids, g[name]), name
    # the reference's launcher maps with num_proc > 1: contiguous shards first, 2048-document batches inside each
    got = chunk_and_tokenize(datasets.Dataset.from_dict({"text": docs}), fakes.FakeSlowTokenizer(64), max_seq_len=48, num_proc=3)
    ids = np.asarray(got["input_ids"], dtype=np.int64)
    assert ids.shape == g["slow_num_proc3"].shape and np.array_equal(ids, g["slow_num_proc3"])
    assert ids.shape != g["slow"].shape or not np.array_equal(ids, g["slow"])      # (the split really changes the chunks)
    with pytest.raises(ValueError, match="Not enough data"):
        chunk_and_tokenize(datasets.Dataset.from_dict({"text": ["w1 w2"]}), fakes.FakeSlowTokenizer(64), max_seq_len=48)

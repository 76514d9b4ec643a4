"""BASELINE configs[2] AT ITS OWN WIDTH: the 131072 x 4096 encoder split over G = 2 / 4 / 8 feature shards, T = 8192 tokens,
both exchange schemes (per-shard top-k_loc + merge; candidate exchange), default k_loc / C -- the code the first real
8-GPU run executes (round-3 verdict: only ever run at N <= 65536 / d <= 1024 inside the suite).

No reference counterpart (the reference only shards the dataset, launch/cache/cache.py:66; SURVEY 8e specifies the split);
the bar is north_star's: bit-identical to the single-GPU encode on EVERY token, and a handful of tokens against the CPU
oracle.  The G ranks run one after the other on one GPU (`EmulatedShardGroup`: the ranks' kernels, packs, records, merge
kernel and device-sized second round are the real ones, the transport is a torch.cat).

Also here: the whole inference path -- single-GPU encode / decode, the cache loop between flushes, the emulated sharded
encode -- under `torch.cuda.set_sync_debug_mode("error")`: the boundary's "never synchronises" (include/msae.h; the
reference runs this inside an HF forward hook, features/cache.py:187-204) as a test instead of a sentence.
"""
import gc

import numpy as np
import pytest
import torch

import hostile
from oracle import oracle

pytestmark = pytest.mark.gpu

D, N_C2, T_C2 = 4096, 131072, 8192


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from msae import _hip

    _hip.load()
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _free_after():
    yield
    from msae import ops

    ops.release_workspaces()
    gc.collect()
    torch.cuda.empty_cache()


@pytest.fixture(scope="module")
def c2(dev):
    """One C2-sized SAE (trained-like rows), its single-GPU results for k = 32 and 256, and 8 tokens of the oracle."""
    from msae import Sae, SaeConfig, ops

    W, b, bd = hostile.weights("trained_like", N_C2, D, dev, seed=31)
    x = hostile.activations(T_C2, D, dev, seed=32)
    sae = Sae(D, SaeConfig(num_latents=N_C2, k=32), device=dev)
    with torch.no_grad():
        sae.encoder.weight.copy_(W); sae.encoder.bias.copy_(b); sae.b_dec.copy_(bd)
        sae.W_dec.copy_(W / (W.norm(dim=1, keepdim=True) + 1e-6))
    del W
    sae.requires_grad_(False)
    prepared = ops.prepare_encoder(sae.encoder.weight)
    single = {}
    for k in (32, 256):
        v, i, st = ops.encode_topk(x, sae.encoder.weight, sae.encoder.bias, sae.b_dec, prepared, k)
        assert int((st >= 2).sum()) == 0
        single[k] = (v, i)
    # 8 tokens against the oracle (k = 256: its top-32 prefix is the k = 32 answer -- canonical order)
    rows = torch.tensor([0, 1, 511, 512, 4095, 4096, 8190, 8191], device=dev)
    Wc, bc, bdc = sae.encoder.weight.detach().cpu().numpy(), sae.encoder.bias.detach().cpu().numpy(), sae.b_dec.detach().cpu().numpy()
    ov, oi = oracle.encode_topk(x[rows].float().cpu().numpy(), Wc, bc, bdc, 256)
    del Wc
    for k in (32, 256):
        assert np.array_equal(single[k][1][rows].cpu().numpy(), oi[:, :k].astype(np.int64))
        assert np.array_equal(single[k][0][rows].cpu().numpy(), ov[:, :k])
    yield sae, x, single
    del sae, prepared
    gc.collect()
    torch.cuda.empty_cache()


@pytest.mark.parametrize("G,k,mode", [(2, 32, "topk"), (4, 32, "topk"), (8, 32, "topk"), (2, 32, "candidates"),
                                      (4, 32, "candidates"), (8, 32, "candidates"), (8, 256, "topk"),
                                      (8, 256, "candidates")])
def test_feature_sharded_at_c2_width_equals_single_gpu(dev, c2, G, k, mode):
    """d = 4096, N = 131072, T = 8192, G shards of N / G rows, default k_loc / C: every token's (values, indices) equal
    the single-GPU msae_encode_topk bit for bit (which the fixture pinned to the oracle on 8 tokens)."""
    from msae.parallel import EmulatedShardGroup, default_candidates, default_k_loc

    sae, x, single = c2
    sae.cfg.k = k
    try:
        grp = EmulatedShardGroup(sae, G, mode=mode)
        e0 = grp.engines[0]
        if mode == "topk":
            assert e0.k_loc == default_k_loc(k, G) and e0.n_loc == N_C2 // G
        else:
            assert e0.mode == "candidates" and e0.n_cand == default_candidates(k, G)
        mv, mi, st = grp.encode(x)
        assert grp.mode == mode, "the shape fell back to the other exchange scheme"
        ev, ei = single[k]
        redo = grp.second_round_tokens
        print(f"\nC2 width, G={G} k={k} mode={mode}: "
              + (f"k_loc={e0.k_loc}, second-round tokens {redo}" if mode == "topk" else
                 f"C={e0.n_cand}, exact recompute on {int((st == 1).sum())} tokens"))
        assert int((st >= 2).sum()) == 0
        assert torch.equal(mi, ei), f"{int((mi != ei).any(dim=1).sum())} tokens differ in their indices"
        assert torch.equal(mv, ev)
    finally:
        sae.cfg.k = 32


def test_forced_second_round_at_c2_width(dev, c2):
    """k_loc = 6 at G = 8 truncates below the shards' typical share (4 +- 1.9): hundreds of tokens are flagged, the
    device-sized second round recomputes exactly those, and the result is still the single-GPU one."""
    from msae.parallel import EmulatedShardGroup

    sae, x, single = c2
    grp = EmulatedShardGroup(sae, 8, mode="topk", k_loc=6)
    mv, mi, _ = grp.encode(x)
    redo = grp.second_round_tokens
    print(f"\nk_loc=6 at G=8: second-round tokens {redo} of {T_C2}")
    assert 0 < redo < T_C2
    assert torch.equal(mi, single[32][1]) and torch.equal(mv, single[32][0])


def test_encode_topk_rows_is_the_exact_path_on_a_device_list(dev):
    """msae_encode_topk_rows: listed tokens get the exact path's top-k (== oracle), unlisted rows stay untouched; an
    empty list does nothing; the list and its count never leave the device."""
    from msae import ops

    d, N, T, k = 1024, 16384, 300, 32
    g = torch.Generator(device=dev).manual_seed(5)
    W = torch.randn(N, d, generator=g, device=dev) / d ** 0.5
    b = torch.randn(N, generator=g, device=dev) * 0.05
    bd = torch.randn(d, generator=g, device=dev) * 0.1
    x = torch.randn(T, d, generator=g, device=dev).to(torch.bfloat16)
    flags = torch.zeros(T, dtype=torch.int32, device=dev)
    pick = torch.tensor([0, 7, 8, 129, 255, 299], device=dev)
    flags[pick] = 1
    rows, n = ops.compact_flags(flags)
    assert int(n) == pick.numel() and torch.equal(rows[: pick.numel()].long(), pick)
    vals = torch.full((T, k), -7.0, device=dev)
    idx = torch.full((T, k), -7, dtype=torch.int64, device=dev)
    st = torch.full((T,), -7, dtype=torch.int32, device=dev)
    ops.encode_topk_rows_(x, W, b, bd, rows, n, k, vals, idx, st)
    ov, oi = oracle.encode_topk(x[pick].float().cpu().numpy(), W.cpu().numpy(), b.cpu().numpy(), bd.cpu().numpy(), k)
    assert np.array_equal(idx[pick].cpu().numpy(), oi.astype(np.int64)) and np.array_equal(vals[pick].cpu().numpy(), ov)
    assert (st[pick] == 1).all()
    rest = torch.ones(T, dtype=torch.bool, device=dev)
    rest[pick] = False
    assert (vals[rest] == -7.0).all() and (idx[rest] == -7).all() and (st[rest] == -7).all()
    # hook edits by feature id, and an empty list
    f_set = int(oi[0, 3])
    ops.encode_topk_rows_(x, W, b, bd, rows, n, k, vals, idx, None, set_feature=f_set, set_value=99.0)
    assert int(idx[0, 0]) == f_set and float(vals[0, 0]) == 99.0
    rows0, n0 = ops.compact_flags(torch.zeros(T, dtype=torch.int32, device=dev))
    before = vals.clone()
    ops.encode_topk_rows_(x, W, b, bd, rows0, n0, k, vals, idx, None)
    assert int(n0) == 0 and torch.equal(vals, before)


# ---- "never synchronises", as a test ----------------------------------------------------------------------------------
class _NoSync:
    """Everything enqueued inside the block must not synchronise the host with the device (torch raises if it does)."""

    def __enter__(self):
        torch.cuda.synchronize()
        self.prev = torch.cuda.get_sync_debug_mode()
        torch.cuda.set_sync_debug_mode("error")

    def __exit__(self, *exc):
        torch.cuda.set_sync_debug_mode(self.prev)
        return False


def test_sync_debug_mode_really_raises(dev):
    """The guard itself: a host read of a device value inside the block raises (so the tests below can fail)."""
    t = torch.ones(4, device=dev)
    with pytest.raises(RuntimeError):
        with _NoSync():
            t.sum().item()


@pytest.mark.parametrize("T", [1, 8, 64, 200, 2048])
def test_single_gpu_encode_decode_never_synchronise(dev, T):
    """Sae.encode (+ hook edits) / Sae.decode on every batch-size regime (S = 1 stream, MFMA stream, weight-stream tile,
    one row of tiles, full tiles): no host synchronisation, results equal to a run outside the guard."""
    from msae import Sae, SaeConfig

    d, N, k = 1024, 16384, 32
    sae = Sae(d, SaeConfig(num_latents=N, k=k), device=dev)
    x = hostile.activations(T, d, dev, seed=3)
    with torch.no_grad():
        ref = sae.encode(x)
        ref_r = sae.decode(ref.top_acts, ref.top_indices)        # (also warms the prepared operands and workspaces)
        with _NoSync():
            top, st = sae.encode(x, return_status=True)
            rec = sae.decode(top.top_acts, top.top_indices)
            top2 = sae.encode(x, set_feature=5, set_value=1e4, zero_feature=int(N - 1))
    assert torch.equal(top.top_indices, ref.top_indices) and torch.equal(top.top_acts, ref.top_acts)
    assert torch.equal(rec, ref_r) and int((st >= 2).sum()) == 0
    assert (top2.top_indices == 5).any(dim=1).all()


def test_cache_loop_between_flushes_never_synchronises(dev):
    """FeatureCache's per-batch work (Sae.encode + Cache.add_topk: COO records kept on the device) inside the guard; the
    flush -- the one place the loop crosses to the host -- outside it.  Records equal a plain sparsify."""
    from msae import Sae, SaeConfig, ops
    from msae.features.cache import Cache

    d, N, k, B, S = 512, 8192, 16, 4, 96
    sae = Sae(d, SaeConfig(num_latents=N, k=k), device=dev)
    hs = [hostile.activations(B * S, d, dev, seed=10 + j).view(B, S, d) for j in range(3)]
    with torch.no_grad():
        sae.encode(hs[0])                                        # warm-up: operands, workspaces
        cache = Cache(shard_size=100, batch_size=B)
        cache.add_topk(*sae.encode(hs[0]), N, 0, "warm")
        cache = Cache(shard_size=100, batch_size=B)
        with _NoSync():
            for j, h in enumerate(hs):
                top = sae.encode(h)
                cache.add_topk(top.top_acts, top.top_indices, N, j, "layers.0")
    cache.save()
    loc = cache.feature_locations["layers.0"]
    with torch.no_grad():
        want = [ops.sparsify(*sae.encode(h), N, row_base=j * B + 100) for j, h in enumerate(hs)]
    assert torch.equal(loc, torch.cat([w[0] for w in want]).cpu())
    assert torch.equal(cache.feature_activations["layers.0"], torch.cat([w[1] for w in want]).cpu())


@pytest.mark.parametrize("mode,k_loc", [("topk", None), ("topk", 6), ("candidates", None)])
def test_emulated_sharded_encode_never_synchronises(dev, mode, k_loc):
    """The feature-sharded encode (G = 8) -- local encodes, packs, merge, the SECOND ROUND with and without flagged tokens,
    candidate records and the owner-side re-score -- enqueues without a host synchronisation."""
    from msae import Sae, SaeConfig, ops
    from msae.parallel import EmulatedShardGroup

    d, N, k, T, G = 1024, 65536, 32, 1024, 8
    sae = Sae(d, SaeConfig(num_latents=N, k=k), device=dev)
    x = hostile.activations(T, d, dev, seed=4)
    kw = {} if k_loc is None else {"k_loc": k_loc}
    grp = EmulatedShardGroup(sae, G, mode=mode, **kw)
    with torch.no_grad():
        ev, ei, _ = ops.encode_topk(x, sae.encoder.weight, sae.encoder.bias, sae.b_dec, ops.prepare_encoder(sae.encoder.weight), k)
        grp.encode(x)                                            # warm-up
        with _NoSync():
            mv, mi, st = grp.encode(x)
            rec = grp.decode(mv, mi)
    assert grp.mode == mode
    assert torch.equal(mi, ei) and torch.equal(mv, ev)
    assert torch.equal(rec, ops.decode(ei, ev, sae.W_dec, sae.b_dec))
    if k_loc == 6:
        assert grp.second_round_tokens > 0

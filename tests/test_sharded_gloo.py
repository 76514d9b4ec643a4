"""CPU, world_size=2 over gloo: the feature-sharded encode (per-shard TopK + all-gather + merge) and
the token-sharded decode reproduce the single-shard result bit for bit.  The local kernels are
replaced by the oracle here (test infrastructure); the collective / merge / slicing logic under test
is the product code in msae/parallel.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, T, d, N, k, out_dir, k_loc=None, cluster=False, diverge=False, edits=None):
    for p in (REPO, REPO / "tests", REPO / "multimodal-sae_amd"):
        sys.path.insert(0, str(p))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    from oracle import oracle
    from msae.parallel import ShardedSae, shutdown

    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed=31)
    if cluster:
        b_enc = b_enc.copy()
        b_enc[: N // world // 4] += 3.0   # most of the global top-k lives in shard 0
    x = synth.activations(T, d, seed=32)
    n_loc = N // world
    lo, hi = rank * n_loc, (rank + 1) * n_loc

    def encode_fn(xt, kk, set_feature=-1, set_value=0.0, zero_feature=-1):
        # an injected encode_fn receives the hooks' GLOBAL feature ids (the product's default one maps them itself)
        loc = lambda f: f - lo if lo <= f < hi else -1
        v, i = oracle.encode_topk(xt.numpy(), W_enc[lo:hi], b_enc[lo:hi], b_dec, kk, set_feature=loc(set_feature),
                                  set_value=set_value, zero_feature=loc(zero_feature))
        return torch.from_numpy(v), torch.from_numpy(i).long(), torch.zeros(len(v), dtype=torch.int32)

    def decode_fn(idx, vals):
        return torch.from_numpy(oracle.decode(idx.numpy(), vals.numpy(), W_dec, b_dec))

    eng = ShardedSae(torch.from_numpy(W_enc[lo:hi]), torch.from_numpy(b_enc[lo:hi]),
                     torch.from_numpy(W_dec), torch.from_numpy(b_dec), k, rank=rank, world=world,
                     group=dist.group.WORLD, encode_fn=encode_fn, decode_fn=decode_fn, k_loc=k_loc,
                     local_decode_max_t=0,      # (a handful of tokens would be decoded locally: exercise the sharded decode)
                     broadcast_input=diverge)
    xt = torch.from_numpy(x)
    if diverge and rank > 0:                    # this rank's "LLM forward" came out different: rank 0's input must win
        xt = xt + 0.25 * torch.randn(xt.shape, generator=torch.Generator().manual_seed(rank))
    if diverge and os.environ.get("MSAE_DEBUG_SHARD_CHECK"):
        eng.broadcast_input = False             # debug mode instead: the mismatch must be DETECTED
        try:
            eng.forward(xt)
            caught = False
        except RuntimeError as e:
            caught = "different activations" in str(e)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), caught=caught)
        shutdown(eng)
        return
    if edits:
        assert not xt.requires_grad
        keep = xt.clone()
        v, i, _ = eng.encode(xt, **edits)
        assert torch.equal(xt, keep)            # the caller's activations are never written (broadcast_input: private buffer)
        out = {"top_acts": v, "top_indices": i, "sae_out": eng.decode(v, i)}
    else:
        out = eng.forward(xt)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), v=out["top_acts"].numpy(),
             i=out["top_indices"].numpy(), r=out["sae_out"].numpy(), redo=eng.second_round_tokens,
             k_loc=eng.k_loc)
    shutdown(eng)


@pytest.mark.parametrize("T,k,k_loc,cluster", [(13, 8, None, False), (8, 8, None, False),
                                               (13, 16, 10, False), (13, 16, 9, True)])
def test_feature_sharded_equals_single_shard(tmp_path, T, k, k_loc, cluster):
    """k_loc < k: per-shard truncation verified after the merge; `cluster` concentrates the global
    top-k in one shard so the second round (full local top-k for the flagged tokens) must run."""
    import synth
    from oracle import oracle

    d, N, world = 64, 1024, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, T, d, N, k, str(tmp_path), k_loc, cluster), nprocs=world, join=True)
    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed=31)
    if cluster:
        b_enc = b_enc.copy()
        b_enc[: N // world // 4] += 3.0
    x = synth.activations(T, d, seed=32)
    ref_v, ref_i = oracle.encode_topk(x, W_enc, b_enc, b_dec, k)
    ref_r = oracle.decode(ref_i, ref_v, W_dec, b_dec)
    for rank in range(world):
        g = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(g["i"], ref_i), f"rank {rank}: merged indices differ"
        assert np.array_equal(g["v"], ref_v)
        assert np.array_equal(g["r"], ref_r)
        if cluster:
            assert int(g["redo"]) > 0 and int(g["k_loc"]) == k_loc   # the second round really ran


@pytest.mark.parametrize("world,T,k,k_loc,cluster,edits,diverge", [
    (4, 5, 8, None, False, None, False),
    (4, 257, 16, 6, True, None, False),                      # forced second round, T % G != 0
    (8, 1, 8, None, False, None, False),                     # T < G: seven ranks own no token of the decode
    (8, 13, 16, 4, True, None, False),
    (8, 16, 8, None, False, {"set_feature": 777, "set_value": 9.5, "zero_feature": 130}, True),   # hook edits + broadcast_input
    (4, 13, 8, 3, True, {"zero_feature": 5}, False),
])
def test_feature_sharded_world_4_and_8(tmp_path, world, T, k, k_loc, cluster, edits, diverge):
    """Round-4 verdict, item 3: the per-shard top-k scheme over gloo at world 4 and 8 -- token counts below G and not
    divisible by G (ranks with an empty token slice in the sharded decode), the truncated lists with a forced second round,
    the hooks' edits by global feature id, broadcast_input with diverged ranks: every rank's merged result and reconstruction
    equal the single-shard oracle bit for bit."""
    import synth
    from oracle import oracle

    d, N = 64, 1024
    mp.spawn(_worker, args=(world, _free_port(), T, d, N, k, str(tmp_path), k_loc, cluster, diverge, edits),
             nprocs=world, join=True)
    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed=31)
    if cluster:
        b_enc = b_enc.copy()
        b_enc[: N // world // 4] += 3.0
    x = synth.activations(T, d, seed=32)
    ref_v, ref_i = oracle.encode_topk(x, W_enc, b_enc, b_dec, k, **(edits or {}))
    ref_r = oracle.decode(ref_i, ref_v, W_dec, b_dec)
    for rank in range(world):
        g = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(g["i"], ref_i), f"rank {rank}: merged indices differ"
        assert np.array_equal(g["v"], ref_v) and np.array_equal(g["r"], ref_r), rank
        if cluster and k_loc is not None:
            assert int(g["redo"]) > 0 and int(g["k_loc"]) == max(k_loc, -(-k // world))


def test_broadcast_input_makes_diverged_ranks_agree(tmp_path, monkeypatch):
    """`--shard-sae` runs one LLM forward per rank; if a rank's hidden state differs (nondeterministic
    kernel, sampling) the merge would combine results of different inputs.  broadcast_input=True: rank 0's
    activations are what every shard encodes -- both ranks return the single-shard result of RANK 0's x."""
    import synth
    from oracle import oracle

    T, d, N, k, world = 13, 64, 1024, 8, 2
    mp.spawn(_worker, args=(world, _free_port(), T, d, N, k, str(tmp_path), None, False, True), nprocs=world, join=True)
    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed=31)
    ref_v, ref_i = oracle.encode_topk(synth.activations(T, d, seed=32), W_enc, b_enc, b_dec, k)
    for rank in range(world):
        g = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(g["i"], ref_i) and np.array_equal(g["v"], ref_v), rank
    # without the broadcast, MSAE_DEBUG_SHARD_CHECK=1 detects the mismatch on every rank instead of merging silently
    monkeypatch.setenv("MSAE_DEBUG_SHARD_CHECK", "1")
    mp.spawn(_worker, args=(world, _free_port(), T, d, N, k, str(tmp_path), None, False, True), nprocs=world, join=True)
    for rank in range(world):
        assert bool(np.load(tmp_path / f"rank{rank}.npz")["caught"]), rank


def test_merge_topk_is_canonical_with_ties():
    from msae.parallel import merge_topk

    vals = torch.tensor([[1.0, 0.0, 2.0, 1.0, 0.0, -0.0, 2.0, -3.0]])
    idx = torch.tensor([[9, 4, 7, 3, 1, 0, 5, 2]])
    v, i = merge_topk(vals, idx, 6)
    assert i.tolist() == [[5, 7, 3, 9, 0, 1]] and v.tolist() == [[2.0, 2.0, 1.0, 1.0, -0.0, 0.0]]
    v, i = merge_topk(vals, idx, 8)
    assert i.tolist()[0][-2:] == [4, 2]


# ---- mode="candidates": per-shard candidate records, all-to-all, owner-side re-score ------------------------------
def _order_key(v: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
    b = np.where(b == 0x80000000, 0, b)
    return np.where(b & 0x80000000, (~b) & 0xFFFFFFFF, b | 0x80000000)


def _cand_worker(rank, world, port, T, d, N, k, C, out_dir, cluster):
    for p in (REPO, REPO / "tests", REPO / "multimodal-sae_amd"):
        sys.path.insert(0, str(p))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    from oracle import oracle
    from msae.parallel import ShardedSae, shutdown

    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed=31)
    if cluster:
        b_enc = b_enc.copy()
        b_enc[: N // world // 4] += 3.0
    x = synth.activations(T, d, seed=32)
    n_loc = N // world
    lo, hi = rank * n_loc, (rank + 1) * n_loc
    stride = 12 * C + 8
    stats = {"fallback": 0}

    def cand_fn(xt):
        """records of this shard as msae_shard_candidates lays them out (exact values as upper values, z sigma 0)"""
        pre = oracle.pre_acts(xt.numpy(), W_enc[lo:hi], b_enc[lo:hi], b_dec)
        rec = np.zeros((len(pre), stride), dtype=np.uint8)
        for t, row in enumerate(pre):
            key = (_order_key(row) << np.uint64(32)) | (np.uint64(0x7FFFFFFF) - (np.arange(lo, hi, dtype=np.uint64)))
            order = np.argsort(key)[::-1]
            rec[t, : 8 * C] = key[order[:C]].view(np.uint8)
            tail = np.array([row[order[C]] if len(order) > C else 0.0, 0.0], dtype=np.float32)
            rec[t, 12 * C:] = tail.view(np.uint8)
        return torch.from_numpy(rec)

    def rescore_fn(xl, recv, t_valid):
        G, per, _ = recv.shape
        r = recv.numpy()
        vals, idx = np.zeros((per, k), np.float32), np.zeros((per, k), np.int64)
        for t in range(t_valid):
            keys = np.concatenate([r[g, t, : 8 * C].view(np.uint64) for g in range(G)])
            tau = max(float(r[g, t, 12 * C: 12 * C + 4].view(np.float32)[0]) for g in range(G))
            feats = (0x7FFFFFFF - (keys[keys != 0] & np.uint64(0xFFFFFFFF)).astype(np.int64))
            pre = oracle.pre_acts(xl[t:t + 1].numpy(), W_enc, b_enc, b_dec)[0]
            ckey = (_order_key(pre[feats]) << np.uint64(32)) | (np.uint64(0x7FFFFFFF) - feats.astype(np.uint64))
            top = feats[np.argsort(ckey)[::-1][:k]]
            if len(top) < k or not pre[top[-1]] > tau:       # the lists cannot prove the top-k: exact recompute
                stats["fallback"] += 1
                v, i = oracle.encode_topk(xl[t:t + 1].numpy(), W_enc, b_enc, b_dec, k)
                vals[t], idx[t] = v[0], i[0]
            else:
                vals[t], idx[t] = pre[top], top
        return torch.from_numpy(vals), torch.from_numpy(idx), torch.zeros(per, dtype=torch.int32)

    def decode_fn(idx, vals):
        return torch.from_numpy(oracle.decode(idx.numpy(), vals.numpy(), W_dec, b_dec))

    eng = ShardedSae(torch.from_numpy(W_enc[lo:hi]), torch.from_numpy(b_enc[lo:hi]), torch.from_numpy(W_dec),
                     torch.from_numpy(b_dec), k, rank=rank, world=world, group=dist.group.WORLD,
                     encode_fn=lambda *a: None, decode_fn=decode_fn, mode="candidates", n_cand=C,
                     cand_fn=cand_fn, rescore_fn=rescore_fn, local_decode_max_t=0)
    out = eng.forward(torch.from_numpy(x))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), v=out["top_acts"].numpy(), i=out["top_indices"].numpy(),
             r=out["sae_out"].numpy(), fallback=stats["fallback"])
    shutdown(eng)


@pytest.mark.parametrize("world,T,k,C,cluster", [(4, 5, 8, 8, False), (4, 257, 16, 8, True), (8, 1, 8, 8, False),
                                                 (8, 13, 16, 8, True), (8, 16, 8, 16, False)])
def test_candidate_exchange_world_4_and_8(tmp_path, world, T, k, C, cluster):
    """mode="candidates" over gloo at world 4 and 8: the all-to-all's padded token slices (T = 1: seven ranks re-score
    nothing; 13 and 257 do not divide), the result all-gather, owners whose lists overflow (cluster)."""
    import synth
    from oracle import oracle

    d, N = 64, 1024
    mp.spawn(_cand_worker, args=(world, _free_port(), T, d, N, k, C, str(tmp_path), cluster), nprocs=world, join=True)
    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed=31)
    if cluster:
        b_enc = b_enc.copy()
        b_enc[: N // world // 4] += 3.0
    x = synth.activations(T, d, seed=32)
    ref_v, ref_i = oracle.encode_topk(x, W_enc, b_enc, b_dec, k)
    ref_r = oracle.decode(ref_i, ref_v, W_dec, b_dec)
    for rank in range(world):
        g = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(g["i"], ref_i) and np.array_equal(g["v"], ref_v) and np.array_equal(g["r"], ref_r), rank


@pytest.mark.parametrize("T,k,C,cluster", [(13, 8, 8, False), (8, 8, 16, False), (13, 16, 8, True)])
def test_candidate_exchange_equals_single_shard(tmp_path, T, k, C, cluster):
    """ShardedSae(mode="candidates") over gloo, world 2: record layout, all-to-all slicing (13 tokens do not divide
    by 2: the padded token), owner-side results and their all-gather, token-sharded decode.  `cluster` puts more of
    the global top-k into shard 0 than its C records hold, so the owner must take the exact recompute."""
    import synth
    from oracle import oracle

    d, N, world = 64, 1024, 2
    port = _free_port()
    mp.spawn(_cand_worker, args=(world, port, T, d, N, k, C, str(tmp_path), cluster), nprocs=world, join=True)
    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed=31)
    if cluster:
        b_enc = b_enc.copy()
        b_enc[: N // world // 4] += 3.0
    x = synth.activations(T, d, seed=32)
    ref_v, ref_i = oracle.encode_topk(x, W_enc, b_enc, b_dec, k)
    ref_r = oracle.decode(ref_i, ref_v, W_dec, b_dec)
    fb = 0
    for rank in range(world):
        g = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(g["i"], ref_i), f"rank {rank}: indices differ"
        assert np.array_equal(g["v"], ref_v)
        assert np.array_equal(g["r"], ref_r)
        fb += int(g["fallback"])
    if cluster:
        assert fb > 0

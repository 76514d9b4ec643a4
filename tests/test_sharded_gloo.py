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


def _worker(rank, world, port, T, d, N, k, out_dir, k_loc=None, cluster=False):
    for p in (REPO, REPO / "tests", REPO / "multimodal-sae_amd"):
        sys.path.insert(0, str(p))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    from oracle import oracle
    from msae.parallel import ShardedSae

    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed=31)
    if cluster:
        b_enc = b_enc.copy()
        b_enc[: N // world // 4] += 3.0   # most of the global top-k lives in shard 0
    x = synth.activations(T, d, seed=32)
    n_loc = N // world
    lo, hi = rank * n_loc, (rank + 1) * n_loc

    def encode_fn(xt, kk):
        v, i = oracle.encode_topk(xt.numpy(), W_enc[lo:hi], b_enc[lo:hi], b_dec, kk)
        return torch.from_numpy(v), torch.from_numpy(i).long(), torch.zeros(len(v), dtype=torch.int32)

    def decode_fn(idx, vals):
        return torch.from_numpy(oracle.decode(idx.numpy(), vals.numpy(), W_dec, b_dec))

    eng = ShardedSae(torch.from_numpy(W_enc[lo:hi]), torch.from_numpy(b_enc[lo:hi]),
                     torch.from_numpy(W_dec), torch.from_numpy(b_dec), k, rank=rank, world=world,
                     group=dist.group.WORLD, encode_fn=encode_fn, decode_fn=decode_fn, k_loc=k_loc)
    out = eng.forward(torch.from_numpy(x))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), v=out["top_acts"].numpy(),
             i=out["top_indices"].numpy(), r=out["sae_out"].numpy(), redo=eng.second_round_tokens,
             k_loc=eng.k_loc)
    dist.barrier()
    dist.destroy_process_group()


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


def test_merge_topk_is_canonical_with_ties():
    from msae.parallel import merge_topk

    vals = torch.tensor([[1.0, 0.0, 2.0, 1.0, 0.0, -0.0, 2.0, -3.0]])
    idx = torch.tensor([[9, 4, 7, 3, 1, 0, 5, 2]])
    v, i = merge_topk(vals, idx, 6)
    assert i.tolist() == [[5, 7, 3, 9, 0, 1]] and v.tolist() == [[2.0, 2.0, 1.0, 1.0, -0.0, 0.0]]
    v, i = merge_topk(vals, idx, 8)
    assert i.tolist()[0][-2:] == [4, 2]

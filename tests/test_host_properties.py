"""Property tests (hypothesis) of the host-side logic around the hot path: the canonical order key,
the merge of per-shard TopK lists, the k_loc truncation rule and the split-index rule of the cache
writer.  CPU only."""
from __future__ import annotations

import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle import oracle
from msae.parallel import canonical_key, merge_topk, token_slice
from msae.features.cache import FeatureCache

finite = st.floats(min_value=-1e6, max_value=1e6, allow_nan=False, allow_infinity=False, width=32)


@settings(max_examples=200, deadline=None)
@given(st.lists(finite, min_size=2, max_size=40), st.data())
def test_canonical_key_orders_by_value_then_index(vals, data):
    idx = data.draw(st.lists(st.integers(0, 2 ** 31 - 2), min_size=len(vals), max_size=len(vals), unique=True))
    v, i = torch.tensor(vals, dtype=torch.float32), torch.tensor(idx, dtype=torch.int64)
    key = canonical_key(v, i)
    order = torch.argsort(key, descending=True).tolist()
    for a, b in zip(order, order[1:]):
        va, vb = float(v[a]), float(v[b])
        assert va > vb or (va == vb and idx[a] < idx[b])       # -0.0 == +0.0 ties break by index


@settings(max_examples=100, deadline=None)
@given(st.integers(1, 6), st.integers(1, 8), st.integers(1, 5), st.integers(0, 2 ** 31 - 1))
def test_merge_of_shard_topk_equals_global_topk(G, k, T, seed):
    """global top-k (value desc, index asc) == merge of the per-shard top-k lists."""
    rng = np.random.default_rng(seed)
    n_loc = k + int(rng.integers(0, 5))
    # few distinct values: plenty of ties inside and across shards
    dense = rng.choice(np.array([0.0, 0.5, 1.0, 1.5, 2.0], dtype=np.float32), size=(T, G * n_loc))
    ref_v, ref_i = oracle.topk(dense, k)
    sv, si = [], []
    for g in range(G):
        v, i = oracle.topk(np.ascontiguousarray(dense[:, g * n_loc:(g + 1) * n_loc]), k)
        sv.append(torch.from_numpy(v)); si.append(torch.from_numpy(i.astype(np.int64)) + g * n_loc)
    mv, mi = merge_topk(torch.cat(sv, 1), torch.cat(si, 1), k)
    assert np.array_equal(mv.numpy(), ref_v) and np.array_equal(mi.numpy().astype(np.int32), ref_i)


@settings(max_examples=200, deadline=None)
@given(st.integers(1, 10 ** 6), st.integers(1, 64))
def test_token_slices_partition_the_batch(T, world):
    covered = 0
    for r in range(world):
        lo, hi, per = token_slice(T, r, world)
        assert lo == min(r * per, T) and lo <= hi <= T
        covered += hi - lo
    assert covered == T


@settings(max_examples=200, deadline=None)
@given(st.integers(2, 1 << 19), st.integers(1, 256))
def test_split_indices_follow_the_reference_rule(width, n_splits):
    """boundaries = linspace(0, width, n+1).long(); split i = (b[i], b[i+1] - 1)
    (sae_auto_interp/features/cache.py:243-247)."""
    fc = FeatureCache.__new__(FeatureCache)
    fc.width = width
    got = fc._generate_split_indices(n_splits)
    b = torch.linspace(0, width, steps=n_splits + 1).long()
    assert [(int(s), int(e)) for s, e in got] == list(zip(b[:-1].tolist(), (b[1:] - 1).tolist()))

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


def test_subtractive_dither_identities_and_residual_distribution():
    """The arithmetic behind the large-batch int8 pass's band (csrc/encode_defs.h "subtractive dither"),
    restated in numpy -- no GPU: (1) with shared dither vectors the subtracted product equals the integer accumulator minus a
    per-feature constant D_n, a per-token constant E_t, plus a scalar F -- exactly, in integers of 2^-34; (2) the residual of a
    subtractively dithered element is uniform on (-1/2, 1/2] WHATEVER the value is (here: values chosen adversarially at
    k + 1/2, where round to nearest is worst and stochastic rounding has its largest variance); (3) the error of a coarse value
    is exactly sum_c (A_c + delta_c) eps_c + delta_c W_c -- the cross term of the two roundings sits inside the first sum; (4) its
    standard deviation is that of the band's sigma (1/12 per rounding)."""
    import numpy as np

    rng = np.random.default_rng(7)
    d, T, N = 2048, 48, 64
    hx, hw = rng.integers(0, 1 << 16, d), rng.integers(0, 1 << 16, d)
    gx, gw = 2 * hx + 1 - (1 << 16), 2 * hw + 1 - (1 << 16)            # units of 2^-17: the dither minus one half
    rx, rw = (2 * hx + 1) / 131072.0, (2 * hw + 1) / 131072.0
    A = rng.normal(0, 30, (T, d))
    A[0] = np.floor(A[0]) + 0.5                                        # adversarial token: every element half way between two steps
    W = rng.normal(0, 30, (N, d))
    q = np.floor(A + rx).astype(np.int64)
    w = np.floor(W + rw).astype(np.int64)
    acc = q @ w.T
    D = (w @ gx)                                                       # [N], units 2^-17
    E = (q @ gw)                                                       # [T], units 2^-17
    F = int((gx * gw).sum())                                           # units 2^-34
    # (1) the identity, in exact integer arithmetic (everything scaled by 2^34)
    lhs = ((q * 131072 - gx).astype(object) @ (w * 131072 - gw).astype(object).T)
    rhs = acc.astype(object) * (1 << 34) - D.astype(object)[None, :] * (1 << 17) - E.astype(object)[:, None] * (1 << 17) + F
    assert (lhs == rhs).all()
    # (2) residuals: in (-1/2, 1/2], mean ~0, variance 1/12 -- also on the adversarial token
    delta = q - (rx - 0.5) - A
    eps = w - (rw - 0.5) - W
    for r in (delta, delta[0], eps):
        assert r.min() > -0.5 - 1e-9 and r.max() <= 0.5 + 1e-9
        assert abs(r.mean()) < 0.02 and abs(r.var() - 1.0 / 12.0) < 0.01
    # (3) the error decomposition, to float accuracy
    coarse = lhs.astype(np.float64) / float(1 << 34)
    exact = A @ W.T
    err = coarse - exact
    pred = (A + delta) @ eps.T + delta @ W.T
    assert np.allclose(err, pred, rtol=0, atol=1e-6 * np.abs(exact).max())
    # (4) its scale: sigma^2 = (|A + delta|^2 + |W|^2) / 12 per pair
    sig = np.sqrt((((A + delta) ** 2).sum(1)[:, None] + (W ** 2).sum(1)[None, :]) / 12.0)
    ratio = err / sig
    assert 0.9 < ratio.std() < 1.1 and np.abs(ratio).max() < 5.0
    # non-subtractive stochastic rounding of the SAME adversarial token: variance 1/4 per element, three times as much
    assert abs((q[0] - A[0]).var() - 0.25) < 0.01

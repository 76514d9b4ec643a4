"""CPU: the oracle (oracle/) is pinned against fixtures produced by the reference itself
(tests/golden/make_golden.py).  Tolerances are the ones stated in DESIGN.md section 4:

  pre-activations / top-k activations : |a - b| <= 1e-4 * max(1, |b|)   (fp32 summation order)
  reconstruction                      : |a - b| <= 1e-4 * max|recon|
  top-k indices                       : identical, in canonical (value desc, index asc) order, on
                                        every row whose reference (k, k+1) gap exceeds EPS_GAP;
                                        as a set restricted to act > 0
"""
import functools

import numpy as np
import pytest

import synth
from oracle import oracle

EPS_GAP = 2e-5
RTOL = 1e-4


def _close(a, b, scale=None):
    scale = np.maximum(1.0, np.abs(b)) if scale is None else scale
    return np.all(np.abs(a - b) <= RTOL * scale)


@functools.lru_cache(maxsize=1)
def _weights(d, N, seed):
    """(the two full-C2 fixtures share their weight seed: 4 GB of counter-based weights are generated once per run)"""
    return synth.sae_weights(d, N, seed)


def _load(golden_dir, name):
    g = np.load(golden_dir / f"{name}.npz")
    d, N, T = int(g["d"]), int(g["N"]), int(g["T"])
    W = _weights(d, N, int(g["wseed"]))
    x = synth.activations(T, d, int(g["xseed"]))
    return g, W, x


@pytest.mark.parametrize("name", ["g1_c1_d768_n4096", "g2_d4096_n16384", "g2_c2_d4096_n131072"])
def test_encode_topk_decode_matches_reference(golden_dir, name):
    """The last case is BASELINE configs[1] at full width (weight generation takes ~a minute)."""
    g, (W_enc, b_enc, W_dec, b_dec), x = _load(golden_dir, name)
    pre = oracle.pre_acts(x, W_enc, b_enc, b_dec)
    assert _close(pre[:8, :256], g["pre_slice"])
    assert np.allclose(pre.astype(np.float64).sum(-1), g["pre_rowsum"], rtol=1e-5)
    assert np.array_equal((pre > 0).sum(-1), g["pre_nnz"]) or \
        np.abs((pre > 0).sum(-1) - g["pre_nnz"]).max() <= 2  # sign flips of ~0 pre-acts
    for k in g["ks"]:
        vals, idx = oracle.topk(pre, int(k))
        ref_v, ref_i, gap = g[f"k{k}_acts"], g[f"k{k}_idx"], g[f"k{k}_gap"]
        assert _close(vals, ref_v)
        safe = gap > EPS_GAP
        assert safe.mean() > 0.9
        for t in np.nonzero(safe)[0]:
            pos = ref_v[t] > 0
            assert set(idx[t][vals[t] > 0]) == set(ref_i[t][pos])
        # canonical order equality wherever neighbouring reference values are separated
        sep = np.all(np.abs(np.diff(ref_v, axis=1)) > EPS_GAP, axis=1) & safe
        assert np.array_equal(idx[sep], ref_i[sep])
        recon = oracle.decode(idx, vals, W_dec, b_dec)
        ref_r = g[f"k{k}_recon"]
        assert _close(recon[safe], ref_r[safe], scale=np.abs(ref_r).max())
        # fused entry point == unfused
        v2, i2 = oracle.encode_topk(x, W_enc, b_enc, b_dec, int(k))
        assert np.array_equal(i2, idx) and np.array_equal(v2, vals)


def check_large_fixture(g, k, vals, idx, recon):
    """g13's compact encode / decode record vs one implementation's (canonical) top-k and reconstruction; shared with the
    GPU test (tests/test_gpu_parity.py).  Same rules as above."""
    ref_v, ref_i, gap = g[f"k{k}_acts"], g[f"k{k}_idx"], g[f"k{k}_gap"]
    assert _close(vals, ref_v)
    safe = gap > EPS_GAP
    assert safe.mean() > 0.9
    pos_v, pos_r = vals > 0, ref_v > 0
    for t in np.nonzero(safe)[0]:
        assert set(idx[t][pos_v[t]]) == set(ref_i[t][pos_r[t]]), t
    sep = np.all(np.abs(np.diff(ref_v, axis=1)) > EPS_GAP, axis=1) & safe
    assert np.array_equal(idx[sep], ref_i[sep])
    pos_ok = np.ones_like(ref_i, dtype=bool)
    pos_ok[:, 1:] &= np.abs(np.diff(ref_v, axis=1)) > EPS_GAP
    pos_ok[:, :-1] &= np.abs(np.diff(ref_v, axis=1)) > EPS_GAP
    pos_ok &= safe[:, None]
    assert np.array_equal(idx[pos_ok], ref_i[pos_ok])
    rows = g["recon_rows_at"]
    ref_rows = g[f"k{k}_recon_rows"]
    scale = np.abs(ref_rows).max()
    ok = safe[rows]
    assert np.all(np.abs(recon[rows][ok] - ref_rows[ok]) <= RTOL * scale)
    # every token's reconstruction through its sum (f64 of the f32 row): |sum error| <= RTOL * sum |recon|
    s = recon.astype(np.float64).sum(-1)
    assert np.all(np.abs(s[safe] - g[f"k{k}_recon_sum"][safe]) <= RTOL * g[f"k{k}_recon_abs"][safe])
    return int(safe.sum())


@pytest.mark.parametrize("name", ["g13_c2_d4096_n131072_t320", "g13_d4096_n16384_t1024"])
def test_large_batch_fixture_matches_reference(golden_dir, name):
    """g13: the reference's encode / decode at T = 1024 (N = 16384) and at the FULL C2 shape with T = 320 -- the batch sizes
    whose HIP kernels bench.py times; the GPU counterpart is test_benchmarked_kernels_match_reference_fixture."""
    g, (W_enc, b_enc, W_dec, b_dec), x = _load(golden_dir, name)
    for k in g["ks"]:
        k = int(k)
        vals, idx = oracle.encode_topk(x, W_enc, b_enc, b_dec, k)
        recon = oracle.decode(idx, vals, W_dec, b_dec)
        check_large_fixture(g, k, vals, idx, recon)


def test_decode_seam_reference_test(golden_dir):
    """train/sae/tests/test_decode.py:6-20: sparse decode == eager decode."""
    g = np.load(golden_dir / "g3_decode_seam.npz")
    out = oracle.decode(g["top_idx"], g["top_vals"], g["W_dec"], None)
    np.testing.assert_allclose(out, g["eager"], rtol=1.3e-6, atol=1e-5)  # assert_allclose defaults


def test_topk_ties_and_zeros_are_canonical():
    lat = np.zeros((3, 50), dtype=np.float32)
    lat[0, [7, 3, 40]] = [2.0, 2.0, 5.0]          # tie between 3 and 7 -> 3 first
    lat[1, :] = 0.0                                # all zeros -> indices 0..k-1
    lat[2, [49, 0]] = [1.0, 1.0]
    v, i = oracle.topk(lat, 4)
    assert i[0].tolist() == [40, 3, 7, 0] and v[0].tolist() == [5.0, 2.0, 2.0, 0.0]
    assert i[1].tolist() == [0, 1, 2, 3]
    assert i[2].tolist() == [0, 49, 1, 2]


def test_cache_sparsify_matches_reference(golden_dir):
    g = np.load(golden_dir / "g4_cache.npz")
    d, N, k = int(g["d"]), int(g["N"]), int(g["k"])
    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, int(g["wseed"]))
    x = g["x"]
    B, S, _ = x.shape
    vals, idx = oracle.encode_topk(x.reshape(B * S, d), W_enc, b_enc, b_dec, k)
    row_base = 5 * 2 + 100  # batch_number*batch_size + shard_size, cache.py:55
    loc, act = oracle.sparsify(vals, idx, B, S, row_base=row_base)
    assert np.array_equal(loc, g["nofilter_locations"])
    np.testing.assert_allclose(act, g["nofilter_activations"], rtol=1e-5)
    bitmap = np.zeros(N, dtype=np.uint8)
    bitmap[g["filter_features"]] = 1
    loc, act = oracle.sparsify(vals, idx, B, S, row_base=row_base, filter_bitmap=bitmap)
    assert np.array_equal(loc, g["filter_locations"])
    np.testing.assert_allclose(act, g["filter_activations"], rtol=1e-5)


def test_hooks_match_reference(golden_dir):
    g = np.load(golden_dir / "g5_hooks.npz")
    d, N, k = int(g["d"]), int(g["N"]), int(g["k"])
    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, int(g["wseed"]))
    for S in (5, 1):
        x = g[f"steer_S{S}_x"][0].astype(np.float32)
        feat, clamp = int(g[f"steer_S{S}_feature"]), float(g[f"steer_S{S}_clamp"])
        v, i = oracle.encode_topk(x, W_enc, b_enc, b_dec, k, set_feature=feat if S != 1 else -1,
                                  set_value=clamp)
        out = oracle.decode(i, v, W_dec, b_dec).astype(np.float16)
        ref = g[f"steer_S{S}_out"][0]
        assert np.abs(out.astype(np.float32) - ref.astype(np.float32)).max() <= 2e-3 * np.abs(ref).max()
    x = g["attr_x"].reshape(-1, d).astype(np.float32)
    for tag, off in (("none", -1), ("off", int(g["attr_off_feature"]))):
        v, i = oracle.encode_topk(x, W_enc, b_enc, b_dec, k, zero_feature=off)
        out = oracle.decode(i, v, W_dec, b_dec).astype(np.float16).reshape(g["attr_x"].shape)
        ref = g[f"attr_{tag}_out"]
        assert np.abs(out.astype(np.float32) - ref.astype(np.float32)).max() <= 2e-3 * np.abs(ref).max()


def test_decode_backward_matches_reference_autograd(golden_dir):
    g = np.load(golden_dir / "g7_train.npz")
    d, N = int(g["d"]), int(g["N"])
    _, _, W_dec, _ = synth.sae_weights(d, N, int(g["wseed"]))
    ga = oracle.decode_bwd_acts(g["dec_idx"], g["dec_gout"], W_dec)
    np.testing.assert_allclose(ga, g["dec_grad_acts"], rtol=1e-4, atol=1e-5)


def test_refport_matches_reference(golden_dir):
    """RefPort (torch-CPU operators, the timed cpu_baseline) reproduces the reference outputs."""
    import torch

    g, (W_enc, b_enc, W_dec, b_dec), x = _load(golden_dir, "g1_c1_d768_n4096")
    port = oracle.RefPort(W_enc, b_enc, W_dec, b_dec, 32)
    xt = torch.from_numpy(x).to(torch.bfloat16)
    recon, acts, idx = port.forward(xt)
    order = np.lexsort((idx.numpy(), -acts.numpy().astype(np.float64)), axis=-1)
    assert np.array_equal(np.take_along_axis(idx.numpy(), order, -1), g["k32_idx"])
    np.testing.assert_allclose(recon.numpy(), g["k32_recon"], rtol=1e-5, atol=1e-5)

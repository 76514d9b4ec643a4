"""GPU parity of the backward / optimiser / attribution kernels AT THE SIZES BASELINE.json's configs name
(round-2 verdict: these kernels were only ever checked at toy sizes).

  configs[3]  training step at C2: d = 4096, N = 131072, T = 8192 per GPU
              (reference train/sae/sae/trainer.py:347-401, sae/kernels.py:10-175)
  configs[4]  width 262144 + grad x act attribution scoring (reference sae/kernels.py:287-400 used as
              features/patching/attribution.py:133-183 would)

References here are the CPU oracle where it finishes in seconds (the scorer: 33 792 dot products), otherwise a
dense / float64 torch restatement on the same GPU (torch is test infrastructure here, never the product path).
Every test frees what it allocated: the box has 288 GB of HBM, the suite must not depend on it.
"""
import gc

import numpy as np
import pytest
import torch

from oracle import oracle

pytestmark = pytest.mark.gpu

D, N_C2, N_C5 = 4096, 131072, 262144


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


def _unit_rows(N, d, dev, seed):
    """[N, d] f32 with unit-norm rows, generated in 8192-row blocks."""
    W = torch.empty(N, d, device=dev)
    for b0 in range(0, N, 8192):
        g = torch.Generator(device=dev).manual_seed(seed * 1000 + b0 // 8192)
        w = torch.randn(8192, d, generator=g, device=dev)
        W[b0:b0 + 8192] = w / w.norm(dim=1, keepdim=True)
    return W


# ---- configs[4]: the batched scorer's primitive at width 262144 ---------------------------------------------------
def test_decode_bwd_acts_at_width_262144_vs_oracle(dev):
    """msae_decode_bwd_acts_f32 at (A = 1024, k = 33 = the scorer's k + 1, N = 262144, d = 4096) against
    oracle.decode_bwd_acts (kernels.py:341-400 restated as one serial f32 chain per pair).  The kernel sums 64
    lane-strided partial chains and reduces them, so the bar is a summation-order bound, 1.5e-6 |g| |w| (64
    roundings of partial sums no larger than |g| |w|); an out-of-range index gives 0 and is flagged
    (kernels.py:389)."""
    from msae import _hip, ops

    A, k = 1024, 33
    W = _unit_rows(N_C5, D, dev, seed=41)
    g = torch.Generator(device=dev).manual_seed(42)
    gout = torch.randn(A, D, generator=g, device=dev)
    idx = torch.randint(0, N_C5, (A, k), generator=g, device=dev)
    idx[0, 0], idx[-1, -1] = 0, N_C5 - 1                     # both ends of the table
    acts = torch.rand(A, k, generator=g, device=dev)
    ga, _ = ops.decode_bwd(idx, acts, W, gout, True, False)
    rows = W[idx.reshape(-1)].cpu().numpy().reshape(A * k, D)     # the oracle only needs the gathered rows
    ref = oracle.decode_bwd_acts(np.arange(A * k, dtype=np.int32).reshape(A, k), gout.cpu().numpy(), rows)
    bound = 1.5e-6 * gout.norm(dim=1, keepdim=True).cpu().numpy()           # |w| = 1
    err = np.abs(ga.cpu().numpy() - ref)
    assert (err <= bound).all(), (float(err.max()), float(bound.min()))
    # out-of-range indices: zero gradient for the pair, flag raised, neighbours untouched
    lib = _hip.load()
    bad = idx.to(torch.int32).contiguous()
    bad[3, 5], bad[700, 0] = N_C5, -1
    out = torch.full((A, k), 7.0, device=dev)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = lib.msae_decode_bwd_acts_f32(_hip.ptr(bad), _hip.ptr(gout), _hip.ptr(W), A, k, N_C5, D, _hip.ptr(out),
                                      _hip.ptr(flag), _hip.stream_of(gout))
    assert rc == 0
    assert int(flag.item()) == 1 and out[3, 5].item() == 0.0 and out[700, 0].item() == 0.0
    keep = torch.ones(A, k, dtype=torch.bool, device=dev)
    keep[3, 5] = keep[700, 0] = False
    assert torch.equal(out[keep], ga[keep])
    ops.set_debug_bounds(True)
    try:
        with pytest.raises(IndexError):
            ops.decode_bwd(bad, acts, W, gout, True, False)
        ops.decode_bwd(idx, acts, W, gout, True, False)           # in range: no complaint
    finally:
        ops.set_debug_bounds(False)


@pytest.mark.parametrize("coarse", ["int8", "fp8"])
def test_attribution_scores_at_width_262144(dev, coarse):
    """(`coarse` = fp8: BASELINE configs[4] as worded -- "width=262144 SAE + attribution-patching grad x act feature scoring, fp8
    MFMA encoder path" -- the scorer's encode on the e4m3 candidate pass; the latents are the exact path's bits either way.)
    config 5's scorer end to end on an SAE of width 262144: fused encode (k + 1 latents) -> decode ->
    d(metric)/d(reconstruction) -> score[t, j] = act_j <g_t, W_dec[j]> - act_r <g_t, W_dec[r]> (the
    Attribution.batched_scores formula) against the direct definition: metric(clean) - metric(reconstruction
    with latent j zeroed, so that the (k+1)-th latent r enters) for a LINEAR metric, where the first-order score
    is exact."""
    from msae import Sae, SaeConfig, ops

    T, k = 512, 32
    ops.set_coarse_mode(coarse)
    sae = Sae(D, SaeConfig(num_latents=N_C5, k=k), device=dev)
    with torch.no_grad():
        sae.encoder.weight.copy_(_unit_rows(N_C5, D, dev, seed=51))
        sae.W_dec.copy_(_unit_rows(N_C5, D, dev, seed=52))
        sae.b_dec.copy_(torch.randn(D, device=dev) * 0.05)
    gx = torch.Generator(device=dev).manual_seed(53)
    x = torch.randn(T, D, generator=gx, device=dev).to(torch.bfloat16)
    probe = torch.randn(T, D, generator=gx, device=dev)              # metric(recon) = <probe, recon>, linear
    with torch.no_grad():
        try:
            va, ia, st = ops.encode_topk(x, sae.encoder.weight, sae.encoder.bias, sae.b_dec, sae._prepared_weights(),
                                         k + 1)
        finally:
            ops.set_coarse_mode("int8")
        assert float((st == 0).float().mean()) > 0.9      # the candidate pass of the mode did the work
        ve, ie = ops.topk(ops.pre_acts(x[:64], sae.encoder.weight, sae.encoder.bias, sae.b_dec), k + 1)
        assert torch.equal(ia[:64], ie) and torch.equal(va[:64], ve)
        dots, _ = ops.decode_bwd(ia, va, sae.W_dec, probe, True, False)
        contrib = va * dots
        scores = (contrib[:, :k] - contrib[:, k:]) * (va[:, :k] > 0)
        clean = sae.decode(va[:, :k].contiguous(), ia[:, :k].contiguous())
        m_clean = (probe * clean).sum(-1)
        for j in (0, 7, 31):
            acts, idx = va.clone(), ia.clone()
            acts[:, j] = va[:, k]                                    # latent j off: the (k+1)-th takes its slot
            idx[:, j] = ia[:, k]
            corrupted = sae.decode(acts[:, :k].contiguous(), idx[:, :k].contiguous())
            ref = m_clean - (probe * corrupted).sum(-1)
            err = (scores[:, j] - ref).abs().max().item()
            assert err <= 2e-4 * ref.abs().max().item() + 1e-5, (j, err)
    del sae


# ---- configs[3]: weight gradients at C2 ---------------------------------------------------------------------------
@pytest.mark.parametrize("k", [32, 160])
def test_decode_bwd_wdec_at_c2_vs_float64_index_add(dev, k):
    """msae_decode_bwd_wdec_f32 at (A = 8192, k = 32 and 160 = k + 4k of a Multi-TopK step, N = 131072, d = 4096):
    the whole 2 GiB gradient against a float64 index_add_ (what triton_sparse_transpose_dense_matmul computes,
    kernels.py:10-175), with a feature hit by 3000 tokens (the L > 1024 in-place sort), one hit by 500 (LDS sort),
    one hit by none, zero activations; rows with exactly one pair are bit-exact products; two calls are
    bit-identical."""
    from msae import ops

    A = 8192
    g = torch.Generator(device=dev).manual_seed(60 + k)
    idx = torch.randint(0, N_C2, (A, k), generator=g, device=dev)
    idx[:3000, 0] = 7
    idx[::16, 1] = 11                                            # 512 tokens
    idx[idx == 5] = 6                                            # feature 5: no pair at all
    acts = torch.rand(A, k, generator=g, device=dev) + 0.05
    acts[::9, 2] = 0.0                                           # pairs that carry nothing (kernels.py:277)
    gout = torch.randn(A, D, generator=g, device=dev)
    W = torch.empty(N_C2, D, device=dev)                         # only its shape is read
    _, gw = ops.decode_bwd(idx, acts, W, gout, False, True)
    _, gw2 = ops.decode_bwd(idx, acts, W, gout, False, True)
    assert torch.equal(gw, gw2), "weight gradient is not bit-reproducible"
    del gw2, W
    ref = torch.zeros(N_C2, D, dtype=torch.float64, device=dev)
    mag = torch.zeros(N_C2, dtype=torch.float64, device=dev)     # sum |act| per row: scale of the rounding bound
    step = max(1, (1 << 27) // (k * D))                          # ~1 GiB of float64 products per chunk
    for a0 in range(0, A, step):
        sl = slice(a0, min(A, a0 + step))
        src = acts[sl].reshape(-1, 1).double() * gout[sl].double().repeat_interleave(k, 0)
        ref.index_add_(0, idx[sl].reshape(-1), src)
        mag.index_add_(0, idx[sl].reshape(-1), acts[sl].reshape(-1).double())
        del src
    assert float(gw[5].abs().max()) == 0.0 and float(ref[5].abs().max()) == 0.0
    worst = 0.0
    L = torch.clamp(torch.bincount(idx.reshape(-1), minlength=N_C2).double(), min=1.0)
    for r0 in range(0, N_C2, 16384):
        sl = slice(r0, r0 + 16384)
        err = (gw[sl].double() - ref[sl]).abs().amax(dim=1)
        # an f32 chain of L terms a_i g_i: |error| <~ sqrt(L) eps sum|a_i g_i| (worst case L eps ...); |g| < 6 here
        tol = 2e-7 * mag[sl] * 6.0 * L[sl].sqrt() + 1e-12
        worst = max(worst, float((err / tol).max()))
    assert worst <= 1.0, worst
    # rows with exactly one (non-zero) pair: the row IS round(act * grad_out[a]), bit for bit
    live = acts.reshape(-1) != 0
    counts = torch.bincount(idx.reshape(-1)[live], minlength=N_C2)
    single = torch.nonzero(counts == 1).flatten()[:4096]
    pos = torch.full((N_C2,), -1, dtype=torch.long, device=dev)
    flat_live = torch.nonzero(live).flatten()
    pos[idx.reshape(-1)[flat_live]] = flat_live                  # (any pair of the row; unique for `single`)
    p = pos[single]
    expect = acts.reshape(-1)[p].unsqueeze(1) * gout[p // k]
    assert torch.equal(gw[single], expect)
    del ref, gw


# ---- configs[3]: parameter-sized passes on the full matrix ----------------------------------------------------------
def test_unit_norm_and_grad_sumsq_on_the_full_c2_matrix(dev):
    from msae import ops

    W = _unit_rows(N_C2, D, dev, seed=71) * (torch.rand(N_C2, 1, device=dev) * 4.0 + 0.25)
    eps = torch.finfo(torch.float32).eps
    ref = W / (torch.norm(W, dim=1, keepdim=True) + eps)          # sae.py:252-255
    ss_ref = float((W.double() ** 2).sum())
    ss = torch.zeros(1, device=dev)
    ops.grad_sumsq_(ss, W)
    assert abs(float(ss) - ss_ref) <= 1e-5 * ss_ref               # 5e8 squares, f32 partial sums + atomics
    ops.unit_norm_rows_(W, eps)
    torch.testing.assert_close(W, ref, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("project", [True, False])
def test_fused_clip_project_adam_on_the_full_c2_matrix(dev, project):
    """clip_grad_norm_(1.0) -> remove_gradient_parallel_to_decoder_directions -> torch.optim.Adam
    (trainer.py:390-400, sae.py:257-271) on a [131072, 4096] parameter, two steps; rows without gradient (86 %
    of the rows carry one in a real step) stay bit-identical in step 1."""
    from msae import ops

    lr = 7e-5
    W0 = _unit_rows(N_C2, D, dev, seed=81)
    Wr = torch.nn.Parameter(W0.clone())
    opt = torch.optim.Adam([Wr], lr=lr)
    W = W0.clone()
    m, v = torch.zeros_like(W), torch.zeros_like(W)
    sumsq = torch.zeros(1, device=dev)
    for step in (1, 2):
        g = torch.Generator(device=dev).manual_seed(82 + step)
        G = torch.randn(N_C2, D, generator=g, device=dev) * 1e-3
        G[torch.rand(N_C2, generator=g, device=dev) < 0.14] = 0.0
        Wr.grad = G.clone()
        torch.nn.utils.clip_grad_norm_([Wr], 1.0)                 # |G| ~ 21: the clip is active
        if project:
            along = (Wr.grad * Wr.data).sum(dim=1, keepdim=True)
            Wr.grad -= along * Wr.data
        opt.step()
        sumsq.zero_()
        ops.grad_sumsq_(sumsq, G)
        ops.adam_rows_(W, G, m, v, step, lr, total_sumsq=sumsq, project=project)
        diff = (W - Wr.data).abs()
        bad = diff > 2e-6 + 1e-5 * Wr.data.abs()
        assert bad.float().mean().item() < 1e-5, f"step {step}: {int(bad.sum())} elements differ"
        assert diff.max().item() <= 2.1 * lr
        if step == 1:
            still = (G == 0).all(dim=1)
            assert torch.equal(W[still], W0[still])
        del G, diff, bad
    for got, ref in ((m, opt.state[Wr]["exp_avg"]), (v, opt.state[Wr]["exp_avg_sq"])):
        tol = 1e-4 * ref.abs() + 1e-5 * ref.abs().max()
        assert bool(((got - ref).abs() <= tol).all())


# ---- configs[3]: one whole optimisation step at C2 against a dense torch restatement --------------------------------
def test_train_step_at_c2_matches_dense_torch_restatement(dev):
    """SaeTrainStep.step on an SAE of d = 4096, N = 131072, k = 32 with T = 8192 tokens (one GPU's batch of
    configs[3]) against the reference trainer's step order (trainer.py:347-401) restated with dense torch ops on
    the same GPU: relu(F.linear) -> top-k -> gather decode -> FVU, autograd, clip_grad_norm_(1.0), decoder-parallel
    projection, torch.optim.Adam.  The selection is taken from the HIP path (its indices are checked to BE a top-k of
    the dense latents up to the GEMM's summation order), so both sides differentiate the same graph."""
    from msae import Sae, SaeConfig
    from msae.train import SaeTrainStep

    T, k = 8192, 32
    lr = 2e-4 / (N_C2 / 2 ** 14) ** 0.5
    sae = Sae(D, SaeConfig(num_latents=N_C2, k=k), device=dev)
    with torch.no_grad():
        sae.encoder.weight.copy_(_unit_rows(N_C2, D, dev, seed=91))
        sae.W_dec.copy_(sae.encoder.weight)                      # the reference's tied initialisation (sae.py:57)
        sae.encoder.bias.copy_(torch.randn(N_C2, device=dev) * 0.01)
        sae.b_dec.copy_(torch.randn(D, device=dev) * 0.05)
    sae.set_decoder_norm_to_unit_norm()                          # trainer.py:347-349: first thing in a step
    gx = torch.Generator(device=dev).manual_seed(92)
    x = torch.randn(T, D, generator=gx, device=dev) + 0.25 * torch.randn(D, generator=gx, device=dev)
    params0 = [p.detach().clone() for p in (sae.encoder.weight, sae.encoder.bias, sae.W_dec, sae.b_dec)]

    # ---- product path: gradients of one forward/backward, then (separately) the whole step
    out = sae(x)
    out.fvu.backward()
    grads = [p.grad.detach().clone() for p in (sae.encoder.weight, sae.encoder.bias, sae.W_dec, sae.b_dec)]
    idx_hip, acts_hip, fvu_hip = out.latent_indices.detach(), out.latent_acts.detach(), float(out.fvu)
    for p in sae.parameters():
        p.grad = None
    del out

    # ---- dense restatement
    We, be, Wd, bd = (torch.nn.Parameter(p.clone()) for p in params0)
    pre = torch.relu(torch.nn.functional.linear(x - bd, We, be))  # [T, N] dense, 4 GiB
    with torch.no_grad():
        kth = pre.gather(1, idx_hip).min(dim=1).values
        above = (pre > (kth * (1 + 1e-5) + 1e-6)[:, None]).sum(1)
        assert int(above.max()) <= k, "the HIP selection is not a top-k of the dense latents"
    acts = pre.gather(1, idx_hip)
    assert (acts.detach() - acts_hip).abs().max().item() <= 1e-4
    recon = torch.zeros(T, D, device=dev)
    for j0 in range(0, k, 8):                                     # gather decode, 1 GiB of rows at a time
        recon = recon + (acts[:, j0:j0 + 8, None] * Wd[idx_hip[:, j0:j0 + 8]]).sum(1)
    recon = recon + bd
    fvu = (recon - x).pow(2).sum() / (x - x.mean(0)).pow(2).sum()
    assert abs(float(fvu) - fvu_hip) <= 1e-4 * abs(fvu_hip)
    fvu.backward()
    del pre, recon, acts
    for name, got, ref in zip(("W_enc", "b_enc", "W_dec", "b_dec"), grads, (We.grad, be.grad, Wd.grad, bd.grad)):
        err = (got - ref).abs().max().item()
        assert err <= 2e-4 * ref.abs().max().item() + 1e-9, (name, err, ref.abs().max().item())
    torch.nn.utils.clip_grad_norm_([We, be, Wd, bd], 1.0)
    with torch.no_grad():
        Wd.grad -= (Wd.grad * Wd.data).sum(dim=1, keepdim=True) * Wd.data
    torch.optim.Adam([We, be, Wd, bd], lr=lr).step()
    del grads

    ts = SaeTrainStep(sae, lr=lr)
    stats = ts.step(x)
    assert abs(float(stats["fvu"]) - float(fvu)) <= 1e-4 * float(fvu)
    if ts.fuse_next_step:
        # the Adam pass has already applied the NEXT step's set_decoder_norm_to_unit_norm (trainer.py:352, sae.py:249-255):
        # compare with the restatement's decoder as the reference would hold it one kernel later
        assert ts._normed_version == sae.W_dec._version
        with torch.no_grad():
            Wd.data /= Wd.data.norm(dim=1, keepdim=True) + torch.finfo(torch.float32).eps
    for name, p, ref, p0 in zip(("W_enc", "b_enc", "W_dec", "b_dec"),
                                (sae.encoder.weight, sae.encoder.bias, sae.W_dec, sae.b_dec),
                                (We, be, Wd, bd), params0):
        diff = (p.detach() - ref.data).abs()
        # Adam's first update is lr * g / (|g| + eps): ill-conditioned only where |g| ~ 1e-8 (see
        # _close_but_for_adam_sign_flips in test_gpu_parity.py)
        bad = diff > 2e-6 + 1e-5 * ref.data.abs() + 0.02 * lr
        assert bad.float().mean().item() < 1e-4, (name, int(bad.sum()), bad.numel())
        assert diff.max().item() <= 2.1 * lr, (name, diff.max().item())
        moved = (p.detach() - p0).abs().max().item()
        assert moved > 0.5 * lr, (name, "the step did not move the parameter")
    fired = torch.zeros(N_C2, dtype=torch.bool, device=dev)
    fired[idx_hip.reshape(-1)] = True
    assert torch.equal(ts.num_tokens_since_fired == 0, fired)
    del sae, ts, We, be, Wd, bd

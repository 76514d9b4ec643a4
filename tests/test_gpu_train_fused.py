"""The round-4 fusions of the training step (configs[3]; reference train/sae/sae/trainer.py:347-401, sae.py:249-271): passes
that re-read a matrix another kernel has just produced are folded into the producer.  Each fused pass must produce THE SAME
BITS as the separate passes it replaces (the squared gradient norm: the same value up to summation order).
"""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu


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


def _state(dev, N, d, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    W = torch.randn(N, d, generator=g, device=dev) / d ** 0.5
    G = torch.randn(N, d, generator=g, device=dev) * 1e-3
    G[::7] = 0.0                                           # rows without a gradient (features that did not fire)
    M = torch.randn(N, d, generator=g, device=dev) * 1e-4
    V = torch.rand(N, d, generator=g, device=dev) * 1e-7
    ss = (G.double() ** 2).sum().float().reshape(1)
    return W, G, M, V, ss


@pytest.mark.parametrize("d,project", [(1024, True), (4096, True), (4096, False), (8192, True), (1000, True), (12288, True)])
def test_adam_pass_with_the_next_steps_renorm_equals_the_two_passes(dev, d, project):
    """adam_rows_(renorm_eps=eps) == adam_rows_ then unit_norm_rows_, bit for bit (W, M, V) -- including the shapes the
    fused kernel hands back to the separate passes (d % 4 != 0, d > 8192)."""
    from msae import ops

    N = 2048
    W, G, M, V, ss = _state(dev, N, d, 3)
    W2, M2, V2 = W.clone(), M.clone(), V.clone()
    eps = torch.finfo(torch.float32).eps
    ops.adam_rows_(W, G, M, V, 5, 1e-3, total_sumsq=ss, project=project)
    ops.unit_norm_rows_(W, eps)
    ops.adam_rows_(W2, G, M2, V2, 5, 1e-3, total_sumsq=ss, project=project, renorm_eps=eps)
    assert torch.equal(W, W2) and torch.equal(M, M2) and torch.equal(V, V2)
    assert torch.allclose(W2.norm(dim=1), torch.ones(N, device=dev), atol=1e-5)


@pytest.mark.parametrize("mode,tokens", [("int8", 8192), ("int8", 64), ("bf16", 8192)])
def test_adam_pass_with_the_encoder_operand_refresh_equals_the_two_passes(dev, mode, tokens):
    """adam_rows_(refresh=buf, tokens_next=T) == adam_rows_ then msae_encoder_refresh_for(T): the updated weight and the
    WHOLE prepared buffer (header with its validity bits, statistics, int8 / bf16 operands in every layout) bit for bit;
    an encode against the fused buffer is verified and equals the exact path."""
    from msae import ops

    N, d, k = 16384, 1024, 32
    ops.set_coarse_mode(mode)
    ops.set_dither("on", seed=0x5EED)      # (the operands are rounded with a hash dither: equal seeds, equal bytes)
    try:
        W, G, M, V, ss = _state(dev, N, d, 5)
        W2, M2, V2 = W.clone(), M.clone(), V.clone()
        # both buffers start zero-filled (the layout has padding nobody writes) with the OLD weights' operands
        nbytes = ops.prepare_encoder(W).numel()
        buf_a = ops.prepare_encoder(W, out=torch.zeros(nbytes, dtype=torch.uint8, device=dev))
        buf_b = ops.prepare_encoder(W, out=torch.zeros(nbytes, dtype=torch.uint8, device=dev))
        assert torch.equal(buf_a, buf_b)
        ops.adam_rows_(W, G, M, V, 2, 1e-3, total_sumsq=ss)
        ops.prepare_encoder(W, out=buf_a, active_mode_only=True, tokens_next=tokens)
        ops.adam_rows_(W2, G, M2, V2, 2, 1e-3, total_sumsq=ss, refresh=buf_b, tokens_next=tokens)
        assert torch.equal(W, W2) and torch.equal(M, M2) and torch.equal(V, V2)
        diff = (buf_a != buf_b).nonzero()
        assert diff.numel() == 0, f"prepared buffers differ at {diff.numel()} bytes, first at {int(diff[0])}"
        b = torch.zeros(N, device=dev)
        x = torch.randn(tokens, d, generator=torch.Generator(device=dev).manual_seed(9), device=dev).to(torch.bfloat16)
        v, i, st = ops.encode_topk(x, W2, b, None, buf_b, k)
        ev, ei = ops.topk(ops.pre_acts(x[:512], W2, b, None), k)
        assert float((st == 0).float().mean()) > 0.95
        assert torch.equal(i[:512], ei) and torch.equal(v[:512], ev)
        if mode == "int8":                 # a refresh with a seed of its own rounds the same weights differently
            ops.set_dither("on", seed=0x5EED + 1)
            buf_c = ops.prepare_encoder(W, out=buf_a.clone(), active_mode_only=True, tokens_next=tokens)
            assert not torch.equal(buf_c, buf_b)
            v, i, st = ops.encode_topk(x, W2, b, None, buf_c, k)
            assert torch.equal(i[:512], ei) and torch.equal(v[:512], ev)
    finally:
        ops.set_coarse_mode("int8")
        ops.set_dither("default")


def test_weight_gradient_kernel_reports_its_rows_squared_norms(dev):
    """collect_wgrad_sumsq: the weight-gradient kernel's per-row |g|^2 == the gradient it wrote, summed per row; its
    fixed-order total == grad_sumsq's total up to summation order; two runs give the same bits."""
    from msae import ops

    A, k, N, d = 4096, 32, 16384, 1024
    g = torch.Generator(device=dev).manual_seed(11)
    idx = torch.randint(0, N, (A, k), generator=g, device=dev)
    acts = torch.rand(A, k, generator=g, device=dev)
    gout = torch.randn(A, d, generator=g, device=dev)
    W = torch.zeros(N, d, device=dev)
    tot = []
    for _ in range(2):
        with ops.collect_wgrad_sumsq() as coll:
            _, gw = ops.decode_bwd(idx, acts, W, gout, False, True)
        calls, ptr, rowsq = coll[W.data_ptr()]
        assert calls == 1 and ptr == gw.data_ptr()
        ref = (gw.double() ** 2).sum(1)
        assert torch.allclose(rowsq.double(), ref, rtol=1e-5, atol=1e-12)
        acc = torch.zeros(1, device=dev)
        ops.sum_into_(acc, rowsq)
        tot.append(acc.clone())
    assert torch.equal(tot[0], tot[1])
    acc2 = torch.zeros(1, device=dev)
    ops.grad_sumsq_(acc2, gw)
    assert abs(float(tot[0]) - float(acc2)) <= 1e-5 * float(acc2)


def test_train_step_with_and_without_the_fused_passes(dev):
    """SaeTrainStep(fuse_next_step=True) against fuse_next_step=False over four steps on the same batches: same losses
    and parameters up to the summation order of the gradient norm (the clip coefficient's last bits), decoder rows
    unit-norm after every step, and the encoder operands the fused Adam pass left behind are the ones the next encode
    uses (no refresh launch: counted through the operand cache's freshness record)."""
    from msae import Sae, SaeConfig, ops
    from msae.train import SaeTrainStep

    d, N, k, T = 512, 8192, 16, 1024
    runs = []
    for fuse in (False, True):
        torch.manual_seed(7)
        sae = Sae(d, SaeConfig(num_latents=N, k=k), device=dev)
        ts = SaeTrainStep(sae, lr=1e-3, fuse_next_step=fuse)
        fvu = []
        for s in range(4):
            x = torch.randn(T, d, generator=torch.Generator(device=dev).manual_seed(100 + s), device=dev)
            fvu.append(float(ts.step(x)["fvu"]))
            if fuse:
                key = ops._train_key(sae.encoder.weight)
                assert ops._TRAIN_FRESH[key][0] == sae.encoder.weight._version     # operands of THIS version: no rebuild next step
                assert ts._normed_version == sae.W_dec._version
                assert torch.allclose(sae.W_dec.norm(dim=1), torch.ones(N, device=dev), atol=1e-5)
        runs.append(({n: p.detach().clone() for n, p in sae.named_parameters()}, fvu))
    (pa, fa), (pb, fb) = runs
    assert max(abs(a - b) for a, b in zip(fa, fb)) <= 1e-5
    lr, steps = 1e-3, 4
    assert set(pa) == {"W_dec", "b_dec", "encoder.weight", "encoder.bias"}
    for name in pa:
        a, b = pa[name], pb[name]
        if name == "W_dec":
            a = a / (a.norm(dim=1, keepdim=True) + torch.finfo(torch.float32).eps)   # the unfused run renormalises next step
        # Adam's update lr * m / (sqrt(v) + eps) is ill-conditioned where |g| ~ eps (an element whose gradient is rounding noise
        # moves by +-lr either way): a rounding-level difference upstream flips a handful of them -- bounded, and rare
        diff = (a - b).abs()
        bad = diff > 2e-6 + 1e-5 * b.abs() + 0.02 * lr * steps
        assert bad.float().mean().item() < 1e-3, (name, int(bad.sum()), bad.numel())
        assert diff.max().item() <= 2.1 * lr * steps, (name, diff.max().item())


def test_sparse_encoder_backward_bias_gradients_without_atomics(dev):
    """The sparse encoder backward (ops._SparseEncode; reference graph: nn.Linear -> relu -> topk, sae.py:172-185): the
    encoder-bias gradient comes out of the weight-gradient kernel (row_act_sum: the feature's latent gradients summed in pair
    order) and the b_dec gradient is -(s^T W_enc) as one streaming read -- against torch autograd of the dense graph, and
    bit-reproducible run to run (index_add_'s atomics were not)."""
    from msae import ops

    d, N, k, T = 512, 8192, 16, 700
    g = torch.Generator(device=dev).manual_seed(21)
    W = (torch.randn(N, d, generator=g, device=dev) / d ** 0.5).requires_grad_()
    b = (torch.randn(N, generator=g, device=dev) * 0.05).requires_grad_()
    bd = (torch.randn(d, generator=g, device=dev) * 0.1).requires_grad_()
    x = torch.randn(T, d, generator=g, device=dev)
    go = torch.randn(T, k, generator=g, device=dev)
    grads = []
    for _ in range(2):
        (acts, idx), = ops.sparse_encode(x, W, b, bd, k)
        (acts * go).sum().backward()
        grads.append([p.grad.clone() for p in (W, b, bd)])
        for p in (W, b, bd):
            p.grad = None
    for a, c in zip(*grads):
        assert torch.equal(a, c), "the sparse encoder backward is not reproducible"
    # dense restatement on the same selection
    Wr, br, bdr = (p.detach().clone().requires_grad_() for p in (W, b, bd))
    pre = torch.relu(torch.nn.functional.linear(x - bdr, Wr, br))
    (pre.gather(1, idx) * go).sum().backward()
    for got, ref, name in zip(grads[0], (Wr.grad, br.grad, bdr.grad), ("W_enc", "b_enc", "b_dec")):
        err = (got - ref).abs().max().item()
        assert err <= 2e-5 * ref.abs().max().item() + 1e-7, (name, err, ref.abs().max().item())
    # the primitive on its own
    s = torch.randn(N, generator=g, device=dev)
    s[::3] = 0.0
    out = ops.weighted_row_sum(W.detach(), s, -1.0)
    ref = -(s.double() @ W.detach().double())
    assert (out.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item() + 1e-7
    assert torch.equal(out, ops.weighted_row_sum(W.detach(), s, -1.0))

"""Heterogeneous ("trained-like" and outright hostile) encoder weights and activations for the
verification tests of the fused encoder, generated on the device with torch generators.

A trained SAE is not i.i.d. Gaussian: row norms spread over an order of magnitude, some rows carry
isolated large weights, dead latents sit at strongly negative bias, features are duplicated.  The
int8 / bf16 candidate pass has a different rounding error on every such row; the verification rule
(csrc/encode_fused.hip) must hold on all of them.  Used by tests/test_gpu_hostile.py,
tools/soak_fused.py and tools/parity_real.py.
"""
from __future__ import annotations

import torch

KINDS = ["gauss", "spiky0.2x100", "spiky1x20n", "spiky5x20", "spiky0.1x1000", "lognorm", "dead", "dup",
         "trained_like"]


def weights(kind: str, N: int, d: int, dev, seed: int = 0):
    """-> (W_enc [N, d] f32, b_enc [N] f32, b_dec [d] f32) on `dev`."""
    g = torch.Generator(device=dev).manual_seed(1000 + seed)
    W = torch.empty(N, d, device=dev)
    for r0 in range(0, N, 16384):                      # bounded temporaries at N = 262144
        r1 = min(N, r0 + 16384)
        blk = torch.randn(r1 - r0, d, generator=g, device=dev)
        W[r0:r1] = blk / blk.norm(dim=1, keepdim=True)
    b = torch.randn(N, generator=g, device=dev) * 0.02
    b_dec = torch.randn(d, generator=g, device=dev) * 0.1

    def spike(pct, mult, renorm):
        n = max(1, int(N * pct / 100))
        rows = torch.randperm(N, generator=g, device=dev)[:n]
        cols = torch.randint(0, d, (n,), generator=g, device=dev)
        sign = torch.where(torch.rand(n, generator=g, device=dev) < 0.5, -1.0, 1.0)
        W[rows, cols] = sign * mult / d ** 0.5
        if renorm:
            W[rows] = W[rows] / W[rows].norm(dim=1, keepdim=True)

    if kind == "gauss":
        pass
    elif kind.startswith("spiky"):                    # spiky<pct>x<mult>[n]: pct % of rows carry one mult-x weight
        body = kind[5:]
        pct, mult = body.rstrip("n").split("x")
        spike(float(pct), float(mult), body.endswith("n"))
    elif kind == "lognorm":                           # log-normal row norms, sigma 0.7
        W *= torch.exp(0.7 * torch.randn(N, 1, generator=g, device=dev))
    elif kind == "dead":                              # a block of near-dead rows at b_enc = -5
        W[: N // 8] *= 1e-3
        b[: N // 8] = -5.0
    elif kind == "dup":                               # exact duplicates: ties across features
        src = torch.randperm(N, generator=g, device=dev)[: N // 16]
        dst = torch.randperm(N, generator=g, device=dev)[: N // 16]
        W[dst] = W[src]
        b[dst] = b[src]
    elif kind == "trained_like":                      # everything at once, plus a shared direction
        W *= torch.exp(0.7 * torch.randn(N, 1, generator=g, device=dev))
        common = torch.randn(d, generator=g, device=dev)
        common /= common.norm()
        W += 0.3 * torch.randn(N, 1, generator=g, device=dev) * common     # correlated rows
        spike(0.5, 50.0, False)
        spike(0.1, 300.0, True)
        dead = torch.randperm(N, generator=g, device=dev)[: N // 10]
        W[dead] *= 1e-2
        b[dead] = -5.0
        src = torch.randperm(N, generator=g, device=dev)[: N // 64]
        dst = torch.randperm(N, generator=g, device=dev)[: N // 64]
        W[dst] = W[src]
        b[dst] = b[src]
        W[7] = 0.0                                    # an all-zero row
    else:
        raise ValueError(kind)
    return W.contiguous(), b.contiguous(), b_dec.contiguous()


def activations(T: int, d: int, dev, seed: int = 0, kind: str = "residual"):
    """bf16 activations shaped like a residual stream: per-dim mean, a few massive dims, heavy-tailed
    token norms, one BOS-like massive token per 512."""
    g = torch.Generator(device=dev).manual_seed(5000 + seed)
    x = torch.randn(T, d, generator=g, device=dev)
    if kind == "gauss":
        return x.to(torch.bfloat16)
    x += 0.25 * torch.randn(d, generator=g, device=dev)
    for j in range(4):
        x[:, (j * 977 + 13) % d] *= 20.0
    x *= torch.exp(0.3 * torch.randn(T, 1, generator=g, device=dev))
    x[::512] *= 30.0
    return x.to(torch.bfloat16)


def status_histogram(status: torch.Tensor) -> dict:
    st = status.reshape(-1)
    out = {"verified": int((st == 0).sum()), "exact_fallback": int((st == 1).sum()),
           "unresolved": int((st >= 2).sum())}
    return out

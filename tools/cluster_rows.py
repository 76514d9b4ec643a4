"""Rows re-scored per token on the near-duplicate-cluster case of tests/test_gpu_hostile.py (msae_options::rows_rescored):
shows that tokens go past the presorted prefix (128) and still verify."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, REPO + '/multimodal-sae_amd', REPO + '/tests'):
    sys.path.insert(0, p)
import hostile
from msae import ops
dev = torch.device('cuda:0')
d, N, T, k = 1024, 32768, 4096, 32
for cluster in (160, 230):
    W, b, bd = hostile.weights("gauss", N, d, dev, seed=11)
    g = torch.Generator(device=dev).manual_seed(77)
    base = torch.randn(d, generator=g, device=dev); base /= base.norm()
    rows = torch.randperm(N, generator=g, device=dev)[:cluster]
    W[rows] = base[None, :] * (1.0 + 1e-4 * torch.arange(cluster, device=dev, dtype=torch.float32))[:, None]
    b[rows] = 0.0
    x = (torch.randn(T, d, generator=g, device=dev) + 6.0 * base[None, :] + bd).to(torch.bfloat16)
    st = torch.zeros(T, dtype=torch.int32, device=dev)
    with ops.rescore_rows(st):
        v, i, _ = ops.encode_topk(x, W.contiguous(), b, bd, ops.prepare_encoder(W.contiguous()), k)
    ok = st > 255                      # verified tokens: rounds << 24 | first round << 12 | rows
    done, rounds = (st[ok] & 0xFFF), ((st[ok] >> 24) & 0x3F)
    print(f"cluster {cluster}: {int(ok.sum())}/{T} verified on the fused path; rows re-scored min/median/max "
          f"{int(done.min())}/{int(done.median())}/{int(done.max())}; rounds max {int(rounds.max())}; "
          f"tokens past 128 rows: {int((done > 128).sum())}")

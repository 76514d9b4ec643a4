"""Sanity / timing of the fused encode + decode on other BASELINE shapes (k=256, N=262144, odd T)."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
dev = torch.device('cuda:0')
for (T, d, N, k) in ((65536, 4096, 131072, 32), (8192, 4096, 131072, 256), (4096, 4096, 262144, 32), (2880, 4096, 131072, 32), (8192, 768, 24576, 32)):
    W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, T, d, N)
    prep = ops.prepare_encoder(W_enc)
    for _ in range(2): v, i, s = ops.encode_topk(x, W_enc, b_enc, b_dec, prep, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): v, i, s = ops.encode_topk(x, W_enc, b_enc, b_dec, prep, k)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 5 * 1e3
    for _ in range(2): r = ops.decode(i, v, W_dec, b_dec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): r = ops.decode(i, v, W_dec, b_dec)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 5 * 1e3
    # exactness spot check on 64 tokens through the exact path
    ev, ei = ops.topk(ops.pre_acts(x[:64], W_enc, b_enc, b_dec), k)
    print(f"T={T} d={d} N={N} k={k}: encode {te:.2f} ms decode {td:.2f} ms -> {T / (te + td) * 1e3:.0f} tok/s; "
          f"verified {(s == 0).float().mean().item():.4f} flagged {(s != 0).sum().item()}; exact-match(64 tok) {bool(torch.equal(i[:64], ei) and torch.equal(v[:64], ev))}")
    del W_enc, W_dec, prep

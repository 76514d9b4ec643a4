"""S = 1 decode step on a feature-sharded SAE, one rank's work emulated on one GPU: fused encode of T = 1 token against
N / G rows (the small-T weight stream), for G = 1, 2, 4, 8 -- judge item: "report the N-sharded number per rank"."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
from msae.parallel import default_k_loc
dev = torch.device('cuda:0'); d, N, k = 4096, 131072, 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, 64, d, N)
xs = x[:1].contiguous()
for G in (1, 2, 4, 8):
    nl, kl = N // G, default_k_loc(k, G)
    W, b = W_enc[:nl].contiguous(), b_enc[:nl].contiguous()
    prep = ops.prepare_encoder(W)
    for _ in range(5): v, i, s = ops.encode_topk(xs, W, b, b_dec, prep, kl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): v, i, s = ops.encode_topk(xs, W, b, b_dec, prep, kl)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 50 * 1e3
    print(f"G={G}: T=1 encode of one rank's {nl} rows, k_loc={kl}: {t:.3f} ms  verified={(s == 0).float().mean().item():.2f}")
    del W, prep

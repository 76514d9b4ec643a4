"""Token-major against feature-major first round of the re-score at mid-size batches (MSAE_FM = 0 / 1 forces the route):
where does fm_pays()'s cost model put the switch, and where does the measurement?"""
import os, sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
dev = torch.device('cuda:0'); d, N, k = 4096, 131072, 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, 8192, d, N)
prep = ops.prepare_encoder(W_enc)
for T in (1024, 2048, 2880, 4096, 6144, 8192):
    xs = x[:T].contiguous()
    row = []
    for fm in ("0", "1", None):
        if fm is None: os.environ.pop("MSAE_FM", None)
        else: os.environ["MSAE_FM"] = fm
        ops._WS_BYTES_CACHE.clear()
        for _ in range(3): ops.encode_topk(xs, W_enc, b_enc, b_dec, prep, k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): v, i, s = ops.encode_topk(xs, W_enc, b_enc, b_dec, prep, k)
        torch.cuda.synchronize(); row.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f"T={T:5d}: encode token-major {row[0]:.3f} ms  feature-major {row[1]:.3f} ms  default {row[2]:.3f} ms  (tokens per feature {1.36 * T * k / N:.2f})")

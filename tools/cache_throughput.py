"""Tokens/s of the feature-activation cache loop on one GPU: Sae.encode (fused) + Cache.add_topk (COO records, kept on the
device, flushed to the host per 256 MB) over 64 batches of 8192 tokens -- the part of FeatureCache.run that is ours
(the LLM forward that produces the hidden states is not).  Prints also the encode-only rate."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
from msae.features.cache import Cache
dev = torch.device('cuda:0'); T, d, N, k, B = 8192, 4096, 131072, 32, 64
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, T, d, N)
prep = ops.prepare_encoder(W_enc)
xs = [x.roll(i, 0) for i in range(4)]
def loop(with_cache):
    cache = Cache(shard_size=0, batch_size=32)          # 32 sequences of 256 tokens per batch
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for b in range(B):
        v, i, s = ops.encode_topk(xs[b % 4], W_enc, b_enc, b_dec, prep, k)
        if with_cache:
            cache.add_topk(v.view(32, 256, k), i.view(32, 256, k), N, b, "layers.24")
    if with_cache:
        cache.flush_pending()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    n = sum(a.numel() for a in cache.feature_activations["layers.24"]) if with_cache else 0
    return el, n
loop(True)
e0, _ = loop(False)
e1, n = loop(True)
print(f"encode only: {B * T / e0 / 1e6:.3f} M tokens/s ({e0 / B * 1e3:.2f} ms per 8192-token batch)")
print(f"encode + COO records + host flush: {B * T / e1 / 1e6:.3f} M tokens/s ({e1 / B * 1e3:.2f} ms per batch), "
      f"{n} records = {n * 28 / 1e6:.0f} MB to the host")

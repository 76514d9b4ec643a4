"""Latency of Sae.encode + decode for small token counts (steering decode steps, S=1)."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
dev = torch.device('cuda:0'); d, N, k = 4096, 131072, 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, 4096, d, N)
prep = ops.prepare_encoder(W_enc)
for T in (1, 2, 4, 8, 16, 17, 32, 33, 64, 128, 129, 192, 256, 257, 1024, 2880):
    xs = x[:T].contiguous()
    res = {}
    for name, fn in (("fused", lambda: ops.encode_topk(xs, W_enc, b_enc, b_dec, prep, k)),
                     ("exact", lambda: ops.topk(ops.pre_acts(xs, W_enc, b_enc, b_dec), k))):
        if name == "exact" and T > 256: continue
        for _ in range(3): out = fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): out = fn()
        torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / 10 * 1e3
    v, i, s = ops.encode_topk(xs, W_enc, b_enc, b_dec, prep, k)
    ev, ei = ops.topk(ops.pre_acts(xs, W_enc, b_enc, b_dec), k) if T <= 256 else (v, i)
    for _ in range(3): r = ops.decode(i, v, W_dec, b_dec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): r = ops.decode(i, v, W_dec, b_dec)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 10 * 1e3
    print(f"T={T:5d}  fused {res['fused']:.3f} ms  exact {res.get('exact', float('nan')):.3f} ms  decode {td:.3f} ms  "
          f"equal={bool(torch.equal(i, ei) and torch.equal(v, ev))} verified={(s == 0).float().mean().item():.3f}")

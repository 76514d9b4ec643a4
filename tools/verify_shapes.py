"""Every token of large / wide shapes against the exact path (the bench's side records only report `verified`): many output tiles per
persistent workgroup in both candidate passes, wide N, k = 256, mid-size batches."""
import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd'); sys.path.insert(0, '/root/repo/tests')
import hostile
from msae import ops
dev = torch.device('cuda:0')
def exact(x, W, b, bd, k, chunk):
    vs, ids = [], []
    for t0 in range(0, x.shape[0], chunk):
        pre = ops.pre_acts(x[t0:t0 + chunk], W, b, bd)
        v, i = ops.topk(pre, k); vs.append(v); ids.append(i); del pre
    return torch.cat(vs), torch.cat(ids)
cases = [(65536, 131072, 32, "trained_like"), (16384, 262144, 32, "trained_like"), (8192, 262144, 256, "gauss"), (32768, 131072, 256, "trained_like"),
         (2880, 262144, 32, "trained_like"), (12288, 65536, 32, "spiky5x20"), (65536, 32768, 32, "lognorm")]
d = 4096
for (T, N, k, kind) in cases:
    W, b, bd = hostile.weights(kind, N, d, dev, seed=51)
    prepared = ops.prepare_encoder(W)
    x = hostile.activations(T, d, dev, seed=52)
    v, i, st = ops.encode_topk(x, W, b, bd, prepared, k)
    ev, ei = exact(x, W, b, bd, k, max(256, min(2048, (1 << 30) // (N * 4))))
    wrong = (i != ei).any(-1) | (v.view(torch.int32) != ev.view(torch.int32)).any(-1)
    print(f"T={T} N={N} k={k} {kind}: wrong {int(wrong.sum())} of {T}, fast path {float((st == 0).float().mean()):.4f}", flush=True)
    assert not bool(wrong.any())
    del W, prepared, x, v, i, ev, ei
    torch.cuda.empty_cache()
print("all shapes exact")

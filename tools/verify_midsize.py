"""Every token of mid-size batches (the lanes-per-row / route switches of the re-score: 256|257, 640|641, 1456|1457, ~2900) against the exact path."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd'); sys.path.insert(0, '/root/repo/tests')
import hostile
from msae import ops
dev = torch.device('cuda:0'); d, N = 4096, 131072
W, b, bd = hostile.weights("trained_like", N, d, dev, seed=71)
prepared = ops.prepare_encoder(W)
for k in (32, 8, 64):
    for T in (200, 256, 257, 300, 512, 640, 641, 1000, 1456, 1457, 2047, 2880, 3000, 4096):
        x = hostile.activations(T, d, dev, seed=72 + T)
        v, i, st = ops.encode_topk(x, W, b, bd, prepared, k)
        wrong = 0
        for t0 in range(0, T, 2048):
            pre = ops.pre_acts(x[t0:t0 + 2048], W, b, bd)
            ev, ei = ops.topk(pre, k); del pre
            wrong += int(((i[t0:t0 + 2048] != ei).any(-1) | (v[t0:t0 + 2048].view(torch.int32) != ev.view(torch.int32)).any(-1)).sum())
        print(f"k={k} T={T}: wrong {wrong}, fast path {float((st == 0).float().mean()):.4f}", flush=True)
        assert wrong == 0
print("all mid-size batches exact")

import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd'); sys.path.insert(0, '/root/repo/tests')
import hostile
from msae import ops
dev = torch.device('cuda:0')
N, d, k, B = int(os.environ.get("DBG_N", 262144)), 4096, 32, 8192
kind = os.environ.get("DBG_KIND", "trained_like")
W, b, bd = hostile.weights(kind, N, d, dev, seed=41)
ops.set_status_detail(True)
prepared = ops.prepare_encoder(W)
tot = 0
for s in range(int(os.environ.get("DBG_BATCHES", 6))):
    x = hostile.activations(B, d, dev, seed=10_000 + s)
    v, i, st = ops.encode_topk(x, W, b, bd, prepared, k)
    wrong = torch.zeros(B, dtype=torch.bool, device=dev)
    evs, eis = [], []
    for t0 in range(0, B, 1024):
        pre = ops.pre_acts(x[t0:t0 + 1024], W, b, bd)
        ev, ei = ops.topk(pre, k)
        evs.append(ev); eis.append(ei); del pre
    ev, ei = torch.cat(evs), torch.cat(eis)
    wrong = (i != ei).any(-1) | (v.view(torch.int32) != ev.view(torch.int32)).any(-1)
    tot += int(wrong.sum())
    if wrong.any() and s < 2:
        for t in wrong.nonzero().flatten()[:4].tolist():
            miss = [f for f in ei[t].tolist() if f not in i[t].tolist()]
            extra = [f for f in i[t].tolist() if f not in ei[t].tolist()]
            pos = [ei[t].tolist().index(f) for f in miss]
            print(f"  batch {s} token {t}: status {int(st[t]) & 0xFF} missing {miss} (exact slots {pos}, values {[float(ev[t, p]) for p in pos]}) extra {extra}; v_k exact {float(ev[t, -1]):.5f} fused {float(v[t, -1]):.5f}; miss%32 {[f % 32 for f in miss]}")
print(f"N={N} kind={kind} env FM={os.environ.get('MSAE_FM')} NOSUB={os.environ.get('MSAE_NO_SUBTRACT')} DITHER={os.environ.get('MSAE_DITHER')}: wrong tokens {tot}")

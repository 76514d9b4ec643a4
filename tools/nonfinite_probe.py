"""Non-finite activations (+inf, -inf, NaN in a few tokens of a batch) through every candidate-pass mode and batch-size class:
no hang, the finite tokens bit-identical to the exact path, the non-finite tokens handled like the exact path handles them."""
import sys

import torch

sys.path[:0] = ["multimodal-sae_amd", "tests"]
import hostile
from msae import ops

dev = torch.device("cuda:0")
d, N, k = 1024, 16384, 32
W, b, bd = hostile.weights("trained_like", N, d, dev, seed=5)
prepared = ops.prepare_encoder(W)
bad_total = 0
for mode in ("int8", "bf16", "fp8", "certified"):
    ops.set_certified(mode == "certified")
    ops.set_coarse_mode("int8" if mode == "certified" else mode)
    prep = ops.prepare_encoder(W)
    for T in (4, 64, 200, 300, 2048):
        x = hostile.activations(T, d, dev, seed=T).float()
        inj = {1: float("inf"), 2: float("-inf"), 3: float("nan")}
        for t, v in inj.items():
            if t < T:
                x[t, 7 * t] = v
        if T > 100:
            x[T - 1, :] = float("nan"); x[T - 2, 5] = float("inf"); x[T - 2, 6] = float("-inf")
        for dt in (torch.float32, torch.bfloat16):
            xx = x.to(dt)
            ops.set_status_detail(True)
            try:
                v, i, st = ops.encode_topk(xx, W, b, bd, prep, k)
            finally:
                ops.set_status_detail(False)
            torch.cuda.synchronize()
            pre = ops.pre_acts(xx, W, b, bd)
            ev, ei = ops.topk(pre, k)
            fin = torch.isfinite(xx.float()).all(-1)
            same_v = (v.view(torch.int32) == ev.view(torch.int32)).all(-1)
            same_i = (i == ei).all(-1)
            ok_fin = bool((same_v & same_i)[fin].all())
            nb = ~fin
            code = (st & 0xFF)
            msg = (f"{mode:9s} T={T:5d} {str(dt)[6:]:8s} finite tokens exact: {ok_fin}   non-finite tokens {int(nb.sum())}: "
                   f"codes {code[nb].tolist()} same as exact path (v, i): {same_v[nb].tolist()} {same_i[nb].tolist()}")
            print(msg, flush=True)
            if not ok_fin:
                bad_total += 1
                print("   differing finite tokens:", (~(same_v & same_i) & fin).nonzero().flatten().tolist()[:10])
            rec = ops.decode(i, v, W.contiguous(), bd) if False else None
ops.set_certified(False); ops.set_coarse_mode("int8")
print("finite-token mismatches:", bad_total)

"""Magnitude extremes through every candidate-pass mode and batch-size class: zero tokens, 1e30 / 1e-30 / denormal scales,
entries near the f32 maximum (squares overflow), one 1e20 outlier beside ordinary dims.  Everything must equal the exact path."""
import sys

import torch

sys.path[:0] = ["multimodal-sae_amd", "tests"]
import hostile
from msae import ops

dev = torch.device("cuda:0")
d, N, k = 1024, 16384, 32
W, b, bd = hostile.weights("trained_like", N, d, dev, seed=5)
bad_total = 0
for mode in ("int8", "bf16", "fp8", "certified"):
    ops.set_certified(mode == "certified")
    ops.set_coarse_mode("int8" if mode == "certified" else mode)
    prep = ops.prepare_encoder(W)
    for T in (8, 64, 200, 300, 2048):
        x = hostile.activations(T, d, dev, seed=T).float()
        x[0] = bd                      # a = 0 exactly
        x[1] = 0.0
        x[2] *= 1e30
        x[3] *= 1e-30
        x[4] *= 1e-42
        x[5] = torch.sign(x[5]) * 3e38
        x[6, 11] = 1e20
        x[7, 13] = -3e38
        for dt in (torch.float32, torch.bfloat16):
            xx = x.to(dt)
            v, i, st = ops.encode_topk(xx, W, b, bd, prep, k, status_detail=True)
            torch.cuda.synchronize()
            pre = ops.pre_acts(xx, W, b, bd)
            ev, ei = ops.topk(pre, k)
            same = (v.view(torch.int32) == ev.view(torch.int32)).all(-1) & (i == ei).all(-1)
            code = (st & 0xFF)
            print(f"{mode:9s} T={T:5d} {str(dt)[6:]:8s} all equal: {bool(same.all())}  special tokens' codes {code[:8].tolist()} "
                  f"fast-path share of the rest {float((code[8:] == 0).float().mean()) if T > 8 else float('nan'):.3f}", flush=True)
            if not bool(same.all()):
                bad_total += 1
                print("   differing tokens:", (~same).nonzero().flatten().tolist()[:10])
ops.set_certified(False); ops.set_coarse_mode("int8")
print("mismatching runs:", bad_total)

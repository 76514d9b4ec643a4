"""Differential fuzz of the exact operators against the CPU oracle over RANDOM shapes -- odd sizes, non-multiples of every vector width, k up
to N, the three activation types: msae_pre_acts_f32, msae_topk_f32, msae_decode_f32, msae_decode_bwd_acts_f32, msae_sparsify_*, and the
fused msae_encode_topk on whatever shape the draw produced (most are outside the fused fast path: the dispatch itself is under test).
Everything bit for bit but the activation gradient (summation-order bound).  usage: fuzz_ops.py [cases] [seed]"""
import os
import random
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, REPO + "/multimodal-sae_amd", REPO + "/tests"):
    sys.path.insert(0, p)
import synth
from msae import ops
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")


def bits_equal(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        return False
    if a.dtype.kind == "f":
        bad = (a.view(np.int32) != b.view(np.int32)) & ~((a == 0) & (b == 0))
    else:
        bad = a != b
    return not bad.any()


bad = 0
counts = {}
for c in range(cases):
    d = rng.choice([rng.randint(1, 300), rng.randint(1, 300), 256, 512, 768, 1024, 1000, 1028])
    N = rng.choice([rng.randint(1, 5000), rng.randint(1, 600), 4096, 8192, 8200, 16384])
    T = rng.choice([1, 2, 3, rng.randint(1, 40), rng.randint(1, 300), 128, 129, 256, 257])
    k = rng.choice([1, 2, rng.randint(1, min(N, 300)), min(N, 32), min(N, 256), N if N <= 512 else 64])
    k = max(1, min(k, N))
    dt = rng.choice([torch.float32, torch.bfloat16, torch.float16])
    W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed=100 + c)
    x = synth.activations(T, d, seed=200 + c, bf16=(dt != torch.float32), n_outlier=min(1, d - 1))
    xt = torch.from_numpy(x).to(dev).to(dt)
    x_up = xt.float().cpu().numpy()
    tW, tb, tWd, tbd = (torch.from_numpy(a).to(dev) for a in (W_enc, b_enc, W_dec, b_dec))
    what = f"case {c}: T={T} d={d} N={N} k={k} {str(dt)[6:]}"
    ok = True
    try:
        pre_ref = oracle.pre_acts(x_up, W_enc, b_enc, b_dec)
        pre = ops.pre_acts(xt, tW, tb, tbd)
        ok &= bits_equal(pre.cpu().numpy(), pre_ref) or print(what, "pre_acts differs") is not None
        rv, ri = oracle.topk(pre_ref, k)
        v, i = ops.topk(pre, k)
        ok &= (bits_equal(v.cpu().numpy(), rv) and bits_equal(i.cpu().numpy().astype(np.int32), ri)) or print(what, "topk differs") is not None
        fv, fi, st = ops.encode_topk(xt, tW, tb, tbd, ops.prepare_encoder(tW), k)
        ok &= (bits_equal(fv.cpu().numpy(), rv) and bits_equal(fi.cpu().numpy().astype(np.int32), ri) and int((st >= 2).sum()) == 0) \
            or print(what, "fused encode differs", torch.bincount(st.flatten().clamp(0, 2), minlength=3).tolist()) is not None
        dec_ref = oracle.decode(ri, rv, W_dec, b_dec)
        dec = ops.decode(i, v, tWd, tbd)
        ok &= bits_equal(dec.cpu().numpy(), dec_ref) or print(what, "decode differs") is not None
        g = synth.normalish(300 + c, T * d).reshape(T, d).astype(np.float32)
        ga_ref = oracle.decode_bwd_acts(ri, g, W_dec)
        ga, _ = ops.decode_bwd(i, v, tWd, torch.from_numpy(g).to(dev), True, False)
        # (64 lane-strided partial chains, then a reduction: a summation-order bound, not bits -- tests/test_gpu_config_sizes.py)
        bound = 1.5e-6 * np.linalg.norm(g, axis=1, keepdims=True) * np.linalg.norm(W_dec[ri.reshape(-1)], axis=1).reshape(T, k) + 1e-30
        ok &= bool((np.abs(ga.cpu().numpy() - ga_ref) <= bound).all()) or print(what, "decode_bwd_acts beyond its bound") is not None
        if T >= 2:
            B, S = (2, T // 2) if T % 2 == 0 else (1, T)
            vv, ii = rv[:B * S].reshape(B, S, k), ri[:B * S].reshape(B, S, k)
            loc_ref, act_ref = oracle.sparsify(vv, ii, B, S, row_base=77)
            loc, act = ops.sparsify(torch.from_numpy(vv).to(dev), torch.from_numpy(ii).to(dev).long(), N, row_base=77)
            ok &= (np.array_equal(loc.cpu().numpy(), loc_ref) and bits_equal(act.cpu().numpy(), act_ref)) or print(what, "sparsify differs") is not None
        fast = int((st == 0).sum())
        counts["fused fast-path tokens"] = counts.get("fused fast-path tokens", 0) + fast
        counts["tokens"] = counts.get("tokens", 0) + T
    except Exception as e:   # an error return of the C ABI on a legal shape is a finding too
        ok = False
        print(what, "EXCEPTION", type(e).__name__, str(e)[:200])
    if not ok:
        bad += 1
print(f"{cases} cases, {bad} mismatches; {counts}")

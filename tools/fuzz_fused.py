"""Differential fuzz of the fused encoder against the exact path (msae_pre_acts_f32 + the same hook edits on the dense latents +
msae_topk_f32) over random shapes, batch sizes either side of the 256-token tile, k, hook edits (sample features 32 j + 13
included) and every candidate-pass mode (int8 dithered / round-to-nearest, bf16, fp8, certified).  usage: fuzz_fused.py [cases] [seed] [modes]
modes = comma list out of int8,bf16,fp8,certified,int8_rn (default int8,int8,bf16: round 4's mix)"""
import os, sys, random
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, REPO + "/multimodal-sae_amd", REPO + "/tests"):
    sys.path.insert(0, p)
from msae import ops
import hostile

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MODES = sys.argv[3].split(",") if len(sys.argv) > 3 else ["int8", "int8", "bf16"]
dev = torch.device("cuda:0")
bad = 0
per_mode = {}
prepared_cache = {}
for c in range(cases):
    N = rng.choice([8192, 16384, 24576, 32768, 65536])
    d = rng.choice([256, 512, 1024, 2048])
    T = rng.choice([1, 7, 16, 17, 33, 64, 65, 100, 128, 129, 160, 192, 255, 256, 257, 300, 511, 512, 513, 1000, 2049])
    k = rng.choice([1, 2, 8, 32, 64, 100, 256])
    kind = rng.choice(["gauss", "trained_like", "spiky1x20n", "lognorm", "dup"])
    coarse = rng.choice(MODES)
    key = (N, d, kind, coarse == "fp8")          # (the fp8 operands take the int8 operands' place in the buffer)
    if key not in prepared_cache:
        if len(prepared_cache) > 6: prepared_cache.clear(); torch.cuda.empty_cache()
        W, b, bd = hostile.weights(kind, N, d, dev, seed=1000 + N // 64 + d + sum(map(ord, kind)))
        if coarse == "fp8": ops.set_coarse_mode("fp8")
        prepared_cache[key] = (W, b, bd, ops.prepare_encoder(W))
        ops.set_coarse_mode("int8")
    W, b, bd, prep = prepared_cache[key]
    x = hostile.activations(T, d, dev, seed=5000 + c)
    pre = ops.pre_acts(x, W, b, bd)
    kw = {}
    e = rng.random()
    if e < 0.3:
        kw["set_feature"] = rng.choice([13 + 32 * rng.randrange(N // 32), rng.randrange(N)]); kw["set_value"] = rng.choice([0.0, 0.5, 10.0, 1e4])
    if 0.2 < e < 0.6:
        row = pre[rng.randrange(T)]
        kw["zero_feature"] = rng.choice([int(row.argmax()), 13 + 32 * int(row[13::32].argmax()), rng.randrange(N)])
    lat = pre.clone()
    if "set_feature" in kw: lat[:, kw["set_feature"]] = kw["set_value"]
    if "zero_feature" in kw: lat[:, kw["zero_feature"]] = 0.0
    ev, ei = ops.topk(lat, k)
    ops.set_coarse_mode("int8" if coarse in ("certified", "int8_rn") else coarse)
    ops.set_certified(coarse == "certified")
    ops.set_dither("off" if coarse == "int8_rn" else "default")
    exact = rng.random() < 0.05                 # msae_options::exact: every token by the in-call exact path (status 1)
    v, i, st = ops.encode_topk(x, W, b, bd, prep, k, exact=exact, **kw)
    ops.set_coarse_mode("int8"); ops.set_certified(False); ops.set_dither("default")
    pm = per_mode.setdefault(coarse, [0, 0, 0])
    pm[0] += 1; pm[1] += int((st == 0).sum()); pm[2] += T
    ok = bool(torch.equal(i, ei) and torch.equal(v, ev) and (st != 2).all() and (not exact or (st == 1).all()))
    if not ok:
        bad += 1
        print(f"MISMATCH case {c}: N={N} d={d} T={T} k={k} {kind} {coarse} {kw}: idx equal {torch.equal(i, ei)} vals equal {torch.equal(v, ev)} "
              f"rows differing {(i != ei).any(dim=1).sum().item()} status {torch.bincount(st.flatten().clamp(0, 2), minlength=3).tolist()}")
print(f"{cases} cases, {bad} mismatches; per mode (cases, tokens verified by the fast path / tokens): "
      + ", ".join(f"{m}: {v[0]}, {v[1]}/{v[2]}" for m, v in sorted(per_mode.items())))

#!/usr/bin/env python
"""Soak of the fused encoder's verification rule: >= 1 M tokens, fused vs exact path, every token.

    python tools/soak_fused.py [--tokens 1048576] [--kind trained_like] [--N 131072] [--d 4096]
                               [--k 32] [--coarse int8|bf16|certified] [--out gpurun_out/soak.json]

For every batch of 8192 fresh activations the fused msae_encode_topk result is compared bit for bit
with msae_pre_acts_f32 + msae_topk_f32 on the same tokens.  "silent" = a token reported verified
(status 0) whose top-k differs from the exact path: must be 0.  The JSON also carries the status
histogram, why tokens took the in-call exact fallback, and both paths' throughput.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for p in (REPO, REPO / "multimodal-sae_amd", REPO / "tests"):
    sys.path.insert(0, str(p))

import torch

import hostile
from msae import ops

REASONS = (("list_overflow", 4), ("tau_le_0", 8), ("fewer_than_k_candidates", 16), ("rows_gt_r_max", 32),
           ("model_check", 64))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=1 << 20)
    ap.add_argument("--kind", default="trained_like")
    ap.add_argument("--N", type=int, default=131072)
    ap.add_argument("--d", type=int, default=4096)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--coarse", default="int8")
    ap.add_argument("--z", type=float, default=7.0)
    ap.add_argument("--reprepare", type=int, default=16,
                    help="re-prepare the encoder operands every this many batches (a fresh seed for the weights' rounding and, "
                         "round 6, for the shared dither vectors of large batches); 0 = once")
    ap.add_argument("--out", default=str(REPO / "gpurun_out" / "soak.json"))
    ap.add_argument("--sae_path", default=None, help="real checkpoint dir (cfg.json + sae.safetensors) instead of --kind")
    ap.add_argument("--acts", default=None, help="safetensors file with [T, d] activations instead of synthetic ones")
    a = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    if a.coarse == "certified":          # msae_options::certified (two planes per operand, deterministic band)
        ops.set_certified(True)
    else:
        ops.set_coarse_mode(a.coarse)
    ops.set_guard_z(a.z)
    ops.set_status_detail(True)
    if a.sae_path:
        from msae import Sae

        sae = Sae.load_from_disk(a.sae_path, device=dev)
        W, b, bd = sae.encoder.weight.data, sae.encoder.bias.data, sae.b_dec.data
        a.N, a.d, a.k, a.kind = W.shape[0], W.shape[1], sae.cfg.k, f"checkpoint {a.sae_path}"
    else:
        W, b, bd = hostile.weights(a.kind, a.N, a.d, dev, seed=41)
    acts = None
    if a.acts:
        from safetensors.torch import load_file

        t = next(iter(load_file(a.acts).values()))
        acts = t.reshape(-1, t.shape[-1])
        assert acts.shape[1] == a.d, f"activations have d = {acts.shape[1]}, the SAE {a.d}"
        a.tokens = min(a.tokens, acts.shape[0] // a.batch * a.batch) or acts.shape[0]
        a.batch = min(a.batch, acts.shape[0])
    prepared = ops.prepare_encoder(W)
    tot = {"tokens": 0, "silent_wrong": 0, "wrong_any": 0, "verified": 0, "exact_fallback": 0, "unresolved": 0}
    reasons = {n: 0 for n, _ in REASONS}
    t_fused = t_exact = 0.0
    chunk = max(256, min(2048, (1 << 30) // (a.N * 4)))
    for s in range((a.tokens + a.batch - 1) // a.batch):
        if acts is not None:
            x = acts[s * a.batch:(s + 1) * a.batch].to(dev)
            x = x if x.dtype in (torch.bfloat16, torch.float16, torch.float32) else x.float()
        else:
            x = hostile.activations(a.batch, a.d, dev, seed=10_000 + s)
        if a.reprepare and s and s % a.reprepare == 0 and a.coarse != "certified":
            prepared = ops.prepare_encoder(W, out=prepared)       # new seeds: the guarantee's randomness is sampled, not fixed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        v, i, status = ops.encode_topk(x, W, b, bd, prepared, a.k)
        torch.cuda.synchronize()
        t_fused += time.perf_counter() - t0
        t0 = time.perf_counter()
        wrong = torch.zeros(x.shape[0], dtype=torch.bool, device=dev)
        for t0_ in range(0, x.shape[0], chunk):
            pre = ops.pre_acts(x[t0_:t0_ + chunk], W, b, bd)
            ev, ei = ops.topk(pre, a.k)
            del pre
            sl = slice(t0_, t0_ + chunk)
            wrong[sl] = (i[sl] != ei).any(-1) | (v[sl].view(torch.int32) != ev.view(torch.int32)).any(-1)
        torch.cuda.synchronize()
        t_exact += time.perf_counter() - t0
        code = status & 0xFF
        tot["tokens"] += x.shape[0]
        tot["silent_wrong"] += int((wrong & (code == 0)).sum())
        tot["wrong_any"] += int(wrong.sum())
        tot["verified"] += int((code == 0).sum())
        tot["exact_fallback"] += int((code == 1).sum())
        tot["unresolved"] += int((code >= 2).sum())
        for n, bit in REASONS:
            reasons[n] += int((((status >> 8) & bit) != 0).sum())
        if s % 16 == 0:
            print(f"batch {s}: {tot}", flush=True)
    res = {"config": vars(a), **tot, "fallback_reasons": reasons,
           "fused_tokens_per_s": tot["tokens"] / t_fused, "exact_tokens_per_s": tot["tokens"] / t_exact,
           "device": torch.cuda.get_device_name(0)}
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(res, indent=1))
    print(json.dumps(res))
    assert tot["silent_wrong"] == 0 and tot["wrong_any"] == 0 and tot["unresolved"] == 0


if __name__ == "__main__":
    main()

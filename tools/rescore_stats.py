"""Rows / rounds of the re-scoring stage per token, from msae_options::rows_rescored (rounds << 24 | first-round rows << 12 | rows):

    python tools/rescore_stats.py [bench|trained_like|lognorm|...]        (K=256 in the environment: k = 256)
"""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, REPO + '/multimodal-sae_amd', REPO + '/tests'):
    sys.path.insert(0, p)
import bench, hostile
from msae import ops
dev = torch.device('cuda:0')
T, d, N, k = 8192, 4096, 131072, int(os.environ.get('K', '32'))
kinds = sys.argv[1:] or ["bench"]
for kind in kinds:
    if kind == "bench":
        W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, T, d, N)
    else:
        W_enc, b_enc, b_dec = hostile.weights(kind, N, d, dev, seed=41)
        x = hostile.activations(T, d, dev, seed=42)
    for mode in ("int8", "bf16"):
        ops.set_coarse_mode(mode)
        prep = ops.prepare_encoder(W_enc)
        buf = torch.zeros(T, dtype=torch.int32, device=dev)
        with ops.rescore_rows(buf):
            v, i, st = ops.encode_topk(x, W_enc, b_enc, b_dec, prep, k)
        torch.cuda.synchronize()
        s, st = buf.cpu(), st.cpu()
        ok = s >= (1 << 24)
        rounds, first, rows = (s[ok] >> 24) & 0x3F, (s[ok] >> 12) & 0xFFF, s[ok] & 0xFFF
        print(f"k={k} {kind}/{mode}: verified {int(ok.sum())}/{T}  rounds hist {torch.bincount(rounds).tolist()}  "
              f"mean rows {rows.float().mean():.1f} (first round {first.float().mean():.1f})  "
              f"rows p50/p99/max {int(rows.float().quantile(0.5))}/{int(rows.float().quantile(0.99))}/{int(rows.max())}  "
              f"not verified: {torch.unique(st[~ok], return_counts=True)}", flush=True)
        del prep
    ops.set_coarse_mode("int8")
    del W_enc

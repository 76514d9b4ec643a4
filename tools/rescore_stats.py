"""Rows / rounds of the re-scoring stage per token (needs a library built with -DMSAE_RESCORE_DEBUG,
which reports (rounds << 24 | first-round rows << 12 | rows) in `status`)."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
dev = torch.device('cuda:0')
T, d, N, k = 8192, 4096, 131072, 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, T, d, N)
prep = ops.prepare_encoder(W_enc)
for call in range(3):
    v, i, s = ops.encode_topk(x, W_enc, b_enc, b_dec, prep, k)
    torch.cuda.synchronize()
    s = s.cpu()
    rounds, first, rows = s >> 24, (s >> 12) & 0xFFF, s & 0xFFF
    print(f"call {call}: rounds hist {torch.bincount(rounds).tolist()}  mean rows {rows.float().mean():.1f}  "
          f"rows hist {dict(zip(*[t.tolist() for t in torch.unique(rows, return_counts=True)]))}"
          f"")

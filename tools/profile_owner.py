"""Kernel-level view of the candidate-exchange mode at G = 8 (one rank's work), for rocprofv3 --kernel-trace --stats."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
from msae.parallel import default_candidates
dev = torch.device('cuda:0'); T, d, N, k, G = 8192, 4096, 131072, 32, 8
W_full, b_full, W_dec, b_dec, x = bench.make_inputs(dev, T, d, N, rows=(0, N), dec_rows=(0, 8))
C, nl, per = default_candidates(k, G), N // G, T // G
xl = x[:per].contiguous()
recv = torch.stack([ops.shard_candidates(xl, b_full[g * nl:(g + 1) * nl], b_dec, ops.prepare_encoder(W_full[g * nl:(g + 1) * nl]),
                                         nl, k, g * nl, C) for g in range(G)]).contiguous()
prep0 = ops.prepare_encoder(W_full[:nl])
for _ in range(20):
    ops.shard_candidates(x, b_full[:nl], b_dec, prep0, nl, k, 0, C)
    ops.rescore_candidates(xl, W_full, b_full, b_dec, k, recv, C)
torch.cuda.synchronize()

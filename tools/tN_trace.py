"""Kernel-level timeline of small-batch encodes (T from argv, default 8), for rocprofv3 --kernel-trace --stats."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0'); d, N, k = 4096, 131072, 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, max(256, T), d, N)
prep = ops.prepare_encoder(W_enc)
xs = x[:T].contiguous()
for _ in range(5): v, i, s = ops.encode_topk(xs, W_enc, b_enc, b_dec, prep, k)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): v, i, s = ops.encode_topk(xs, W_enc, b_enc, b_dec, prep, k)
torch.cuda.synchronize(); print(f"T={T} encode wall {(time.perf_counter()-t0)/50*1e3:.3f} ms/step")

"""Kernel-level timeline of one T=1 encode + decode (steering decode step), for rocprofv3 --kernel-trace."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
dev = torch.device('cuda:0'); d, N, k = 4096, 131072, 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, 256, d, N)
prep = ops.prepare_encoder(W_enc)
xs = x[:1].contiguous()
for _ in range(5):
    v, i, s = ops.encode_topk(xs, W_enc, b_enc, b_dec, prep, k); r = ops.decode(i, v, W_dec, b_dec)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    v, i, s = ops.encode_topk(xs, W_enc, b_enc, b_dec, prep, k); r = ops.decode(i, v, W_dec, b_dec)
torch.cuda.synchronize(); print(f"T=1 encode+decode wall {(time.perf_counter()-t0)/50*1e3:.3f} ms/step")

"""pre_acts_f32_kernel alone: T = 2048 tokens at the C2 shape, a few launches (for rocprofv3 passes and a wall-clock rate)."""
import sys, time, torch
sys.path[:0] = [".", "multimodal-sae_amd", "tests"]
import bench
from msae import ops
dev = torch.device("cuda:0")
T, d, N = 2048, 4096, 131072
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, T, d, N)
del W_dec
out = ops.pre_acts(x, W_enc, b_enc, b_dec)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t0 = time.perf_counter()
for _ in range(n):
    out = ops.pre_acts(x, W_enc, b_enc, b_dec)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"pre_acts_f32 T={T}: {dt * 1e3:.3f} ms = {2.0 * T * d * N / dt / 1e12:.1f} TFLOP/s")

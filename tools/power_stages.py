"""Loop ONE stage of the path for a few seconds (so that the SMU's power / clock readings settle on it); run under tools/smi_sample.sh.
usage: power_stages.py encode|decode|gemv [seconds]"""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, REPO + "/multimodal-sae_amd"):
    sys.path.insert(0, p)
import bench
from msae import ops
what = sys.argv[1]; secs = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
dev = torch.device("cuda:0"); T, d, N, k = 8192, 4096, 131072, 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, T, d, N)
prep = ops.prepare_encoder(W_enc)
vals, idx = ops.encode_topk(x, W_enc, b_enc, b_dec, prep, k)[:2]
if what == "gemv": x = x[:1].contiguous()
torch.cuda.synchronize()
print("ready", flush=True)
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50):
        if what == "decode": ops.decode(idx, vals, W_dec, b_dec)
        else: ops.encode_topk(x, W_enc, b_enc, b_dec, prep, k)
    torch.cuda.synchronize(); n += 50
e1.record(); torch.cuda.synchronize()
print(f"{what}: {e0.elapsed_time(e1) / n:.3f} ms per call over {n} calls")

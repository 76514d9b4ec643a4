"""Wall time per call of the steering hook's S = 1 step (encode + decode + cast, features/steering.py:105-124) at C2: the
eager hook against the HIP-graph replay (msae/features/hooks.py:_DecodeStepGraph), each call followed by a device
synchronisation (latency: what one hook call adds to a generation step) and in a back-to-back loop (throughput: what it
adds when the host runs ahead)."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import Sae, SaeConfig
from msae.features import hooks
dev = torch.device('cuda:0'); d, N, k = 4096, 131072, 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, 64, d, N)
sae = Sae(d, SaeConfig(num_latents=N, k=k), device=dev).eval()
with torch.no_grad():
    sae.encoder.weight.copy_(W_enc); sae.encoder.bias.copy_(b_enc); sae.W_dec.copy_(W_dec); sae.b_dec.copy_(b_dec)
sae.invalidate_prepared()
hs = [x[i:i + 1].to(torch.float16).unsqueeze(0).contiguous() for i in range(32)]
for name, flag in (("eager", False), ("graph", True)):
    layer = torch.nn.Identity()
    hd = hooks.clamp_features_max(sae, 5, layer, graph_step=flag)
    with torch.no_grad():
        for h in hs[:4]: layer(h)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(8):
            for h in hs:
                layer(h); torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / (8 * len(hs)) * 1e3
        t0 = time.perf_counter()
        for rep in range(8):
            for h in hs: layer(h)
        host = (time.perf_counter() - t0) / (8 * len(hs)) * 1e3     # host time to ENQUEUE a call (the GPU runs behind)
        torch.cuda.synchronize()
        thr = (time.perf_counter() - t0) / (8 * len(hs)) * 1e3
    print(f"S=1 hook, {name:5s}: {lat:.3f} ms per call with a sync after each, {thr:.3f} ms back to back, host enqueue {host:.3f} ms per call")
    for hdl in hd: hdl.remove()

"""Throughput of one SAE optimisation step (BASELINE configs[3]) on one MI355X, C2 shape."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
from msae import Sae, SaeConfig
from msae.train import SaeTrainStep
dev = torch.device('cuda:0'); d, N, k, T = 4096, 131072, 32, 8192
sae = Sae(d, SaeConfig(num_latents=N, k=k), device=dev)
ts = SaeTrainStep(sae, auxk_alpha=0.0)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(T, d, generator=g, device=dev)
for _ in range(2): st = ts.step(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): st = ts.step(x)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
print(f"train step T={T} d={d} N={N} k={k}: {t*1e3:.1f} ms/step -> {T/t:.0f} tokens/s  fvu={st['fvu']:.4f}")

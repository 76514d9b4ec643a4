"""Does the ORDER of timed passes matter, or the instrumentation?  One process, the bench batch, alternating passes of 5 warm-up + 20 timed
steps: un-instrumented / instrumented (stage events + re-score statistics) / ...; wall clock taken INSIDE any sampler context."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
dev = torch.device('cuda:0'); d, N, k, T = 4096, 131072, 32, 8192
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, T, d, N)
xs = [x] + [bench.make_inputs(dev, T, d, 8192, seed=7919 * j)[4] for j in range(1, 4)]
prep = ops.prepare_encoder(W_enc)
rows = torch.zeros(T, dtype=torch.int32, device=dev)
def step(xx):
    v, i, s = ops.encode_topk(xx, W_enc, b_enc, b_dec, prep, k)
    return ops.decode(i, v, W_dec, b_dec)
def run(instr, steps=20, warm=5):
    for i in range(warm): step(xs[i % 4])
    torch.cuda.synchronize()
    if instr:
        prof = ops.StageProfile(steps)
        with ops.profiling(prof), ops.rescore_rows(rows):
            t0 = time.perf_counter()
            for i in range(steps): step(xs[i % 4])
            torch.cuda.synchronize(); el = time.perf_counter() - t0
        prof.close()
    else:
        t0 = time.perf_counter()
        for i in range(steps): step(xs[i % 4])
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    return el / steps * 1e3
out = []
for instr in (False, True, False, True, False, False, True, True):
    out.append(("instr" if instr else "plain", run(instr)))
print("  ".join(f"{n} {t:.3f}" for n, t in out))

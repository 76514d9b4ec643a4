"""Small-batch encode + decode replayed from a HIP graph (torch.cuda.CUDAGraph): the C ABI allocates nothing and never
synchronises, so a caller can capture a whole steering step once and replay it -- the kernels of a T = 1 step are short
enough for the launch gaps to matter."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
dev = torch.device('cuda:0'); d, N, k = 4096, 131072, 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, 512, d, N)
prep = ops.prepare_encoder(W_enc)


def step(xs):
    v, i, s = ops.encode_topk(xs, W_enc, b_enc, b_dec, prep, k)
    return v, i, s, ops.decode(i, v, W_dec, b_dec)


for T in (1, 4, 16, 64, 256):
    xs = x[:T].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step(xs)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = step(xs)
    ref = step(xs)
    g.replay(); torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(out, ref))
    xs.copy_(x[T:2 * T]); g.replay(); torch.cuda.synchronize()          # new input in the captured buffer
    same2 = all(torch.equal(a, b) for a, b in zip(out, step(x[T:2 * T].contiguous())))
    def timeit(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    t_eager, t_graph = timeit(lambda: step(xs)), timeit(g.replay)
    print(f"T={T:4d}  encode + decode: eager {t_eager:.3f} ms  graph replay {t_graph:.3f} ms  replay == eager: {same and same2}")

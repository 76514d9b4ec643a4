"""Stage clocks (msae_profile_*) of one fused encode at small token counts: prep | sample | tau | main | rescore | fallback, ms."""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
dev = torch.device('cuda:0'); d, N, k = 4096, 131072, 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, 4096, d, N)
prep = ops.prepare_encoder(W_enc)
Ts = [int(a) for a in sys.argv[1:]] or [64, 128, 129, 192, 256, 257, 512]
for T in Ts:
    xs = x[:T].contiguous()
    for _ in range(3): ops.encode_topk(xs, W_enc, b_enc, b_dec, prep, k)
    prof = ops.StageProfile(20)
    with ops.profiling(prof):
        for _ in range(20): ops.encode_topk(xs, W_enc, b_enc, b_dec, prep, k)
    torch.cuda.synchronize()
    st = prof.read().mean(0)
    print(f"T={T:4d} stages(prep,sample,tau,main,rescore,fallback) = {np.round(st, 4).tolist()}  sum {st.sum():.3f} ms")

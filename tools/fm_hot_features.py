"""Feature-major re-score with HOT features: h features that are candidates of EVERY token of the batch (dense features of a real SAE): their
pair counts and scatter cursors are same-address atomics.  Stage clocks of the encode with the route forced on / off (MSAE_FM), h = 0, 1, 4, 16, 64."""
import os, sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops
dev = torch.device('cuda:0'); T, d, N, k = 8192, 4096, 131072, int(sys.argv[1]) if len(sys.argv) > 1 else 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, T, d, N)
prep = ops.prepare_encoder(W_enc)
for h in (0, 1, 4, 16, 64):
    b = b_enc.clone()
    b[torch.arange(h, device=dev) * 1009 + 77] += 30.0          # far above every token's k-th value
    for _ in range(3): ops.encode_topk(x, W_enc, b, b_dec, prep, k)
    prof = ops.StageProfile(10)
    with ops.profiling(prof):
        for _ in range(10): v, i, s = ops.encode_topk(x, W_enc, b, b_dec, prep, k)
    torch.cuda.synchronize()
    st = prof.read().mean(0); prof.close()
    ev, ei = ops.topk(ops.pre_acts(x[:256], W_enc, b, b_dec), k)
    print(f"MSAE_FM={os.environ.get('MSAE_FM', 'default')} k={k} hot features {h:3d}: rescore stage {st[4]:.3f} ms  (encode {st.sum():.3f})  verified {(s == 0).float().mean().item():.3f}  "
          f"== exact path on 256 tokens: {bool(torch.equal(i[:256], ei) and torch.equal(v[:256], ev))}")

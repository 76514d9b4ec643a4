"""Per-rank cost of the feature-sharded encode, emulated on ONE GPU (rank 0's shard, no collectives)."""
import ctypes, sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/multimodal-sae_amd')
import bench
from msae import ops, _hip
from msae.parallel import default_k_loc
lib = _hip.load()
dev = torch.device('cuda:0'); T, d, N, k = 8192, 4096, 131072, 32
for G in (8, 4, 2, 1):
    kl = default_k_loc(k, G)
    nl = N // G
    W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, T, d, N, rows=(0, nl))
    prep = ops.prepare_encoder(W_enc)
    for _ in range(3): v, i, s = ops.encode_topk(x, W_enc, b_enc, b_dec, prep, kl)
    torch.cuda.synchronize()
    lib.msae_profile_begin(10); t0 = time.perf_counter()
    for _ in range(10): v, i, s = ops.encode_topk(x, W_enc, b_enc, b_dec, prep, kl)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10 * 1e3
    buf = (ctypes.c_float * 60)(); n = ctypes.c_int(0); lib.msae_profile_end(buf, ctypes.byref(n))
    st = np.array(buf[:]).reshape(10, 6).mean(0)
    gathered = torch.stack((v.view(torch.int32), i.to(torch.int32)), 0).repeat(G, 1, 1).contiguous()
    for _ in range(3): ops.merge_topk_gathered(gathered, T, G, kl, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ops.merge_topk_gathered(gathered, T, G, kl, k)
    torch.cuda.synchronize(); tm = (time.perf_counter() - t0) / 10 * 1e3
    xs = x[: T // G]
    vv, ii = v[: T // G, :].repeat(1, (k + kl - 1) // kl)[:, :k].contiguous(), i[: T // G].repeat(1, (k + kl - 1) // kl)[:, :k].contiguous()
    for _ in range(3): ops.decode(ii, vv, W_dec[:nl], b_dec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ops.decode(ii, vv, W_dec[:nl], b_dec)
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 10 * 1e3
    print(f"G={G} k_loc={kl}: encode {t:.3f} ms  stages(prep,sample,tau,gemm,rescore,fallback)={np.round(st,3).tolist()}  merge {tm:.3f}  decode(T/G) {td:.3f}  verified {(s==0).float().mean().item():.4f}")
    del W_enc, W_dec, prep

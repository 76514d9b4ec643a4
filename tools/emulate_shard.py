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
    prof = ops.StageProfile(10); t0 = time.perf_counter()
    with ops.profiling(prof):
        for _ in range(10): v, i, s = ops.encode_topk(x, W_enc, b_enc, b_dec, prep, kl)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10 * 1e3
    st = prof.read().mean(0); prof.close()
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
    # the unconditionally enqueued second round (round 4: no host read of the flag count) with NO flagged token: compaction of
    # the flags, the device-sized exact recompute (empty passes), the masked merge of the [T, k] round-2 pairs
    t2 = float("nan")
    if kl < k:
        flags = torch.zeros(T, dtype=torch.int32, device=dev)
        mv, mi = torch.zeros(T, k, device=dev), torch.zeros(T, k, dtype=torch.int64, device=dev)
        g2 = torch.zeros(G * 2, T, k, dtype=torch.int32, device=dev)
        def second():
            rows, n = ops.compact_flags(flags)
            v2 = torch.zeros(T, k, device=dev); i2 = torch.zeros(T, k, dtype=torch.int64, device=dev)
            ops.encode_topk_rows_(x, W_enc, b_enc, b_dec, rows, n, k, v2, i2, None)
            ops.merge_topk_gathered_masked_(g2, T, G, k, k, flags, mv, mi)
        for _ in range(3): second()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): second()
        torch.cuda.synchronize(); t2 = (time.perf_counter() - t0) / 10 * 1e3
    print(f"G={G} k_loc={kl}: encode {t:.3f} ms  stages(prep,sample,tau,gemm,rescore,fallback)={np.round(st,3).tolist()}  merge {tm:.3f}  second round (empty) {t2:.3f}  decode(T/G) {td:.3f}  verified {(s==0).float().mean().item():.4f}")
    del W_enc, W_dec, prep

# ---- mode="candidates": per-rank cost = shard_candidates over all T tokens + rescore_candidates over T/G tokens.
# The other shards' records are emulated by this shard's own records with the feature ids shifted into their ranges
# (same list sizes and error bands; the re-score then reads rows of all shards, as on the real group).
from msae.parallel import default_candidates
W_full, b_full, W_dec, b_dec, x = bench.make_inputs(dev, T, d, N, rows=(0, N), dec_rows=(0, 8))
for G in (8, 4, 2):
    C = default_candidates(k, G)
    nl = N // G
    preps = ops.prepare_encoder(W_full[:nl])
    per = T // G
    def sender():
        return ops.shard_candidates(x, b_full[:nl], b_dec, preps, nl, k, 0, C)
    for _ in range(3): recs = sender()
    torch.cuda.synchronize(); prof = ops.StageProfile(10); t0 = time.perf_counter()
    with ops.profiling(prof):
        for _ in range(10): recs = sender()
    torch.cuda.synchronize(); ts = (time.perf_counter() - t0) / 10 * 1e3
    st = prof.read().mean(0); prof.close()
    # real records of every shard for rank 0's tokens (each shard's candidate pass run here, untimed)
    allr = []
    for g in range(G):
        pg = ops.prepare_encoder(W_full[g * nl:(g + 1) * nl])
        allr.append(ops.shard_candidates(x[:per].contiguous(), b_full[g * nl:(g + 1) * nl], b_dec, pg, nl, k, g * nl, C))
        del pg
    recv = torch.stack(allr).contiguous()
    xl = x[:per].contiguous()
    for _ in range(3): v, i, s = ops.rescore_candidates(xl, W_full, b_full, b_dec, k, recv, C)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): v, i, s = ops.rescore_candidates(xl, W_full, b_full, b_dec, k, recv, C)
    torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / 10 * 1e3
    ev, ei, _ = ops.encode_topk(xl, W_full, b_full, b_dec, ops.prepare_encoder(W_full), k)
    same = bool(torch.equal(ei, i) and torch.equal(ev, v))
    print(f"G={G} candidates C={C}: shard pass {ts:.3f} ms stages(prep,sample,tau,gemm,pack)={np.round(st[:5],3).tolist()}  "
          f"owner re-score of T/G tokens {tr:.3f} ms  record {recs.shape[1]} B/token/shard  exact-recompute {(s==1).float().mean().item():.4f}  "
          f"== single GPU: {same}")
    del preps

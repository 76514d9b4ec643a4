import sys, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/multimodal-sae_amd')
import bench
from msae import ops, _hip
dev=torch.device('cuda:0'); T,d,N=8192,4096,131072
for G,kl in ((8,16),(4,24),(2,32),(1,32)):
    nl=N//G
    W_enc,b_enc,W_dec,b_dec,x=bench.make_inputs(dev,T,d,N,rows=(0,nl))
    prep=ops.prepare_encoder(W_enc)
    for _ in range(3): v,i,s=ops.encode_topk(x,W_enc,b_enc,b_dec,prep,kl)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): v,i,s=ops.encode_topk(x,W_enc,b_enc,b_dec,prep,kl)
    torch.cuda.synchronize(); t=(time.perf_counter()-t0)/10*1e3
    # merge cost on [T, G*kl]
    from msae.parallel import merge_topk
    av=v.repeat(1,G); ai=i.repeat(1,G)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): merge_topk(av,ai,32)
    torch.cuda.synchronize(); tm=(time.perf_counter()-t0)/10*1e3
    print(f"G={G} k_loc={kl}: local encode {t:.3f} ms, merge {tm:.3f} ms, verified {(s==0).float().mean().item():.4f}")
    del W_enc,W_dec,prep

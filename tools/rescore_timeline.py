"""Timeline of select_rescore_kernel for small batches (thread 0 of the first 64 tokens; s_memtime stamps = shader clocks, ~2.1-2.4 GHz).  Needs
    MSAE_DBG_NAME=libmsae_rtl.so MSAE_DBG_FLAGS="-DMSAE_RESCORE_TL -Wno-inline-asm" sh tools/build_dbg.sh
    MSAE_HIP_LIB=tools/bin/libmsae_rtl.so python tools/rescore_timeline.py [T]
stamps: 0 start | 1 list loaded | 2 list sorted | 3 first-round size known | 4 rows of round 1 read | 5 results sorted |
6 / 7 the same for round 2 | ... | 14 done"""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, REPO + '/multimodal-sae_amd'):
    sys.path.insert(0, p)
import bench
from msae import _hip, ops
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda:0'); d, N, k = 4096, 131072, 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, max(T, 256), d, N)
prep = ops.prepare_encoder(W_enc)
xs = x[:T].contiguous()
for _ in range(3): ops.encode_topk(xs, W_enc, b_enc, b_dec, prep, k)
torch.cuda.synchronize()
lib = _hip.load()
buf = (ctypes.c_ulonglong * (64 * 16))()
lib.msae_debug_rescore_timeline.restype = ctypes.c_int
assert lib.msae_debug_rescore_timeline(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(64, 16)[:min(T, 64)]
rounds, done = t[:, 15] >> 32, t[:, 15] & 0xFFFFFFFF
print(f"T={T}: rounds per token {np.bincount(rounds)[1:]} (1, 2, ...), rows re-scored median {np.median(done):.0f}")
names = ["load list", "sort list", "size of round 1", "rows of round 1", "sort results"]
tick = 1e-3  # kilo-cycles
rel = (t[:, :15] - t[:, :1]) * tick
for i, n in enumerate(names):
    print(f"  {n:18s} {np.median(rel[:, i + 1] - rel[:, i]):7.1f}  kcyc")
two = rounds >= 2
if two.any():
    print(f"  rows of round 2    {np.median(rel[two, 6] - rel[two, 5]):7.1f}  kcyc   sort {np.median(rel[two, 7] - rel[two, 6]):7.1f}  kcyc  ({two.sum()} tokens)")
print(f"  total (start -> done) median {np.median(rel[:, 14]):7.1f}  kcyc, max {rel[:, 14].max():7.1f}  kcyc; first start -> last done {(t[:, 14].max() - t[:, 0].min()) * tick:7.1f}  kcyc")

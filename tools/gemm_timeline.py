"""Per-tile timeline of the candidate-pass GEMM (workgroup 0, wave 0; s_memtime stamps).  Needs the tuning build:
    MSAE_DBG_NAME=libmsae_tl.so MSAE_DBG_FLAGS="-DMSAE_GEMM_TIMELINE -Wno-inline-asm" sh tools/build_dbg.sh
    MSAE_HIP_LIB=tools/bin/libmsae_tl.so python tools/gemm_timeline.py
stamps: 0 tile start | 1 behind the first barrier | 2 behind the second barrier | 3 k-loop done | 7 epilogue barriers passed | 4 element
loop done | 5 queue flushed | 6 back in the tile loop"""
import ctypes, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, REPO + '/multimodal-sae_amd'):
    sys.path.insert(0, p)
import bench
from msae import _hip, ops
dev = torch.device('cuda:0'); T, d, N, k = 8192, 4096, 131072, 32
W_enc, b_enc, W_dec, b_dec, x = bench.make_inputs(dev, T, d, N)
prep = ops.prepare_encoder(W_enc)
for _ in range(3): ops.encode_topk(x, W_enc, b_enc, b_dec, prep, k)
torch.cuda.synchronize()
lib = _hip.load()
buf = (ctypes.c_ulonglong * 512)()
lib.msae_debug_timeline.restype = ctypes.c_int
assert lib.msae_debug_timeline(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(64, 8)
if "--k" in sys.argv:   # build with -DMSAE_GEMM_TIMELINE=2 [-DMSAE_TLK_WAVE=w]: stamps inside k-tile 8 of each output tile
    seg = np.diff(t[2:62, :6], axis=1)
    for n, v in zip(["wait vmcnt (own DMA pieces)", "wait barrier", "issue next k-tile's DMA", "ds_read + MFMA", "to next iteration"],
                    np.median(seg, axis=0)):
        print(f"  {n:30s} {v:8.0f}")
    print(f"  k-tile period                  {np.median(t[2:62, 5] - t[2:62, 0]):8.0f}")
    sys.exit(0)
names = ["prologue->barrier0", "k-tile 0 (outlier)", "k-tiles 1..", "park + 2 barriers", "element loop", "flush", "loop back"]
order = [0, 1, 2, 3, 7, 4, 5, 6]                 # stamp 7 (epilogue barriers passed) sits between 3 and 4
seg = np.diff(t[:, order], axis=1)[2:62]       # skip the first / last tiles
nxt = (t[1:, 0] - t[:-1, 6])[2:61]
print("cycles per tile (median over tiles 2..61), s_memtime ticks = 100 MHz? -> see ratio to total")
tot = np.median(t[3:62, 0] - t[2:61, 0])
for n, v in zip(names, np.median(seg, axis=0)):
    print(f"  {n:22s} {v:9.0f}  ({v / tot:5.1%})")
print(f"  {'gap to next tile':22s} {np.median(nxt):9.0f}")
print(f"  tile period            {tot:9.0f}")

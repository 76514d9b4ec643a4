"""How wide would the error band of an fp8 (OCP e4m3) candidate pass be?  (BASELINE configs[4] names an "fp8 MFMA encoder
path"; DESIGN.md section 3 explains why the product's 2x-rate pass is int8.)

Emulation on the bench workload (d = 4096, N = 131072, k = 32), 256 tokens, torch on the GPU (a measurement aid, not product
code): operands rounded to the candidate format exactly as a candidate pass would hold them -- per-row / per-token scales, then
  int8   rint(v / s),  s = max|v| / 127      (outlier dims of x kept exact here: the product quantises them at their own scale)
  e4m3   float8_e4m3fn(v / s),  s = max|v| / 448
  e4m3 + MX   the same element type with OCP MX scales: one power-of-two scale per 32 elements along k (round 4: what the
         hardware's scaled MFMA takes; it absorbs dynamic range -- the massive-activation dims need no outlier tile -- but the
         elements keep their 3 mantissa bits)
  bf16   bfloat16(v)
-- coarse values c = q(a) . q(W_n) in float64, exact values p in float64.  Reported per format:
  * rms and max of (c - p) / sigma_model over all pairs near the top (the model's sigma: int8 as in encode_fused.hip; fp8 / bf16
    sigma^2 = 2 r |a|_4^2 |W_n|_4^2 with r the variance of one relative rounding);
  * rows per token an exact re-score would have to read: features with c + 7 sigma_emp >= v_k (sigma_emp = the EMPIRICAL rms of
    c - p of that token over all features, i.e. the narrowest band a correct model could use).
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, REPO + "/multimodal-sae_amd"):
    sys.path.insert(0, p)
import bench  # noqa: E402

dev = torch.device("cuda:0")
T, d, N, k = 256, 4096, 131072, 32
W, b_enc, _, b_dec, x = bench.make_inputs(dev, T, d, N)
a = x.float() - b_dec
out_dims = [(j * 977 + 13) % d for j in range(4)]
inl = torch.ones(d, dtype=torch.bool, device=dev)
inl[out_dims] = False


def q_int8(v, dims_exact=None):
    vv = v.clone()
    if dims_exact is not None:
        vv[:, ~dims_exact] = 0
    s = vv.abs().amax(dim=1, keepdim=True) / 127.0
    q = torch.round(vv / s) * s
    if dims_exact is not None:
        q[:, ~dims_exact] = v[:, ~dims_exact]
    return q


def q_e4m3(v):
    s = v.abs().amax(dim=1, keepdim=True) / 448.0
    return (v / s).to(torch.float8_e4m3fn).float() * s


def q_e4m3_mx(v):
    """OCP MX (what v_mfma_scale_f32_32x32x64_f8f6f4 consumes): blocks of 32 elements along k share a power-of-two scale
    X = 2^(floor(log2(max|block|)) - 8) (e4m3's emax), elements e4m3(v / X) with saturation at +-448."""
    blk = v.reshape(v.shape[0], -1, 32)
    amax = blk.abs().amax(dim=2, keepdim=True).clamp_min(1e-30)
    X = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)
    q = (blk / X).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * X
    return q.reshape(v.shape)


def q_bf16(v):
    return v.to(torch.bfloat16).float()


def matmul64(A, B):      # [T, d] x [N, d]^T in float64, in row blocks of B
    out = torch.empty(A.shape[0], B.shape[0], dtype=torch.float64, device=dev)
    for n0 in range(0, B.shape[0], 16384):
        out[:, n0:n0 + 16384] = A.double() @ B[n0:n0 + 16384].double().T
    return out


p = matmul64(a, W) + b_enc.double()
vk = torch.relu(p).topk(k, dim=1).values[:, -1:]
print(f"T={T} d={d} N={N} k={k}: exact k-th value mean {vk.mean():.3f}")
for name, qa, qw in (("int8", q_int8(a, inl), q_int8(W)), ("e4m3", q_e4m3(a), q_e4m3(W)),
                     ("e4m3 + MX per-32 scales", q_e4m3_mx(a), q_e4m3_mx(W)), ("bf16", q_bf16(a), q_bf16(W))):
    c = matmul64(qa, qw) + b_enc.double()
    err = c - p
    sig = err.pow(2).mean(dim=1, keepdim=True).sqrt()                 # empirical per-token rms over all features
    rows = ((c + 7.0 * sig) >= vk).sum(dim=1).float()
    near = (p >= vk - 14.0 * sig)                                       # pairs near the top
    print(f"{name}: rms error {sig.mean():.5f} (x{(sig / sig.new_tensor(1.0)).mean():.5f}), error / k-th value {float((sig / vk).mean()):.4f}, "
          f"max |err| / sigma_emp among near-top pairs {float((err.abs() / sig)[near].max()):.2f}, "
          f"rows per token with c + 7 sigma_emp >= v_k: mean {rows.mean():.1f}, p99 {rows.quantile(0.99):.0f}, max {rows.max():.0f}")
    del c, err

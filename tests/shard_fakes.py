"""CPU stand-ins for the device side of bench.py (test infrastructure): a runtime object with the interface of
bench.HipRuntime whose engines are the PRODUCT's msae.parallel.ShardedSae with the local kernels replaced by the oracle --
the collectives, slicing, merge and the whole of bench.main()'s control flow are the real code, over gloo."""
from __future__ import annotations

import contextlib
import time

import numpy as np
import torch
import torch.distributed as dist

import synth
from oracle import oracle


def _order_key(v: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
    b = np.where(b == 0x80000000, 0, b)
    return np.where(b & 0x80000000, (~b) & 0xFFFFFFFF, b | 0x80000000)


class _Event:
    def __init__(self):
        self.t = 0.0

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _Sampler:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def summary(self):
        return {}


class FakeRuntime:
    backend = "gloo"
    dry_run = True
    d_model, width = 64, 1024

    def __init__(self, stall_rank=None, raise_rank=None, faulty_mode="candidates"):
        self.stall_rank, self.raise_rank, self.faulty_mode = stall_rank, raise_rank, faulty_mode

    def device(self, local_rank):
        return torch.device("cpu")

    def init_process_group(self, dev):
        dist.init_process_group("gloo")

    def sync(self):
        pass

    def event(self):
        return _Event()

    def clock_sampler(self, dev):
        return _Sampler()

    def stage_profile(self, steps):
        return None

    def contexts(self, prof, rows_buf):
        return contextlib.nullcontext(), contextlib.nullcontext()

    def options(self, **kw):
        return contextlib.nullcontext()

    def make_inputs(self, dev, T, d, N, seed=0, rows=None, dec_rows=None):
        W_enc, b_enc, W_dec, b_dec = synth.sae_weights(d, N, seed=31)
        lo, hi = rows if rows else (0, N)
        dlo, dhi = dec_rows if dec_rows else (lo, hi)
        x = synth.activations(T, d, seed=100 + seed)
        t = torch.from_numpy
        return t(W_enc[lo:hi].copy()), t(b_enc[lo:hi].copy()), t(W_dec[dlo:dhi].copy()), t(b_dec.copy()), t(x)

    def single_gpu_encode(self, x, W_full, b_full, b_dec, k):
        v, i = oracle.encode_topk(x.numpy(), W_full.numpy(), b_full.numpy(), b_dec.numpy(), k)
        return torch.from_numpy(v), torch.from_numpy(i).long()

    def engine(self, W_enc_shard, b_enc_shard, W_dec, b_dec, k, rank=0, world=1, group=None, force_collectives=False,
               mode="topk", W_enc_full=None, b_enc_full=None, **kw):
        from msae.parallel import ShardedSae, default_candidates

        We, be, Wd, bd = (a.numpy() for a in (W_enc_shard, b_enc_shard, W_dec, b_dec))
        n_loc = We.shape[0]
        lo = rank * n_loc if world > 1 else 0
        hi = lo + n_loc
        faulty = mode == self.faulty_mode and world > 1

        def trouble():
            if faulty and rank == self.raise_rank:
                raise RuntimeError("injected failure")
            if faulty and rank == self.stall_rank:
                time.sleep(3600)

        def encode_fn(xt, kk, set_feature=-1, set_value=0.0, zero_feature=-1):
            trouble()
            loc = lambda f: f - lo if lo <= f < hi else -1
            v, i = oracle.encode_topk(xt.numpy(), We, be, bd, kk, set_feature=loc(set_feature), set_value=set_value,
                                      zero_feature=loc(zero_feature))
            return torch.from_numpy(v), torch.from_numpy(i).long(), torch.zeros(len(v), dtype=torch.int32)

        def decode_fn(idx, vals):
            return torch.from_numpy(oracle.decode(idx.numpy(), vals.numpy(), Wd, bd))

        extra = {}
        if mode == "candidates" and world > 1:
            C = default_candidates(k, world)
            stride = 12 * C + 8
            Wf, bf = W_enc_full.numpy(), b_enc_full.numpy()

            def cand_fn(xt):
                trouble()
                pre = oracle.pre_acts(xt.numpy(), We, be, bd)
                rec = np.zeros((len(pre), stride), dtype=np.uint8)
                for t, row in enumerate(pre):
                    key = (_order_key(row) << np.uint64(32)) | (np.uint64(0x7FFFFFFF) - np.arange(lo, hi, dtype=np.uint64))
                    order = np.argsort(key)[::-1]
                    rec[t, : 8 * C] = key[order[:C]].view(np.uint8)
                    tail = np.array([row[order[C]] if len(order) > C else 0.0, 0.0], dtype=np.float32)
                    rec[t, 12 * C:] = tail.view(np.uint8)
                return torch.from_numpy(rec)

            def rescore_fn(xl, recv, t_valid):
                G, per, _ = recv.shape
                r = recv.numpy()
                vals, idx = np.zeros((per, k), np.float32), np.zeros((per, k), np.int64)
                for t in range(t_valid):
                    keys = np.concatenate([r[g, t, : 8 * C].view(np.uint64) for g in range(G)])
                    tau = max(float(r[g, t, 12 * C: 12 * C + 4].view(np.float32)[0]) for g in range(G))
                    feats = (0x7FFFFFFF - (keys[keys != 0] & np.uint64(0xFFFFFFFF)).astype(np.int64))
                    pre = oracle.pre_acts(xl[t:t + 1].numpy(), Wf, bf, bd)[0]
                    ckey = (_order_key(pre[feats]) << np.uint64(32)) | (np.uint64(0x7FFFFFFF) - feats.astype(np.uint64))
                    top = feats[np.argsort(ckey)[::-1][:k]]
                    if len(top) < k or not pre[top[-1]] > tau:
                        v, i = oracle.encode_topk(xl[t:t + 1].numpy(), Wf, bf, bd, k)
                        vals[t], idx[t] = v[0], i[0]
                    else:
                        vals[t], idx[t] = pre[top], top
                return torch.from_numpy(vals), torch.from_numpy(idx), torch.zeros(per, dtype=torch.int32)

            extra = dict(cand_fn=cand_fn, rescore_fn=rescore_fn, n_cand=C)
        return ShardedSae(W_enc_shard, b_enc_shard, W_dec, b_dec, k, rank=rank, world=world, group=group,
                          force_collectives=force_collectives, encode_fn=encode_fn, decode_fn=decode_fn, mode=mode,
                          W_enc_full=W_enc_full, b_enc_full=b_enc_full, **extra)

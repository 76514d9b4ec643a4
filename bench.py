#!/usr/bin/env python
"""bench.py -- tokens/sec through SAE encode + TopK + decode (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--tokens T] [--k 32] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of T synthetic activations already resident in
HBM: fused Sae.encode (int8 MFMA candidate GEMM + exact f32 re-score + TopK) followed by the
k-sparse decode.  Workload = BASELINE.json configs[1]: d_model=4096, width=131072, k=32, bf16
activations shaped like a residual stream (a few x20 outlier dims), random-init unit-norm weights.

N > 1 (configs[2], the split north_star names) is the HEADLINE of a multi-GPU run: the 131072-feature
axis is sharded over the ranks (N/G rows of W_enc each); every rank runs the candidate pass over the SAME
T tokens against its shard.  Two exchange schemes are timed in the same run ("shard_modes") and the faster
bit-identical one is the headline: (a) per-shard EXACT top-k_loc pairs, one RCCL all-gather, merge with a
truncation re-check; (b) per-shard top-C candidates by upper bound, one RCCL all-to-all, exact re-score on the
token's owner against the replicated f32 W_enc (the HBM-bound re-score then runs once per token instead of
once per token and rank), all-gather of the results.  The decode is token-sharded with an all-gather of the
reconstruction.  Total work is fixed as G grows: "scaling": "strong", value = T * steps / max-over-ranks
time; the first 256 tokens are compared with a single-GPU encode.  The reference's own
multi-GPU mode -- token-sharded replicas, no data-path collective (launch/cache/cache.py:66) -- is
measured FIRST in the same run and reported under "replicas" (weak scaling); it is also the provisional
headline, so a feature-sharded leg that raises, stalls (watchdog) or is not bit-identical on some node
still leaves a measured whole-job line ("headline" says which leg the line's value comes from).

Optional real inputs (N = 1): --sae_path <dir with cfg.json + sae.safetensors> and/or
--acts <file.safetensors holding one [T, d] tensor> replace the synthetic SAE / activations.

Secondary records in the same line (N = 1, synthetic inputs; round-4 verdict items 1, 6): "k256" (the released checkpoint's k on
the same batch), "zipf" (heavy-tailed feature usage: firing frequency proportional to 1 / rank, a handful of dense features),
"exact_modes" (ms per step of the fused encode under msae_options::certified / ::exact on the bench batch) and "dither_off"
(the round-to-nearest statistical mode of ABI 3 beside the dithered default the headline runs); round 6: "t2880" / "t65536" (one
anyres image per call, eight bench batches per call), "sustained" (2000 steps: clock and power at equilibrium), and in
"roofline" the STEP's own fraction of BASELINE.md section 3's max(...) ("step_frac").

Timing protocol (round 6): the headline's K steps run on the UN-instrumented loop; the stage clocks / re-score statistics come
from a pass of the same W + K steps with the HIP events on that runs first ("ms_per_step_instrumented"; "timing_order" says so in
the line; the events cost 0.02-0.03 ms per step).  The wall clock of a timed loop is read INSIDE the clock sampler's context: until
round 6 the region enclosed the sampler's exit -- a join on a thread sitting in a sleep or an hwmon read, 0-4 ms per loop
(profiles/r06_timing_harness.txt).

`main(argv, rt)`: everything device-specific goes through a small runtime object (HipRuntime below); tests/test_bench_dryrun.py
drives the same control flow -- legs, watchdog, JSON schema -- on CPU over gloo at world 8 with injected kernels, so the first real
multi-GPU run cannot die in Python (round-4 verdict, item 3).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import contextlib
import ctypes
import json
import os
import sys
import threading
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
for p in (REPO, REPO / "multimodal-sae_amd", REPO / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import numpy as np
import torch
import torch.distributed as dist

D_MODEL, WIDTH = 4096, 131072
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_I8_TOPS = 5000.0       # dense int8 MFMA = 2x the bf16 rate (guide: i8 "~2x bf16")
SUSTAINED_I8_TOPS = 3944.0  # the guide's MEASURED int8 MFMA micro-benchmark ceiling (MI355X_MICROARCH.md, matrix cores table)
SUSTAINED_BF16_TFLOPS = 2382.0
PEAK_HBM_GBS = 8000.0
STAGES = ["prep", "sample_gemm", "threshold_topk", "main_gemm", "select_rescore", "exact_fallback"]

SHARDED_LEG_TIMEOUT_S = 240   # watchdog of the second (RCCL) leg; the headline line is printed regardless


def make_inputs(dev, T, d, N, seed=0, rows=None, dec_rows=None):
    """Synthetic SAE + activations.  `rows` = (lo, hi) slice of the feature axis of W_enc / b_enc held by
    this rank (`dec_rows` likewise for W_dec; default = the same slice).  The SAE (weights, biases) is
    the same on every rank; `seed` only selects the activation batch."""
    gw = torch.Generator(device=dev).manual_seed(1234)
    g = torch.Generator(device=dev).manual_seed(4321 + seed)
    lo, hi = rows if rows else (0, N)
    dlo, dhi = dec_rows if dec_rows else (lo, hi)
    # generate per 8192-row block so every rank draws identical values for its slice
    W_enc = torch.empty(hi - lo, d, device=dev)
    W_dec = torch.empty(dhi - dlo, d, device=dev)
    blk = 8192
    for b0 in range(0, N, blk):
        gb = torch.Generator(device=dev).manual_seed(977 * (b0 // blk) + 5)
        we = torch.randn(blk, d, generator=gb, device=dev)
        wd = torch.randn(blk, d, generator=gb, device=dev)
        s0, s1 = max(b0, lo), min(b0 + blk, hi)
        if s0 < s1:
            we = we / we.norm(dim=1, keepdim=True)
            W_enc[s0 - lo:s1 - lo] = we[s0 - b0:s1 - b0]
        s0, s1 = max(b0, dlo), min(b0 + blk, dhi)
        if s0 < s1:
            wd = wd / wd.norm(dim=1, keepdim=True)
            W_dec[s0 - dlo:s1 - dlo] = wd[s0 - b0:s1 - b0]
        del we, wd
    b_enc = (torch.randn(N, generator=gw, device=dev) * 0.02)[lo:hi].contiguous()
    b_dec = torch.randn(d, generator=gw, device=dev) * 0.1
    x = torch.randn(T, d, generator=g, device=dev) + 0.25 * torch.randn(d, generator=g, device=dev)
    for j in range(4):
        x[:, (j * 977 + 13) % d] *= 20.0
    return W_enc, b_enc, W_dec, b_dec, x.to(torch.bfloat16)


def load_real_inputs(dev, sae_path, acts_path, T):
    """--sae_path / --acts: a released checkpoint directory and/or cached activations."""
    from safetensors.torch import load_file

    out = {}
    if sae_path:
        from msae import Sae

        sae = Sae.load_from_disk(sae_path, device=dev)
        out.update(W_enc=sae.encoder.weight.data, b_enc=sae.encoder.bias.data, W_dec=sae.W_dec.data,
                   b_dec=sae.b_dec.data, k=sae.cfg.k)
    if acts_path:
        t = next(iter(load_file(acts_path).values()))
        x = t.reshape(-1, t.shape[-1])[:T].to(dev)
        out["x"] = x if x.dtype in (torch.bfloat16, torch.float16, torch.float32) else x.float()
    return out


def host_cpu_info():
    """(model string, physical cores, logical cpus) of the host (SURVEY 8d asks for both beside the CPU baseline)."""
    model, phys = None, None
    try:
        ids = set()
        cur = {}
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if ":" not in line:
                    if "physical id" in cur or "core id" in cur:
                        ids.add((cur.get("physical id"), cur.get("core id")))
                    cur = {}
                    continue
                key, val = (t.strip() for t in line.split(":", 1))
                cur[key] = val
                if key == "model name" and model is None:
                    model = val
        if cur and ("physical id" in cur or "core id" in cur):
            ids.add((cur.get("physical id"), cur.get("core id")))
        phys = len(ids) or None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    return model or "unknown", phys or logical, logical


def cpu_baseline(W_enc, b_enc, W_dec, b_dec, x, k, sample_T=256, reps=5):
    """The reference algorithm on torch-CPU operators (oracle.RefPort), timed on the host cores: threads = the PHYSICAL
    cores and half of them (hyper-threads and a second socket's NUMA hops do not help an f32 GEMM of this size), best
    of the two reported, both stated."""
    from oracle import oracle

    model, phys, logical = host_cpu_info()
    port = oracle.RefPort(W_enc.cpu(), b_enc.cpu(), W_dec.cpu(), b_dec.cpu(), k)
    xs = x[:sample_T].cpu()
    tried = {}
    prev = torch.get_num_threads()
    try:
        for n in sorted({max(1, phys), max(1, phys // 2)}, reverse=True):
            torch.set_num_threads(n)
            port.forward(xs)  # warm-up
            times = []
            for _ in range(reps if n == phys else max(2, reps // 2)):
                t0 = time.perf_counter()
                port.forward(xs)
                times.append(time.perf_counter() - t0)
            tried[n] = float(np.median(times))
    finally:
        torch.set_num_threads(prev)
    best = min(tried, key=tried.get)
    t = tried[best]
    rec = {"value": sample_T / t, "unit": "tokens/s", "cores": best, "kind": "port",
           "cpu_model": model, "physical_cores": phys, "logical_cpus": logical,
           "threads_tried": {str(n): round(sample_T / v, 1) for n, v in tried.items()},
           "sample": f"T={sample_T} tokens of the same workload, median of {reps} calls of "
                     f"RefPort.forward (F.linear+relu, topk, eager scatter+matmul decode), f32, "
                     f"{t * 1e3:.0f} ms/call with {best} threads"}
    # The same port pinned to ONE NUMA node (threads and first-touched weights on the same socket: oracle/cpu_baseline_worker.py,
    # a separate process because the mask must be set before torch starts its thread pool).  The faster of the two is the value.
    try:
        pinned = numa_pinned_baseline(W_enc.shape[1], W_enc.shape[0], k, sample_T, reps)
    except Exception as e:  # noqa: BLE001 -- the unpinned number stands
        pinned = {"error": f"{type(e).__name__}: {e}"[:200]}
    rec["numa_pinned"] = pinned
    if pinned.get("tokens_per_s", 0.0) > rec["value"]:
        rec.update(value=pinned["tokens_per_s"], cores=pinned["threads"],
                   sample=rec["sample"] + f"; best: the same port pinned to NUMA node {pinned.get('node')} "
                                          f"({pinned['threads']} threads, {pinned['ms_per_call']:.0f} ms/call)")
    return rec


def numa_pinned_baseline(d, N, k, sample_T, reps):
    """oracle/cpu_baseline_worker.py on the CPUs of NUMA node 0 with one thread per physical core of that node."""
    import glob
    import subprocess

    nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*/cpulist"))
    if not nodes:
        return {"skipped": "no NUMA topology in sysfs"}
    cpulist = open(nodes[0]).read().strip()
    from oracle.cpu_baseline_worker import parse_cpulist

    cpus = parse_cpulist(cpulist)
    # physical cores of the node: one sibling per core
    seen, phys_cpus = set(), []
    for c in cpus:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            phys_cpus.append(c)
    threads = max(1, len(phys_cpus))
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "cpu_baseline_worker.py")
    r = subprocess.run([sys.executable, worker, ",".join(str(c) for c in phys_cpus), str(threads), str(d), str(N), str(k),
                        str(sample_T), str(reps)], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": r.stderr[-300:]}
    out = json.loads(r.stdout.strip().splitlines()[-1])
    out.update(node=0, numa_nodes=len(nodes))
    return out


class ClockSampler:
    """Shader clock and package power of the benched GPU, sampled from amdgpu's hwmon files beside the timed loop (a host
    thread reading two sysfs files every ~2 ms: nothing is enqueued on the device).  The candidate GEMM runs at the
    package power limit (profiles/r03_power.txt), so the clock the loop actually got is part of the measurement."""

    def __init__(self, dev):
        import glob

        self.freq = self.power = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            want = "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            want = None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        hit = [c for c in cards if want and want in os.path.realpath(c)] or (cards if len(cards) == 1 else [])
        for c in hit[:1]:
            f = glob.glob(c + "/hwmon/hwmon*/freq1_input")
            pw = glob.glob(c + "/hwmon/hwmon*/power1_average") or glob.glob(c + "/hwmon/hwmon*/power1_input")
            self.freq, self.power = (f[0] if f else None), (pw[0] if pw else None)
        self.samples, self._stop, self._th = [], threading.Event(), None

    def _read(self, path):
        try:
            with open(path) as fh:
                return float(fh.read().strip())
        except Exception:
            return None

    def _run(self):
        while not self._stop.is_set():
            f = self._read(self.freq) if self.freq else None
            w = self._read(self.power) if self.power else None
            self.samples.append((f, w))
            time.sleep(0.002)

    def __enter__(self):
        if self.freq or self.power:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._th is not None:
            self._th.join()
        return False

    def summary(self):
        f = [a for a, _ in self.samples if a]
        w = [b for _, b in self.samples if b]
        return {"effective_sclk_mhz": (sum(f) / len(f) / 1e6) if f else None,
                "avg_power_w": (sum(w) / len(w) / 1e6) if w else None, "samples": len(self.samples),
                "source": "amdgpu hwmon freq1_input / power1_average sampled every ~2 ms beside the timed loop (whole "
                          "step: ~2/3 candidate GEMM at the package power limit, ~1/3 HBM-bound stages at the full clock)"}


def csrc_sha16() -> str:
    """Content hash of the kernel sources (the GPU box has no .git): ties a committed counter file to the tree it measured."""
    import hashlib

    h = hashlib.sha256()
    src = REPO / "multimodal-sae_amd" / "csrc"
    for f in sorted(src.glob("*")):
        if f.suffix in (".hip", ".h", ".sh"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def load_traffic(kernel: str):
    """Counter bytes per launch of the dominant kernel from the COMMITTED PMC summary (separate rocprofv3 --pmc passes
    of an earlier run of this command, tools/gpu_pmc.sh) -- not measured in this run.  The file carries the hash of the
    csrc/ tree the passes ran on (`_csrc_sha16`, tools/pmc_traffic.py); `stale` says whether the benched tree differs.
    -> (bytes or None, source, extra fields)."""
    f = REPO / "profiles" / "pmc_traffic.json"
    if f.exists():
        try:
            j = json.loads(f.read_text())
            e = j.get(kernel, {})
            extra = {k: e[k] for k in ("effective_sclk_mhz", "mfma_busy_frac", "duration_ms_in_counter_pass") if k in e}
            then, now = j.get("_csrc_sha16"), csrc_sha16()
            extra.update(csrc_sha16_at_counter_pass=then, csrc_sha16_benched=now, stale=(then != now))
            return e.get("bytes_per_launch"), "profiles/pmc_traffic.json (rocprofv3 --pmc passes of an earlier run of this command; L2->fabric bytes, Infinity-Cache hits included)", extra
        except Exception:
            return None, None, {}
    return None, None, {}


def zipf_bias(x, b_dec, N, k, dev):
    """Bias schedule of the "zipf" record: feature usage proportional to 1 / rank.  Target firing frequency
    f_r = min(0.5, c / r), sum_r f_r = k (c = 2.6 at N = 131072, k = 32: five dense features firing on half of the tokens,
    the top 1 % of the features above 2e-3, the tail below 1e-5 -- the heavy-tailed shape of a trained SAE, against
    the uniform k / N = 2.4e-4 of random unit rows).  With random unit rows the pre-activation of a token is N(b_n, s^2),
    s = |x - b_dec| / sqrt(d); a feature fires when it clears the common threshold theta s, so
    b_n = s (theta - Phi^-1(1 - f_n)), theta = Phi^-1(1 - k / N).  Ranks are spread over the feature axis by a fixed permutation."""
    a = x.float() - b_dec
    s = float(a.norm(dim=1).mean()) / a.shape[1] ** 0.5
    r = torch.arange(1, N + 1, dtype=torch.float64, device=dev)
    lo, hi = 0.0, 100.0
    for _ in range(60):                      # c with sum min(0.5, c / r) = k
        c = 0.5 * (lo + hi)
        if float(torch.clamp(c / r, max=0.5).sum()) > k:
            hi = c
        else:
            lo = c
    f = torch.clamp(c / r, max=0.5)
    theta = float(torch.special.ndtri(torch.tensor(1.0 - k / N, dtype=torch.float64)))
    b = s * (theta - torch.special.ndtri(1.0 - f))
    perm = torch.randperm(N, generator=torch.Generator(device=dev).manual_seed(2718), device=dev)
    out = torch.empty(N, dtype=torch.float32, device=dev)
    out[perm] = b.float()
    return out


class HipRuntime:
    """Everything of the bench that touches a device or a collective backend.  tests/test_bench_dryrun.py substitutes a CPU /
    gloo runtime with injected kernels to run main()'s control flow without a GPU."""

    backend = "nccl"            # == RCCL on ROCm
    dry_run = False
    d_model, width = D_MODEL, WIDTH

    def device(self, local_rank: int):
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)
        return dev

    def init_process_group(self, dev):
        dist.init_process_group(self.backend, device_id=dev)

    def sync(self):
        torch.cuda.synchronize()

    def event(self):
        return torch.cuda.Event(enable_timing=True)

    def clock_sampler(self, dev):
        return ClockSampler(dev)

    def make_inputs(self, dev, T, d, N, seed=0, rows=None, dec_rows=None):
        return make_inputs(dev, T, d, N, seed=seed, rows=rows, dec_rows=dec_rows)

    def engine(self, *a, **kw):
        from msae.parallel import ShardedSae

        eng = ShardedSae(*a, **kw)
        eng.reuse_buffers = True      # a streaming loop: the gathered reconstruction is consumed before the ring wraps
        return eng

    def stage_profile(self, steps):
        from msae import ops

        return ops.StageProfile(steps)

    def contexts(self, prof, rows_buf):
        """context managers active around the timed loop: stage events + per-token re-score statistics"""
        from msae import ops

        return ops.profiling(prof), ops.rescore_rows(rows_buf)

    def single_gpu_encode(self, x, W_full, b_full, b_dec, k):
        from msae import ops

        v, i, _ = ops.encode_topk(x, W_full, b_full, b_dec, ops.prepare_encoder(W_full), k)
        return v, i

    def options(self, **kw):
        """context manager: process defaults of the fused encoder's options for a block (exact / certified / dither)"""
        import contextlib

        from msae import ops

        @contextlib.contextmanager
        def cm():
            prev = (ops._defaults.exact, ops._defaults.dither, getattr(ops._defaults, "certified", False), ops._defaults.coarse)
            try:
                if "coarse" in kw:
                    ops.set_coarse_mode(kw["coarse"])
                if "exact" in kw:
                    ops.set_exact(kw["exact"])
                if "dither" in kw:
                    ops.set_dither(kw["dither"])
                if "certified" in kw:
                    ops.set_certified(kw["certified"])
                yield
            finally:
                ops.set_exact(prev[0])
                ops.set_coarse_mode(prev[3])
                ops.set_dither(prev[1], ops._defaults.dither_seed)
                if hasattr(ops, "set_certified"):
                    ops.set_certified(prev[2])

        return cm()


def main(argv=None, rt=None):
    rt = rt or HipRuntime()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tokens", type=int, default=8192, help="tokens per step (whole job)")
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--batches", type=int, default=4, help="distinct activation batches rotated through the timed loop")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1: skip the k256 / zipf / exact_modes / dither_off records (profiling runs)")
    ap.add_argument("--no-replicas", action="store_true",
                    help="N > 1: skip the second (token-sharded replicas, weak scaling) measurement")
    ap.add_argument("--cpu-sample", type=int, default=256)
    ap.add_argument("--sae_path", default=None, help="N = 1: checkpoint dir (cfg.json + sae.safetensors)")
    ap.add_argument("--acts", default=None, help="N = 1: safetensors file with one [T, d] activation tensor")
    ap.add_argument("--json-out", default=None, help="also write the JSON line to this file")
    args = ap.parse_args(argv)

    # stdout carries exactly ONE line, the JSON record: whatever a library prints there (RCCL's version banner at
    # communicator creation, for one) goes to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = rt.device(local_rank)
    ddp = world > 1 or "RANK" in os.environ  # launched by torch.distributed.run
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        rt.init_process_group(dev)

    T, d, N, k = args.tokens, rt.d_model, rt.width, args.k
    force = os.environ.get("MSAE_FORCE_COLLECTIVES") == "1"      # exercise the RCCL path on a 1-rank group
    sharded = ddp and (world > 1 or force)
    assert N % world == 0, "the feature axis must divide over the ranks"
    n_loc = N // world
    lo, hi = rank * n_loc, (rank + 1) * n_loc
    data = "synthetic"
    # ---- headline engine.  N = 1: the whole SAE on one GPU.  N > 1: rank g holds rows [g N/G, (g+1) N/G)
    # of W_enc / b_enc and a replicated W_dec; every rank sees the same T tokens (seed 0)
    engine_cand = None
    if sharded:
        # every rank keeps the whole f32 W_enc (2 GiB of 288 GB): the candidate-exchange mode re-scores a token's
        # candidates on the token's owner, and the check below encodes 256 tokens on one GPU
        W_full, b_full, W_dec, b_dec, x = rt.make_inputs(dev, T, d, N, seed=0, rows=(0, N), dec_rows=(0, N))
        W_enc, b_enc = W_full[lo:hi], b_full[lo:hi]
        engine = rt.engine(W_enc, b_enc, W_dec, b_dec, k, rank=rank, world=world, group=dist.group.WORLD,
                           force_collectives=force)
        if os.environ.get("MSAE_SHARD_MODE", "both") != "topk":
            engine_cand = rt.engine(W_enc, b_enc, W_dec, b_dec, k, rank=rank, world=world, group=dist.group.WORLD,
                                    force_collectives=force, mode="candidates", W_enc_full=W_full, b_enc_full=b_full)
    else:
        W_enc, b_enc, W_dec, b_dec, x = rt.make_inputs(dev, T, d, N, seed=rank)
        if args.sae_path or args.acts:
            real = load_real_inputs(dev, args.sae_path, args.acts, T)
            if "W_enc" in real:
                W_enc, b_enc, W_dec, b_dec, k = real["W_enc"], real["b_enc"], real["W_dec"], real["b_dec"], real["k"]
                N, d = W_enc.shape
                n_loc = N
            if "x" in real:
                x = real["x"]
                T = x.shape[0]
            elif x.shape[1] != d:
                _, _, _, _, x = rt.make_inputs(dev, T, d, 8192, seed=rank)
            data = "user-supplied files: " + ", ".join(f"{n}={v}" for n, v in (("sae", args.sae_path), ("acts", args.acts)) if v)
        engine = rt.engine(W_enc, b_enc, W_dec, b_dec, k)

    def more_batches(x0, seed0):
        """The timed loop streams DISTINCT activation batches (round-3 verdict: one batch fed to every step keeps its
        quantised copy, its a32 and the rows its tokens gather resident in the Infinity Cache): x0 + args.batches - 1
        more of the same distribution.  User-supplied activations are split into that many batches when they are long
        enough, else reused."""
        if args.acts:
            return [x0]
        return [x0] + [rt.make_inputs(dev, T, d, min(N, 8192), seed=seed0 + 7919 * j)[4] for j in range(1, args.batches)]

    rows_buf = torch.zeros(max(T, 65536), dtype=torch.int32, device=dev)    # (the t65536 record's tokens fit as well)
    sampler_out = {}

    def timed(eng, xin, steps, warmup, profile, gather=True):
        xs = xin if isinstance(xin, list) else [xin]
        for i in range(warmup):
            eng.forward(xs[i % len(xs)], async_gather=eng.collective, gather=gather)
        eng.synchronize()
        rt.sync()
        dec_ev = [(rt.event(), rt.event()) for _ in range(steps)]
        prof = None
        if profile:   # stage events are recorded on the launch stream during the timed region
            prof = rt.stage_profile(steps)
            eng.decode_events, eng.decode_event_i = dec_ev, 0
        if ddp:
            dist.barrier()
        rt.sync()
        sampler = rt.clock_sampler(dev)
        # (an un-instrumented loop carries neither the stage events nor the per-token statistics store)
        ctx_a, ctx_b = rt.contexts(prof, rows_buf) if profile else (contextlib.nullcontext(), contextlib.nullcontext())
        # The clock starts INSIDE the contexts and stops before they exit: the sampler's thread start and, above all, its join --
        # the thread sits in a 2-ms sleep or in an hwmon read (the SMU's power query takes milliseconds) when it is told to stop --
        # are harness time, not step time.  (Until round 6 they were inside the timed region: ~2 ms per timed loop, i.e. 0.1 ms per
        # step of a 20-step loop, 0.2 ms of a 10-step secondary record; the 2000-step record was the only clean one.)
        with sampler, ctx_a, ctx_b:
            t0 = time.perf_counter()
            for i in range(steps):
                out = eng.forward(xs[i % len(xs)], async_gather=eng.collective, gather=gather)
            eng.synchronize()
            rt.sync()
            if ddp:
                dist.barrier()
            el = time.perf_counter() - t0
        sampler_out.clear()
        sampler_out.update(sampler.summary())
        stage, dec_ms = np.zeros((0, 6)), float("nan")
        if profile and prof is not None:
            stage = prof.read().astype(np.float64)
            prof.close()
            if eng.decode_event_i:
                dec_ms = float(np.mean([a.elapsed_time(b) for a, b in dec_ev[: eng.decode_event_i]]))
        eng.decode_events = None
        if ddp:
            tmax = torch.tensor([el], device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el, out, stage, dec_ms

    def timed_clean(eng, xin, steps, warmup, gather=True):
        """Two passes of the SAME protocol (W warm-up steps, then K timed steps between barriers + synchronisations):
        first the INSTRUMENTED one (stage events, per-token re-score statistics: what the roofline fields are computed from),
        then the un-instrumented one, whose time is the headline -- an instrument must not sit inside the number it decorates
        (the events cost 0.02-0.03 ms per step; the order of the passes does not matter: tools/pass_order.py,
        profiles/r06_timing_harness.txt).  Both passes' times are reported, and the order is named in the line ("timing_order").
        -> (elapsed, out, stage, dec_ms, elapsed_instrumented)"""
        el_i, out_i, stage, dec_ms = timed(eng, xin, steps, warmup, profile=True, gather=gather)
        el, out, _, _ = timed(eng, xin, steps, warmup, profile=False, gather=gather)
        return el, out_i, stage, dec_ms, el_i

    # One JSON line is owed whatever happens in a collective: a wedged RCCL call cannot be caught, so a
    # watchdog prints what has been measured so far and ends the job.
    res = {"metric": "tokens/sec through SAE encode+TopK+decode, d=4096 width=131072", "value": None,
           "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
           "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
           "dtype": {"b": "bf16", "f": "fp8 (e4m3)"}.get(os.environ.get("MSAE_COARSE", "int8")[:1], "int8") +
                    " MFMA candidate select + f32 exact re-score/decode (outputs f32-exact)",
           "data": data,
           "dither": "off (round to nearest: statistical contract)" if os.environ.get("MSAE_DITHER", "1")[0] == "0" else
                     "on (default: stochastically rounded int8 operands, the dither subtracted again by the candidate passes -- the miss bound k exp(-z^2/2) holds for every input at round to nearest's band; seeds drawn by the library at prepare)"}
    if rt.dry_run:
        res["dry_run"] = "CPU / gloo with injected kernels: control flow only, no number in this line is a measurement"
    if ddp:
        # what the communicator itself reports (the line of a multi-GPU run must show that RCCL really spanned the ranks)
        res["collective_backend"] = dist.get_backend()
        res["rccl_world"] = dist.get_world_size()
    emitted = threading.Event()

    def emit():
        if rank == 0 and not emitted.is_set():
            emitted.set()
            line = json.dumps(res)
            print(line, file=json_out, flush=True)
            if args.json_out:
                with open(args.json_out, "w") as fh:
                    fh.write(line + "\n")

    def on_stall():
        res["error"] = f"a collective leg did not finish within {SHARDED_LEG_TIMEOUT_S} s"
        emit()
        os._exit(0)

    watchdog = None
    if sharded:
        watchdog = threading.Timer(float(os.environ.get("MSAE_BENCH_WATCHDOG_S", SHARDED_LEG_TIMEOUT_S)), on_stall)
        watchdog.daemon = True
        watchdog.start()

    FAIL_KEY, DONE_KEY = "msae_bench_failed", "msae_bench_line_printed"

    def fail(msg):
        """A leg raised on this rank.  Rank 0 prints what res holds (the last completed leg is its headline) and ends the
        job.  Any other rank tells rank 0 through the rendezvous store (its peers are parked in a collective this rank will
        never join, and a launcher that sees a rank die kills rank 0 before it can print), then waits to be ended: rank 0's
        monitor prints the line within seconds; the watchdogs are the backstop."""
        res["error"] = msg
        print("bench.py: " + msg, file=sys.stderr, flush=True)
        if rank == 0:
            emit()
            os._exit(0)
        t_end = time.time() + float(os.environ.get("MSAE_BENCH_WATCHDOG_S", SHARDED_LEG_TIMEOUT_S)) + 30.0
        try:
            store = dist.distributed_c10d._get_default_store()
            store.set(FAIL_KEY, msg)
            while time.time() < t_end and not store.check([DONE_KEY]):
                time.sleep(0.25)
        except Exception:  # noqa: BLE001 -- the store lives on rank 0: unreachable means rank 0 has ended
            pass
        os._exit(1)

    if sharded and rank == 0:
        def monitor():
            try:
                store = dist.distributed_c10d._get_default_store()
            except Exception:  # noqa: BLE001
                return
            while not emitted.is_set():
                try:
                    if store.check([FAIL_KEY]):
                        res["error"] = store.get(FAIL_KEY).decode(errors="replace")
                        emit()
                        store.set(DONE_KEY, "1")
                        time.sleep(1.0)             # (the failed rank reads the key and leaves before the store does)
                        os._exit(0)
                except Exception:  # noqa: BLE001
                    return
                time.sleep(1.0)

        threading.Thread(target=monitor, daemon=True).start()

    def stage_fields(stage, dec_ms, out, kk):
        """stage clocks + re-score statistics of one timed loop -> dict"""
        rec = {}
        if len(stage):
            rec["stage_ms"] = {n: float(v) for n, v in zip(STAGES, stage.mean(0))}
            rec["stage_ms"]["decode"] = dec_ms
        rec["fast_path_verified_frac"] = float((out["status"] == 0).float().mean().item())
        rb = rows_buf[: out["status"].shape[0]] if out["status"].shape[0] <= rows_buf.shape[0] else rows_buf
        got = rb[rb > 0]
        if got.numel():
            rec["rows_rescored_per_token"] = float((got & 0xFFF).float().mean().item())
            rec["rescore_rounds_per_token"] = float(((got >> 24) & 0x3F).float().mean().item())
            rec["rescore_feature_major"] = bool(((got >> 30) & 1).any().item())
        return rec

    def roofline_fields(stage, dec_ms, out, rows, tokens_decoded, with_traffic):
        mean = stage.mean(0)
        cm = {"b": "bf16", "f": "fp8"}.get(os.environ.get("MSAE_COARSE", "int8")[:1], "int8")
        i8 = cm == "int8"
        # the int8 main pass of a batch of more than 256 tokens runs over 31/32 of the features (the sample pass has
        # scored the rest, encode_fused.hip: main_row): count the work the launch does, not the width
        width = rows - rows // 32 if (i8 and T > 256 and not os.environ.get("MSAE_GEMM_ROWMAJOR")) else rows
        ach = 2.0 * T * d * width / (float(mean[3]) * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if cm == "bf16" else PEAK_I8_TOPS         # (dense fp8 = dense int8 = 2x bf16, the guide)
        kname = "gemm_kernel<%s,THRESH>" % cm
        sustained = {"int8": SUSTAINED_I8_TOPS, "bf16": SUSTAINED_BF16_TFLOPS, "fp8": SUSTAINED_I8_TOPS}[cm]
        traffic, traffic_src, pmc_extra = load_traffic(kname) if with_traffic else (None, None, {})
        res["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                           "frac": ach / peak,
                           "sustained_peak": sustained, "sustained_frac": ach / sustained,
                           "ops": "2*T*d*W multiply-adds counted as 2 ops each (int8 MACs on the int8 path), W = %d columns of "
                                  "rank 0's launch (N_rank %d); sustained_peak = the guide's measured MFMA micro-benchmark "
                                  "ceiling; the kernel runs at the package power limit (profiles/r03_power.txt, r04_gemm4w_power.txt)" % (width, rows),
                           "traffic": traffic, "traffic_source": traffic_src, "launch_ms": float(mean[3])}
        # The STEP's own roofline: max(T F_tok / P_mfma, (N d s_w + T (d s_x + B_dec)) / BW_hbm) over the
        # measured step -- what the whole encode + TopK + decode could cost at the peaks of the operand type the pass computes in
        s_w = 2 if cm == "bf16" else 1
        f_tok = 2.0 * d * rows + 2.0 * k * d
        b_dec_tok = k * d * 4 + k * 8 + d * 4
        t_mfma = T * f_tok / (peak * 1e12)
        t_hbm = (rows * d * s_w + T * (d * x.element_size() + b_dec_tok)) / (PEAK_HBM_GBS * 1e9)
        step_ms = res.get("ms_per_step") or float("nan")
        res["roofline"]["step_floor_ms"] = max(t_mfma, t_hbm) * 1e3
        res["roofline"]["step_frac"] = max(t_mfma, t_hbm) * 1e3 / step_ms
        res["roofline"]["step_note"] = ("BASELINE.md section 3: max(T F_tok / P_mfma, (N d s_w + T (d s_x + B_dec)) / BW_hbm) / ms_per_step; "
                                        "F_tok = 2 d N + 2 k d, B_dec = k d 4 + k 8 + d 4, P_mfma = %g TOP/s, BW_hbm = %g GB/s" % (peak, PEAK_HBM_GBS))
        if pmc_extra:   # from the same committed counter passes: GRBM_GUI_ACTIVE / 8 / the kernel's duration THERE, MFMA-busy share
            res["roofline"]["counter_pass"] = pmc_extra
        res.update(stage_fields(stage, dec_ms, out, k))
        bytes_dec = tokens_decoded * (k * d * 4 + k * 8 + d * 4)
        # ALGORITHMIC bytes / time: W_dec rows shared by tokens are served by L2 / Infinity Cache, so this can exceed
        # the HBM rate; it is not an HBM measurement (the counter bytes are in profiles/)
        res["decode_algorithmic_gbs"] = {"achieved": bytes_dec / (dec_ms * 1e-3) / 1e9, "unit": "GB/s",
                                         "bytes_per_token": k * d * 4 + k * 8 + d * 4}
        res["clock"] = dict(sampler_out)

    workload = ("BASELINE configs[1]: d_model=%d width=%d k=%d, %%s %s activations/step resident in HBM, %s" % (
        d, N, k, str(x.dtype).replace("torch.", "").replace("bfloat16", "bf16").replace("float", "f"),
        ("f32 weights of the checkpoint " + args.sae_path) if args.sae_path else "random-init unit-norm f32 weights"))

    if not sharded:
        xs = more_batches(x, rank)
        elapsed, out, stage, dec_ms, el_inst = timed_clean(engine, xs, args.steps, args.warmup)
        res.update(ms_per_step=elapsed / args.steps * 1e3, value=T * args.steps / elapsed,
                   ms_per_step_instrumented=el_inst / args.steps * 1e3,
                   timing_order="instrumented pass (W warm-up + K steps, stage events on) first, then the headline pass (W warm-up + K "
                                "steps, no instrumentation)",
                   config={"workload": workload % ("T=%d" % T), "tokens_per_step": T, "k": k,
                           "parallelism": "single GPU", "distinct_batches": len(xs)})
        if len(stage):
            roofline_fields(stage, dec_ms, out, N, T, with_traffic=True)
        # ---- secondary records (same batches; never the headline).  Each is guarded: a failure is recorded, not raised.
        if not args.no_secondary and not (args.sae_path or args.acts) and world == 1:
            sec_steps, sec_warm = max(3, min(args.steps, 10)), 2      # (each record: the instrumented pass, then the timed one -- as the headline)

            def record(name, fn):
                try:
                    res[name] = fn()
                except Exception as e:  # noqa: BLE001 -- the headline is owed whatever a secondary leg does
                    res[name] = {"error": f"{type(e).__name__}: {e}"[:300]}

            def run_k256():
                e2 = rt.engine(W_enc, b_enc, W_dec, b_dec, 256)
                el, o, st, dm, _ = timed_clean(e2, xs, sec_steps, sec_warm)
                rec = {"k": 256, "ms_per_step": el / sec_steps * 1e3, "value": T * sec_steps / el, "unit": "tokens/s",
                       "note": "the released 131k checkpoint's k (train/sae/README.md:33-45); same batches"}
                rec.update(stage_fields(st, dm, o, 256))
                return rec

            def run_zipf():
                bz = zipf_bias(xs[0], b_dec, N, k, dev)
                e3 = rt.engine(W_enc, bz, W_dec, b_dec, k)
                el, o, st, dm, _ = timed_clean(e3, xs, sec_steps, sec_warm)
                idx = o["top_indices"]
                counts = torch.bincount(idx.flatten(), minlength=N).float()
                top1 = counts.sort(descending=True).values[: max(1, N // 100)].sum() / max(1.0, float(counts.sum()))
                rec = {"ms_per_step": el / sec_steps * 1e3, "value": T * sec_steps / el, "unit": "tokens/s",
                       "share_of_latents_on_the_top_1pct_features": float(top1),
                       "features_fired_in_batch": int((counts > 0).sum()),
                       "note": "heavy-tailed feature usage: firing frequency ~ 1/rank via the encoder bias (zipf_bias), "
                               "a handful of dense features on ~half of the tokens; weights and batches as the headline"}
                rec.update(stage_fields(st, dm, o, k))
                return rec

            def run_modes():
                out_m = {}
                for mode in ("certified", "exact"):
                    try:
                        with rt.options(**{mode: True}):
                            el, o, _, _ = timed(engine, xs[:1], 2, 1, profile=False)
                        out_m[mode] = {"ms_per_step": el / 2 * 1e3,
                                       "status_0_frac": float((o["status"] == 0).float().mean().item())}
                    except Exception as e:  # noqa: BLE001
                        out_m[mode] = {"error": f"{type(e).__name__}: {e}"[:200]}
                out_m["note"] = ("whole step (encode + decode) under msae_options::certified (two int8 planes per operand, "
                                 "deterministic band) and ::exact (every token through the f32 MFMA path); the headline is "
                                 "the default: dithered int8 candidate pass")
                return out_m

            def run_dither_off():
                with rt.options(dither="off"):
                    e4 = rt.engine(W_enc, b_enc, W_dec, b_dec, k)      # (its operands are prepared under the option)
                    el, o, st, dm, _ = timed_clean(e4, xs, sec_steps, sec_warm)
                rec = {"ms_per_step": el / sec_steps * 1e3, "value": T * sec_steps / el, "unit": "tokens/s",
                       "note": "msae_options::dither = OFF: round-to-nearest int8 operands, the statistical contract of ABI 3"}
                rec.update(stage_fields(st, dm, o, k))
                return rec

            def run_fp8():
                with rt.options(coarse="fp8"):
                    e5 = rt.engine(W_enc, b_enc, W_dec, b_dec, k)      # (prepared under the mode: e4m3 operands)
                    el, o, st, dm, _ = timed_clean(e5, xs, sec_steps, sec_warm)
                rec = {"ms_per_step": el / sec_steps * 1e3, "value": T * sec_steps / el, "unit": "tokens/s",
                       "note": "MSAE_COARSE_FP8: the candidate pass on e4m3 operands (v_mfma_f32_32x32x16_fp8_fp8) -- BASELINE "
                               "configs[4]'s 'fp8 MFMA encoder path'; same exact outputs, a ~5x wider band than int8"}
                rec.update(stage_fields(st, dm, o, k))
                return rec

            def run_tokens(Tn, what):
                xs_n = [rt.make_inputs(dev, Tn, d, min(N, 8192), seed=101 + 7919 * j)[4] for j in range(len(xs))]
                el, o, st, dm, _ = timed_clean(engine, xs_n, sec_steps, sec_warm)
                rec = {"tokens_per_step": Tn, "ms_per_step": el / sec_steps * 1e3, "value": Tn * sec_steps / el, "unit": "tokens/s",
                       "note": what}
                rec.update(stage_fields(st, dm, o, k))
                return rec

            def run_sustained():
                n = 2000
                el, o, _, _ = timed(engine, xs, n, 0, profile=False)
                return {"steps": n, "ms_per_step": el / n * 1e3, "value": T * n / el, "unit": "tokens/s", "clock": dict(sampler_out),
                        "note": "the headline loop for %d steps (~%d s): clock and package power at thermal / power-management "
                                "equilibrium, sampled beside the loop" % (n, int(el + 0.5))}

            record("k256", run_k256)
            record("t2880", lambda: run_tokens(2880, "one anyres LLaVA-NeXT image per call (~2880 tokens: the reference's --batch_size 1 "
                                                     "cache run, README.md:46-56): per-call fixed cost counts here"))
            record("t65536", lambda: run_tokens(65536, "eight bench batches per call"))
            record("sustained", run_sustained)
            record("coarse_fp8", run_fp8)
            record("zipf", run_zipf)
            record("exact_modes", run_modes)
            record("dither_off", run_dither_off)
    else:
        # ---- leg 0: token-sharded replicas, the reference's own multi-GPU mode (launch/cache/cache.py:66; weak
        # scaling).  Every rank holds the WHOLE SAE and encodes its own batch; no data-path collective, so nothing in
        # it can wedge.  It is the PROVISIONAL headline: if a feature-sharded leg below raises or stalls on this node,
        # the printed line still carries a measured whole-job number and names the leg that failed.
        if not args.no_replicas:
            _, _, _, _, x_own = rt.make_inputs(dev, T, d, min(N, 8192), seed=1 + rank)        # this rank's own batches
            x_own = more_batches(x_own, 1 + rank)
            rep = rt.engine(W_full, b_full, W_dec, b_dec, k)
            el_r, out_r, stage_r, dec_r, _ = timed_clean(rep, x_own, args.steps, args.warmup)
            del rep, x_own
            par = f"dp{world}: token-sharded replicas, {T} tokens/step/GPU, no data-path collective"
            res["replicas"] = {"value": world * T * args.steps / el_r, "unit": "tokens/s",
                               "ms_per_step": el_r / args.steps * 1e3, "scaling": "weak",
                               "tokens_per_step": world * T, "parallelism": par}
            res.update(value=res["replicas"]["value"], ms_per_step=res["replicas"]["ms_per_step"], scaling="weak",
                       headline="replicas",
                       config={"workload": workload % ("%d x T=%d" % (world, T)), "tokens_per_step": world * T, "k": k,
                               "parallelism": par})
            if len(stage_r):
                roofline_fields(stage_r, dec_r, out_r, N, T, with_traffic=False)
        # ---- legs 1 and 2: the split north_star names (BASELINE configs[2]), total work fixed ("strong").  Both
        # exchange schemes are timed; the headline is the faster one whose first 256 tokens are bit-identical to a
        # single-GPU encode of the same tokens.
        shard_modes = res["shard_modes"] = {}
        xs = more_batches(x, 0)                      # the same batches on every rank
        x_last = xs[(args.steps - 1) % len(xs)]      # the batch of the step whose outputs are checked
        n_chk = min(256, T)
        chk_v, chk_i = rt.single_gpu_encode(x_last[:n_chk], W_full, b_full, b_dec, k)
        same = lambda o: bool(torch.equal(chk_i, o["top_indices"][:n_chk]) and torch.equal(chk_v, o["top_acts"][:n_chk]))
        best = None
        legs = [("per_shard_topk", engine, "per-shard exact top-%d, RCCL all-gather + merge" % engine.k_loc)]
        if engine_cand is not None:
            legs.append(("candidate_exchange", engine_cand,
                         "per-shard top-%d candidates by upper bound, RCCL all-to-all, exact re-score on the token's "
                         "owner against the replicated W_enc, all-gather of the results" % engine_cand.n_cand))
        for name, eng, desc in legs:
            try:
                el, o, st, dm, _ = timed_clean(eng, xs, args.steps, args.warmup)
            except Exception as e:
                shard_modes[name] = {"error": f"rank {rank}: {type(e).__name__}: {e}"}
                fail(f"feature-sharded leg {name} raised on rank {rank}")
            ok = same(o)
            shard_modes[name] = {"ms_per_step": el / args.steps * 1e3, "bit_identical_256": ok,
                                 "second_round_tokens": eng.second_round_tokens}
            shard_modes[name].update({"k_loc": eng.k_loc} if name == "per_shard_topk" else
                                     {"candidates_per_shard": eng.n_cand})
            if ok and (best is None or el < best[0]):
                best = (el, o, st, dm, eng, desc)
                # the headline from here on (also what the watchdog prints if the next leg stalls)
                res.update(value=T * args.steps / el, ms_per_step=el / args.steps * 1e3, scaling="strong",
                           headline="feature-sharded: " + name,
                           sharded_bit_identical_to_single_gpu_on_256_tokens=True,
                           second_round_tokens=eng.second_round_tokens,
                           config={"workload": workload % ("T=%d" % T), "tokens_per_step": T, "k": k,
                                   "parallelism": f"feature-sharded x{world} (BASELINE configs[2]): {n_loc} rows of "
                                                  f"W_enc per rank, {desc}, token-sharded decode + all-gather of the "
                                                  f"reconstruction"})
                if len(st):
                    roofline_fields(st, dm, o, n_loc, -(-T // world), with_traffic=False)
            # beside the timed step, never inside it: the same leg with the reconstruction left token-sharded (what a
            # token-sharded consumer needs: no 16 KiB/token all-gather), and each collective of the step on its own
            try:
                el_n, _, _, _ = timed(eng, xs, args.steps, min(args.warmup, 2), profile=False, gather=False)
                shard_modes[name]["ms_per_step_no_recon_gather"] = el_n / args.steps * 1e3
                shard_modes[name]["collective_ms"] = {a: round(b, 4) for a, b in
                                                      eng.time_collectives(T, d, steps=5, sync=rt.sync).items()}
            except Exception as e:  # noqa: BLE001
                shard_modes[name]["side_measurements_error"] = f"rank {rank}: {type(e).__name__}: {e}"[:300]
        if best is None:
            res["sharded_bit_identical_to_single_gpu_on_256_tokens"] = False
            res["error"] = ("no feature-sharded leg reproduced the single-GPU encode on this node" +
                            ("; headline = replicas" if "replicas" in res else ""))
            if "replicas" not in res:     # --no-replicas: report the first leg, flagged
                m = shard_modes["per_shard_topk"]
                res.update(value=T / (m["ms_per_step"] * 1e-3), ms_per_step=m["ms_per_step"], scaling="strong")

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not (args.sae_path or args.acts) and not rt.dry_run:
        res["cpu_baseline"] = cpu_baseline(W_enc, b_enc, W_dec, b_dec, x, k, args.cpu_sample)
    emit()
    if ddp:
        if watchdog is not None:
            watchdog.cancel()
        from msae.parallel import shutdown

        shutdown(engine, engine_cand)


if __name__ == "__main__":
    main()

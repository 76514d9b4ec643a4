"""NUMA-pinned leg of bench.py's `cpu_baseline` (TEST INFRASTRUCTURE: only bench.py's cpu_baseline leg runs this file).

The in-process baseline runs torch's sgemm with threads wherever the scheduler puts them and memory wherever torch first
touched it; on a two-socket host half of the 4 GiB of weights is then a socket away from the cores that read it (round-5
verdict, weak 9: 313 tokens/s = ~10 % of the host's fp32 peak).  This worker pins ITSELF to the CPUs of one NUMA node before
torch is imported (threads created later inherit the mask; first touch puts the weights on that node) and times the same
RefPort.forward on the same kind of synthetic workload (unit-norm f32 rows, bf16 activations with four x20 dims; the CPU
generator's values, not the GPU's -- the timing does not depend on them).  usage: cpu_baseline_worker.py <cpulist> <threads>
<d> <N> <k> <T> <reps>;  prints one JSON line."""
import json
import os
import sys
import time


def parse_cpulist(s: str):
    out = []
    for part in s.split(","):
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        elif part.strip():
            out.append(int(part))
    return out


def main():
    cpus, threads, d, N, k, T, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    want = parse_cpulist(cpus)
    try:
        os.sched_setaffinity(0, want)
    except OSError:
        pass
    mask = len(os.sched_getaffinity(0))      # (read now: OMP_PROC_BIND later narrows the MAIN thread to its own place)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    import numpy as np
    import torch

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != here]      # (run as a script: oracle/ itself is sys.path[0])
    sys.path.insert(0, os.path.dirname(here))
    from oracle import oracle

    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(1234)
    W_enc = torch.randn(N, d, generator=g)
    W_enc /= W_enc.norm(dim=1, keepdim=True)
    W_dec = torch.randn(N, d, generator=g)
    W_dec /= W_dec.norm(dim=1, keepdim=True)
    b_enc = torch.randn(N, generator=g) * 0.02
    b_dec = torch.randn(d, generator=g) * 0.1
    x = torch.randn(T, d, generator=g) + 0.25 * torch.randn(d, generator=g)
    for j in range(4):
        x[:, (j * 977 + 13) % d] *= 20.0
    x = x.to(torch.bfloat16)
    port = oracle.RefPort(W_enc, b_enc, W_dec, b_dec, k)
    port.forward(x)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        port.forward(x)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    print(json.dumps({"tokens_per_s": T / t, "ms_per_call": t * 1e3, "threads": threads, "cpus": cpus,
                      "affinity": mask}))


if __name__ == "__main__":
    main()

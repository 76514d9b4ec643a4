#!/usr/bin/env python
"""profiles/pmc_traffic.json (read by bench.py for roofline.traffic) from the PMC summary of tools/gpu_pmc.sh.
FETCH_SIZE / WRITE_SIZE come from separate rocprofv3 --pmc passes, in KiB; FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B).  Both count L2 -> fabric
requests, Infinity-Cache hits included: an upper bound on HBM bytes."""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(src))
KEYS = {"gemm_kernel<int8,THRESH>": ("gemm_kernel<GemmCfg<256, 256, 2, 2, 4, true, 0>, false>",),
        "gemm_kernel<bf16,THRESH>": ("gemm_kernel<GemmCfg<256, 256, 2, 2, 4, false, 0>, false>",),
        "select_rescore_kernel": ("select_rescore_kernel<1>", "select_rescore_kernel<1, false"), "decode_fwd_v4_kernel": ("decode_fwd_v4_kernel",)}
out = {}
for name, pats in KEYS.items():
    for k, v in d.items():
        if any(p in k for p in pats) and "FETCH_SIZE" in v:
            f = v["FETCH_SIZE"]["mean"] * 1024 * 2
            w = v.get("WRITE_SIZE", {"mean": 0})["mean"] * 1024
            out[name] = {"bytes_per_launch": f + w, "fetch_bytes": f, "write_bytes": w,
                         "launches_averaged": v["FETCH_SIZE"]["n"],
                         "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `bench.py --steps 2 "
                                 "--warmup 1`; KiB, FETCH_SIZE doubled (gfx950 rule); L2->fabric incl. Infinity-Cache hits"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))

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
        # the re-score of the bench batch since round 4: PHASE 1 | counting sort | fm_dot_kernel (the row reads) | PHASE 2
        "fm_dot_kernel": ("fm_dot_kernel",), "select_rescore_kernel<PHASE 1>": ("select_rescore_kernel<1, false, false, 1>",),
        "select_rescore_kernel<PHASE 2>": ("select_rescore_kernel<1, false, true, 2>",),
        "select_rescore_kernel": ("select_rescore_kernel<1>", "select_rescore_kernel<1, false, false, 0>"),
        "decode_fwd_v4_kernel": ("decode_fwd_v4_kernel",)}
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
# effective shader clock of the dominant kernel DURING the counter pass: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the
# kernel's mean duration in the same pass's kernel trace (argv[3] = that pass's *_kernel_trace.csv)
if len(sys.argv) > 3:
    import csv
    for name, pats in KEYS.items():
        ks = [k for k in d if any(p in k for p in pats) and "GRBM_GUI_ACTIVE" in d[k]]
        if not ks or name not in out:
            continue
        durs = []
        with open(sys.argv[3]) as fh:
            for row in csv.DictReader(fh):
                kn = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                if any(p in kn for p in pats):
                    durs.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        if durs:
            cyc = d[ks[0]]["GRBM_GUI_ACTIVE"]["mean"] / 8.0
            ms = sum(durs) / len(durs) / 1e6
            out[name].update(effective_sclk_mhz=cyc / (ms * 1e3), grbm_cycles_per_xcd=cyc, duration_ms_in_counter_pass=ms)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in d[ks[0]]:
                out[name]["mfma_busy_frac"] = d[ks[0]]["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (1024.0 * cyc)
# the tree the counter passes ran on (bench.py compares it with the tree it benches: roofline.counter_pass.stale)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    out["_csrc_sha16"] = bench.csrc_sha16()
except Exception as e:  # noqa: BLE001
    out["_csrc_sha16"] = None
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))

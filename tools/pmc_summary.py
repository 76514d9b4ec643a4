#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs into per-kernel means (one row per kernel)."""
import csv
import glob
import json
import sys
from collections import defaultdict

out = {}
for d in sys.argv[2:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-90:]
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for name, ctrs in acc.items():
            for c, vals in ctrs.items():
                out.setdefault(name, {})[c] = {"mean": sum(vals) / len(vals), "n": len(vals)}
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for name, ctrs in sorted(out.items()):
    if any(k in name for k in ("gemm_kernel", "select_rescore", "decode_fwd", "topk_rows", "pre_acts")):
        print(name[-70:], {c: round(v["mean"], 1) for c, v in ctrs.items()})

#!/usr/bin/env python
"""Fused vs exact path on a REAL checkpoint and/or REAL cached activations (SURVEY.md section 7.3-7,
north_star: "outputs match ... on identical cached LLaVA-NeXT layer-24 activations").

    python tools/parity_real.py --sae_path <dir with cfg.json + sae.safetensors> \
                                --acts <file.safetensors with one [T, 4096] tensor> [--coarse int8|bf16]

Neither exists in the offline build image; until they do the same check runs on the synthetic
"trained_like" SAE (tests/hostile.py) -- which is what this command does without arguments.  Output (JSON, also
written to --out): tokens, verified-but-wrong tokens (must be 0), status histogram and fallback reasons,
fused and exact tokens/s.  Rows read per token: tools/rescore_stats.py (msae_options::rows_rescored).
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import soak_fused

if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--tokens" not in argv:
        argv += ["--tokens", "65536"]
    if "--out" not in argv:
        argv += ["--out", str(soak_fused.REPO / "gpurun_out" / "parity_real.json")]
    soak_fused.main(argv)

"""OPT-IN alias: `sae_auto_interp.*` -> `msae.*`, so the reference's README commands run verbatim on the HIP path
(round-4 verdict, item 9):

    PYTHONPATH=multimodal-sae_amd:multimodal-sae_amd/compat \\
        python -m sae_auto_interp.launch.cache.cache_image <model> <dataset> --sae_path ... --n_splits 128 --save_dir ...
    (reference README.md:46-56, 106-110; likewise torchrun ... -m sae_auto_interp.launch.features.steering ...)

and `from sae_auto_interp.sae import Sae, SaeConfig`, `from sae_auto_interp.features import FeatureImageCache`, ... resolve
to the drop-in modules.  Nothing imports this package unless `multimodal-sae_amd/compat` is put on the path on purpose --
side by side with a checkout of the reference on the same path it would shadow it, which is the point and the risk.

Only names that exist under `msae` resolve; the reference's other subpackages (explainers, scorers, clients: out of scope,
DESIGN.md section 8) raise ModuleNotFoundError naming the alias.

Mechanism: compat/_msae_alias.py (shared with the `sae` alias of the trainer's top-level package)."""
from __future__ import annotations

import _msae_alias

_msae_alias.install(__name__, "sae_auto_interp", "msae", "the SAE hot path")

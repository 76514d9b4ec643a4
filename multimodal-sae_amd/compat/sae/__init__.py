"""OPT-IN alias: the trainer's top-level package `sae` (reference train/sae/sae/) -> `msae.sae`, so that

    from sae import Sae, SaeConfig
    from sae.utils import eager_decode, triton_decode          # train/sae/tests/test_decode.py:3

resolve to the drop-in modules (`triton_decode` = the HIP gather-matmul, `eager_decode` = the reference's dense restatement,
sae/utils.py:108-116) and the reference's only test runs verbatim against this library (tests/test_alias_package.py).  Covers
`sae`, `sae.sae`, `sae.config`, `sae.utils`; the training LOOP of the reference (`sae.trainer.SaeTrainer`, wandb, data
loading) is out of scope -- its inner step is `msae.train.SaeTrainStep`.  Nothing imports this package
unless `multimodal-sae_amd/compat` is put on the path on purpose.  Mechanism: compat/_msae_alias.py."""
from __future__ import annotations

import _msae_alias

_msae_alias.install(__name__, "sae", "msae.sae", "Sae / SaeConfig / the decoder seam")

"""SaeConfig / TrainConfig -- field-for-field mirror of the reference dataclasses
(sae_auto_interp/sae/config.py:7-78) without the simple_parsing dependency.  `to_dict()` gives the
same keys the reference writes into cfg.json (sae.py:150-162)."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import List, Union


class _Serializable:
    def to_dict(self) -> dict:
        return dataclasses.asdict(self)

    @classmethod
    def from_dict(cls, d: dict, drop_extra_fields: bool = True):
        names = {f.name for f in dataclasses.fields(cls)}
        extra = set(d) - names
        if extra and not drop_extra_fields:
            raise TypeError(f"unexpected fields {sorted(extra)}")
        return cls(**{k: v for k, v in d.items() if k in names})


@dataclass
class SaeConfig(_Serializable):
    """Configuration of a TopK sparse autoencoder (config.py:7-29)."""

    expansion_factor: int = 32
    """Multiple of the input dimension to use as the SAE dimension."""
    normalize_decoder: bool = True
    """Normalize the decoder weights to have unit norm."""
    num_latents: int = 0
    """Number of latents to use. If 0, use `expansion_factor`."""
    k: int = 32
    """Number of nonzero features."""
    multi_topk: bool = False
    """Use Multi-TopK loss."""
    signed: bool = False
    """Kept so older checkpoints' cfg.json load."""


@dataclass
class TrainConfig(_Serializable):
    """Training hyper-parameters (config.py:32-78); consumed by the trainer row."""

    sae: SaeConfig = field(default_factory=SaeConfig)
    batch_size: int = 8
    grad_acc_steps: int = 1
    micro_acc_steps: int = 1
    lr: Union[float, None] = None
    lr_warmup_steps: int = 1000
    auxk_alpha: float = 0.0
    dead_feature_threshold: int = 10_000_000
    hookpoints: List[str] = field(default_factory=list)
    layers: List[int] = field(default_factory=list)
    layer_stride: int = 1
    distribute_modules: bool = False
    save_every: int = 1000
    log_to_wandb: bool = True
    run_name: Union[str, None] = None
    wandb_log_frequency: int = 1

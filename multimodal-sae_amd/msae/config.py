"""CLI configuration of the caching entry points -- same fields, defaults, positional arguments and
`--flag` names as the reference's `CacheConfig` (sae_auto_interp/config.py:76-117), parsed with
argparse instead of simple_parsing (not installed on the target image)."""
from __future__ import annotations

import argparse
import dataclasses
from dataclasses import dataclass
from typing import Optional, Sequence, Union


@dataclass
class CacheConfig:
    model: str = "EleutherAI/pythia-160m"
    """Name of the model to use (positional)."""
    dataset: str = "togethercomputer/RedPajama-Data-1T-Sample"
    """Path to the dataset (positional)."""
    sae_path: Union[str, None] = None
    """Path to your trained sae, can be either local or on the hub"""
    batch_size: int = 32
    """Number of sequences to process in a batch"""
    load_in_8bit: bool = False
    """Load the model in 8-bit mode."""
    split: str = "train"
    """Dataset split to use."""
    n_splits: int = 2
    """Number of splits to divide .safetensors into"""
    ctx_len: int = 2048
    """Context length of the autoencoder. Each batch is shape (batch_size, ctx_len)"""
    hf_token: Union[str, None] = None
    """Huggingface API token for downloading models."""
    save_dir: str = "./features_cache"
    """Save dir for your feature"""
    verbosity: str = "INFO"
    """Verbosity level"""
    filters_path: Optional[str] = None
    """The json file for filtering the features and sae should be in a json file"""

    def to_dict(self):
        return dataclasses.asdict(self)


@dataclass
class AttributionConfig:
    """sae_auto_interp/config.py:121-138."""
    model: str = "EleutherAI/pythia-160m"
    """Name of the model to use (positional)."""
    data_path: str = "./data/digit.json"
    """Path to the dataset. Should be a formated json file"""
    sae_path: Union[str, None] = None
    """Path to your trained sae, can be either local or on the hub"""
    selected_sae: str = "layers.24"
    """Name of the selected sae"""
    save_dir: str = "./attribution_cache"
    """Save dir for your feature attribution result"""
    method: str = "exact"
    """"exact": the reference's per-feature loop; "batched": all features from one forward + backward"""

    def to_dict(self):
        return dataclasses.asdict(self)


def parse_attribution_config(argv: Optional[Sequence[str]] = None) -> AttributionConfig:
    p = argparse.ArgumentParser(description="Attribution patching (MI355X HIP path)")
    for f in dataclasses.fields(AttributionConfig):
        if f.name == "model":
            p.add_argument("model", nargs="?", default=f.default, type=str)
        else:
            p.add_argument(f"--{f.name}", type=str, default=f.default)
    return AttributionConfig(**vars(p.parse_args(argv)))


_POSITIONAL = ("model", "dataset")


def parse_cache_config(argv: Optional[Sequence[str]] = None) -> CacheConfig:
    p = argparse.ArgumentParser(description="Cache SAE feature activations (MI355X HIP path)")
    for f in dataclasses.fields(CacheConfig):
        if f.name in _POSITIONAL:
            p.add_argument(f.name, nargs="?", default=f.default, type=str)
        elif f.type in (bool, "bool"):
            p.add_argument(f"--{f.name}", action=argparse.BooleanOptionalAction, default=f.default)
        else:
            typ = int if f.type in (int, "int") else str
            p.add_argument(f"--{f.name}", type=typ, default=f.default)
    return CacheConfig(**vars(p.parse_args(argv)))

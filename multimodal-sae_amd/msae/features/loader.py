"""Reader side of the feature-cache format -- the contract the writer in cache.py must satisfy.

Restates the naming / lookup rule of the reference's `FeatureDataset` and `TensorBuffer`
(sae_auto_interp/features/loader.py:28-90,143-196): the feature axis is cut at
`linspace(0, width, n_splits + 1)`, split i lives in `<raw_dir>/<module>/{start}_{end-1}.safetensors`,
a feature's records are the rows of that file whose third location column equals the feature id, and
consumers get `locations[:, :2]` = (row, position) plus the activations.  CPU-side post-processing,
deliberately not accelerated.
"""
from __future__ import annotations

import os
from typing import Dict, Iterator, List, NamedTuple, Optional

import torch
from safetensors.torch import load_file
from torch import Tensor


class FeatureRecords(NamedTuple):
    module: str
    feature: int
    locations: Tensor      # [n, 2] int64 (row, position)
    activations: Tensor    # [n] f32


def split_edges(width: int, n_splits: int) -> Tensor:
    return torch.linspace(0, width, steps=n_splits + 1).long()   # loader.py:143-144


def split_path(raw_dir: str, module: str, width: int, n_splits: int, feature: int) -> str:
    """File that holds `feature` (loader.py:164-187: bucketize(right=True) over the edges)."""
    edges = split_edges(width, n_splits)
    b = int(torch.bucketize(torch.tensor([feature]), edges, right=True)[0])
    start, end = int(edges[b - 1]), int(edges[b])
    return f"{raw_dir}/{module}/{start}_{end - 1}.safetensors"


class SplitBuffer:
    """One split file, lazily loaded; iterates / indexes per feature like TensorBuffer."""

    def __init__(self, path: str, module: str, features: Optional[Tensor] = None, min_examples: int = 0):
        self.path, self.module, self.features, self.min_examples = path, module, features, min_examples
        self.locations = self.activations = None

    def _load(self):
        if self.locations is None:
            data = load_file(self.path)
            self.locations, self.activations = data["locations"], data["activations"]
            if self.features is None:
                self.features = torch.unique(self.locations[:, 2])

    def get(self, feature: int) -> FeatureRecords:
        self._load()
        mask = self.locations[:, 2] == feature
        return FeatureRecords(self.module, int(feature), self.locations[mask][:, :2], self.activations[mask])

    def __iter__(self) -> Iterator[FeatureRecords]:
        self._load()
        for f in self.features.tolist():
            rec = self.get(f)
            if len(rec.activations) >= self.min_examples:   # loader.py:103-106
                yield rec


class FeatureDataset:
    """All (or selected) features of the cached modules (loader.py:130-196)."""

    def __init__(self, raw_dir: str, width: int, n_splits: int, modules: Optional[List[str]] = None,
                 features: Optional[Dict[str, Tensor]] = None, min_examples: int = 0):
        self.buffers: List[SplitBuffer] = []
        edges = split_edges(width, n_splits)
        modules = sorted(os.listdir(raw_dir)) if modules is None else modules
        for module in modules:
            if features is None:
                for s, e in zip(edges[:-1].tolist(), edges[1:].tolist()):
                    self.buffers.append(SplitBuffer(f"{raw_dir}/{module}/{s}_{e - 1}.safetensors", module,
                                                    min_examples=min_examples))
            else:
                sel = features[module]
                bucket = torch.bucketize(sel, edges, right=True)
                for b in torch.unique(bucket).tolist():
                    s, e = int(edges[b - 1]), int(edges[b])
                    self.buffers.append(SplitBuffer(f"{raw_dir}/{module}/{s}_{e - 1}.safetensors", module,
                                                    sel[bucket == b], min_examples=min_examples))

    def __len__(self):
        return len(self.buffers)

    def __iter__(self) -> Iterator[FeatureRecords]:
        for buf in self.buffers:
            yield from buf

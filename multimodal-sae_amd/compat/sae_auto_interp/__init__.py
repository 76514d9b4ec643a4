"""OPT-IN alias: `sae_auto_interp.*` -> `msae.*`, so the reference's README commands run verbatim on the HIP path
(round-4 verdict, item 9):

    PYTHONPATH=multimodal-sae_amd:multimodal-sae_amd/compat \\
        python -m sae_auto_interp.launch.cache.cache_image <model> <dataset> --sae_path ... --n_splits 128 --save_dir ...
    (reference README.md:46-56, 106-110; likewise torchrun ... -m sae_auto_interp.launch.features.steering ...)

and `from sae_auto_interp.sae import Sae, SaeConfig`, `from sae_auto_interp.features import FeatureImageCache`, ... resolve
to the drop-in modules.  Nothing imports this package unless `multimodal-sae_amd/compat` is put on the path on purpose --
side by side with a checkout of the reference on the same path it would shadow it, which is the point and the risk.

Mechanism: a meta-path finder answers every `sae_auto_interp[.x.y]` with a spec whose loader hands back the ALREADY IMPORTED
`msae[.x.y]` module object (one module, two names: class identity, isinstance checks and module state are shared) and
forwards `get_code` / `is_package` / `get_source` to the real loader, which is what `python -m` (runpy) needs.  Only names that
exist under `msae` resolve; the reference's other subpackages (explainers, scorers, clients: out of scope)
raise ModuleNotFoundError naming the alias."""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.util
import sys

_ALIAS, _TARGET = "sae_auto_interp", "msae"


def _target_name(fullname: str) -> str:
    return _TARGET + fullname[len(_ALIAS):]


class _AliasLoader(importlib.abc.InspectLoader):
    def __init__(self, target: str):
        self.target = target

    def _real(self):
        spec = importlib.util.find_spec(self.target)
        if spec is None or spec.loader is None:
            raise ImportError(f"{self.target} has no loader")
        return spec

    def create_module(self, spec):
        return importlib.import_module(self.target)      # the very module object msae.* is

    def exec_module(self, module):
        pass                                             # already executed under its own name

    def is_package(self, fullname):
        return self._real().submodule_search_locations is not None

    def get_code(self, fullname):                        # runpy: `python -m sae_auto_interp.launch...` runs msae's code
        return self._real().loader.get_code(self.target)

    def get_source(self, fullname):
        return self._real().loader.get_source(self.target)


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != _ALIAS and not fullname.startswith(_ALIAS + "."):
            return None
        tname = _target_name(fullname)
        try:
            real = importlib.util.find_spec(tname)
        except (ImportError, ValueError):
            real = None
        if real is None:
            raise ModuleNotFoundError(
                f"No module named {fullname!r}: the sae_auto_interp alias covers the SAE hot path only ({tname} does not exist; "
                "multimodal-sae_amd/compat shadows the reference package)", name=fullname)
        spec = importlib.util.spec_from_loader(fullname, _AliasLoader(tname), origin=real.origin,
                                               is_package=real.submodule_search_locations is not None)
        if real.submodule_search_locations is not None:
            spec.submodule_search_locations = list(real.submodule_search_locations)
        return spec


def _install() -> None:
    if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _AliasFinder())
    # this very package object: become msae itself for attribute access (`sae_auto_interp.sae`, `.features`, ...)
    me = sys.modules[__name__]
    real = importlib.import_module(_TARGET)
    me.__dict__.update({k: v for k, v in real.__dict__.items() if not k.startswith("__")})
    me.__path__ = list(real.__path__)


_install()

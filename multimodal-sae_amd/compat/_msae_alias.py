"""One module object under two names: the meta-path finder behind the opt-in alias packages of this directory
(`sae_auto_interp` -> `msae`, and the trainer's top-level `sae` -> `msae.sae`).

A finder answers every `<alias>[.x.y]` with a spec whose loader hands back the ALREADY IMPORTED `<target>[.x.y]` module object
(class identity, isinstance checks and module state are shared) and forwards `get_code` / `is_package` / `get_source` to the real
loader, which is what `python -m` (runpy) needs.  Only names that exist under the target resolve; everything else raises
ModuleNotFoundError naming the alias.  The aliased module keeps its OWN `__spec__` / `__name__` (importlib would otherwise
stamp the alias's spec on it: ADVICE r5)."""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.util
import sys


class _AliasLoader(importlib.abc.InspectLoader):
    def __init__(self, target: str):
        self.target = target
        self._own = None

    def _real(self):
        spec = importlib.util.find_spec(self.target)
        if spec is None or spec.loader is None:
            raise ImportError(f"{self.target} has no loader")
        return spec

    def create_module(self, spec):
        mod = importlib.import_module(self.target)       # the very module object the target name resolves to
        self._own = (getattr(mod, "__spec__", None), getattr(mod, "__loader__", None), getattr(mod, "__package__", None))
        return mod

    def exec_module(self, module):
        # already executed under its own name; give it back the attributes importlib's module_from_spec replaced
        if self._own is not None:
            module.__spec__, module.__loader__, module.__package__ = self._own

    def is_package(self, fullname):
        return self._real().submodule_search_locations is not None

    def get_code(self, fullname):                        # runpy: `python -m <alias>.launch...` runs the target's code
        return self._real().loader.get_code(self.target)

    def get_source(self, fullname):
        return self._real().loader.get_source(self.target)


class AliasFinder(importlib.abc.MetaPathFinder):
    def __init__(self, alias: str, target: str, scope: str):
        self.alias, self.target, self.scope = alias, target, scope

    def find_spec(self, fullname, path=None, target=None):
        if fullname != self.alias and not fullname.startswith(self.alias + "."):
            return None
        tname = self.target + fullname[len(self.alias):]
        try:
            real = importlib.util.find_spec(tname)
        except (ImportError, ValueError):
            real = None
        if real is None:
            raise ModuleNotFoundError(
                f"No module named {fullname!r}: the {self.alias} alias covers {self.scope} only ({tname} does not exist; "
                "multimodal-sae_amd/compat shadows the reference package)", name=fullname)
        spec = importlib.util.spec_from_loader(fullname, _AliasLoader(tname), origin=real.origin,
                                               is_package=real.submodule_search_locations is not None)
        if real.submodule_search_locations is not None:
            spec.submodule_search_locations = list(real.submodule_search_locations)
        return spec


def install(alias_module_name: str, alias: str, target: str, scope: str) -> None:
    """Called by the alias package's __init__: register the finder and make the package object itself read like the target."""
    if not any(isinstance(f, AliasFinder) and f.alias == alias for f in sys.meta_path):
        sys.meta_path.insert(0, AliasFinder(alias, target, scope))
    me = sys.modules[alias_module_name]
    real = importlib.import_module(target)
    me.__dict__.update({k: v for k, v in real.__dict__.items() if not k.startswith("__")})
    me.__path__ = list(real.__path__)

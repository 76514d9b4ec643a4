"""The opt-in `sae_auto_interp` / `sae` aliases (multimodal-sae_amd/compat): the reference's module paths and README commands
resolve to the drop-in package.  CPU only: imports, class identity, `python -m ... --help`.  (The reference's own test,
train/sae/tests/test_decode.py, runs through the `sae` alias on the GPU: tests/test_gpu_dropin.py.)"""
import os
import subprocess
import sys

from conftest import REPO

ENV = dict(os.environ, PYTHONPATH=os.pathsep.join([str(REPO / "multimodal-sae_amd"), str(REPO / "multimodal-sae_amd" / "compat")]))


def _py(code: str):
    return subprocess.run([sys.executable, "-c", code], env=ENV, capture_output=True, text=True, timeout=300, cwd="/tmp")


def test_reference_import_paths_resolve_to_the_drop_in_modules():
    r = _py("import sae_auto_interp, msae\n"
            "from sae_auto_interp.sae import Sae, SaeConfig\n"
            "from sae_auto_interp.sae.utils import decoder_impl\n"
            "from sae_auto_interp.features import FeatureCache, FeatureImageCache\n"
            "import sae_auto_interp.features.cache as c1, msae.features.cache as c2\n"
            "import msae.sae\n"
            "assert Sae is msae.sae.Sae and c1 is c2 and sae_auto_interp.sae is msae.sae\n"
            "s = Sae(16, SaeConfig(num_latents=64, k=4))\n"
            "assert isinstance(s, msae.Sae)\n"
            "try:\n"
            "    import sae_auto_interp.explainers\n"
            "    raise SystemExit('out-of-scope subpackage resolved')\n"
            "except ModuleNotFoundError as e:\n"
            "    assert 'alias' in str(e)\n"
            "print('ok')")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:]


def test_readme_commands_run_under_the_reference_module_paths():
    """reference README.md:46-56 / 106-110: `python -m sae_auto_interp.launch.cache.cache_image ...` etc. (--help: argument
    surface only; the launch scripts themselves run under torchrun in tests/test_gpu_dropin.py)."""
    for mod in ("sae_auto_interp.launch.cache.cache_image", "sae_auto_interp.launch.cache.cache",
                "sae_auto_interp.launch.features.steering", "sae_auto_interp.launch.features.attribution_patching"):
        r = subprocess.run([sys.executable, "-m", mod, "--help"], env=ENV, capture_output=True, text=True, timeout=300, cwd="/tmp")
        assert r.returncode == 0, (mod, r.stderr[-1500:])
        assert "sae_path" in r.stdout or "sae-path" in r.stdout, (mod, r.stdout[-500:])


def test_trainer_package_alias_and_module_specs_survive():
    """`sae` (the trainer's top-level package, reference train/sae/sae/) -> msae.sae; the aliased modules keep their own
    __spec__ / __name__ (importlib stamped the alias's spec on msae.sae)."""
    r = _py("import msae.sae, msae.sae.utils\n"
            "from sae.utils import eager_decode, triton_decode, decoder_impl\n"          # train/sae/tests/test_decode.py:3
            "from sae import Sae, SaeConfig\n"
            "import sae, sae.utils, sae_auto_interp.sae\n"
            "assert Sae is msae.sae.Sae and sae.utils is msae.sae.utils and eager_decode is msae.sae.utils.eager_decode\n"
            "assert decoder_impl is triton_decode\n"
            "assert msae.sae.__spec__.name == 'msae.sae' and msae.sae.utils.__spec__.name == 'msae.sae.utils', msae.sae.__spec__\n"
            "assert msae.sae.__name__ == 'msae.sae' and msae.sae.__package__ == 'msae.sae'\n"
            "try:\n"
            "    import sae.trainer\n"
            "    raise SystemExit('out-of-scope module resolved')\n"
            "except ModuleNotFoundError as e:\n"
            "    assert 'alias' in str(e)\n"
            "print('ok')")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:]


def test_sae_disable_triton_selects_the_eager_decoder():
    """The reference's import-time switch (sae/utils.py:119-129)."""
    r = subprocess.run([sys.executable, "-c", "from sae.utils import decoder_impl, eager_decode; assert decoder_impl is eager_decode; print('ok')"],
                       env=dict(ENV, SAE_DISABLE_TRITON="1"), capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-1500:]

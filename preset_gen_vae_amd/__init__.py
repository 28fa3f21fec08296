"""Importable alias of the ``preset-gen-vae_amd/`` package directory (a hyphen cannot appear in a Python module
name): ``import preset_gen_vae_amd`` executes ``preset-gen-vae_amd/__init__.py`` with that directory as the package
path, so ``preset_gen_vae_amd.model.build`` etc. resolve to the files under ``preset-gen-vae_amd/``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "preset-gen-vae_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))

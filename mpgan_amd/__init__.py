"""Importable alias of the product package.

The sources live in `cross-modality-minipig-gan_amd/` (a directory name Python
cannot import directly); this shim points the `mpgan_amd` package at that
directory and runs its `__init__`.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "cross-modality-minipig-gan_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f

"""Import alias: the package directory is named after the reference repository
(``vit-bias-aware-structural-distillation_amd``), which is not a valid Python
identifier.  ``import basd_amd`` resolves into that directory."""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "vit-bias-aware-structural-distillation_amd")
__path__ = [_REAL]
with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))

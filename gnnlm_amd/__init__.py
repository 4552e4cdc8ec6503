"""Import alias: the product package lives in the directory ``gnn-lm_amd/`` (not a valid Python
identifier), this shim makes it importable as ``gnnlm_amd``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gnn-lm_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f

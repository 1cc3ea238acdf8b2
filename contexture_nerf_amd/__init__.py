"""Import shim: the product package lives in `contexture-nerf_amd/` (hyphen, not importable by name).
This package forwards `contexture_nerf_amd.*` to that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "contexture-nerf_amd")
__path__[:] = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))

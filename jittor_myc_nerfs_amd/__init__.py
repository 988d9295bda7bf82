"""Importable alias for the package directory ``jittor-myc-nerfs_amd/`` (a hyphen is not a valid
Python identifier).  All code lives in that directory; this module only redirects ``__path__``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "jittor-myc-nerfs_amd")
__path__[:] = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f

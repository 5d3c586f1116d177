"""Import alias: the sources live in ``patch-ioner_amd/`` (a directory name python cannot import),
this one-file package redirects ``import patchioner_amd[.x]`` there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "patch-ioner_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f, _real

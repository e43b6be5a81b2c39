"""Import shim: the product package lives in ``face-crop-plus_amd/`` (a name
Python cannot import directly); this package simply extends its search path to
that directory so ``import face_crop_plus_amd`` works from the repo root."""
import os as _os

_here = _os.path.dirname(_os.path.abspath(__file__))
_real = _os.path.join(_os.path.dirname(_here), "face-crop-plus_amd")
__path__.append(_real)

with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))

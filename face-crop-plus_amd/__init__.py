"""face-crop-plus hot path, MI355X-native (gfx950 HIP kernels behind a C ABI).

Mirrors the reference's public surface (``src/face_crop_plus/__init__.py:1``,
``models/__init__.py:1-3``): ``Cropper``, ``RetinaFace``, ``RRDBNet``, ``BiSeNet``.
Heavy imports are lazy so that ``import face_crop_plus_amd`` never needs a GPU.
"""
__version__ = "0.1.0"

_LAZY = {
    "Cropper": "cropper",
    "RetinaFace": "retinaface",
    "RRDBNet": "rrdb",
    "BiSeNet": "bise",
}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        mod = importlib.import_module(f"{__name__}.{_LAZY[name]}")
        return getattr(mod, name)
    raise AttributeError(name)

"""File decode / encode of the host pipeline (Pillow), free of torch imports: the worker processes of ``_io_pool.py``
import only this module.  ``utils.py`` re-exports ``read_image`` / ``write_image`` (reference ``utils.py:228-271``,
``cropper.py:605-609``)."""
from __future__ import annotations

import os
import warnings

import numpy as np


def read_image(path: str):
    """One file -> RGB uint8 HWC array, or None (with the reference's warning) when it cannot be read.
    Like ``cv2.imread`` (utils.py:262), the EXIF orientation tag is applied: phone photos arrive upright, and
    user-supplied landmark files — which live in that oriented frame — point at the right pixels."""
    from PIL import Image, ImageOps
    try:
        with Image.open(path) as im:
            im = ImageOps.exif_transpose(im)
            return np.asarray(im.convert("RGB"), dtype=np.uint8)
    except Exception:
        warnings.warn(f"Could not read the image {path}")
        return None


# Encoder settings of ``cv2.imwrite`` with no parameters (cropper.py:605-609), so that files written here have the
# fidelity and roughly the size of the reference's: JPEG quality 95 with 4:2:0 chroma subsampling (Pillow's own
# default, quality 75, is visibly lossier), PNG at zlib level 1 (lossless either way: only size / speed differ),
# WebP lossless (OpenCV's default quality setting means lossless).
_ENCODER_KW = {
    ".jpg": dict(format="JPEG", quality=95, subsampling="4:2:0"),
    ".jpeg": dict(format="JPEG", quality=95, subsampling="4:2:0"),
    ".jpe": dict(format="JPEG", quality=95, subsampling="4:2:0"),
    ".png": dict(format="PNG", compress_level=1),
    ".webp": dict(format="WEBP", lossless=True),
    ".bmp": dict(format="BMP"),
    ".tif": dict(format="TIFF"),
    ".tiff": dict(format="TIFF"),
}


_NOT_CV2_FORMATS = {"GIF", "PDF", "ICO", "ICNS", "PALM", "MPO", "XBM", "IM", "MSP", "PCX", "DDS", "TGA", "SGI", "EPS", "SPIDER",
                    "BLP", "BUFR", "GRIB", "HDF5", "DIB", "APNG"}   # Pillow writes them, cv2.imwrite refuses: skipped like there


def write_image(path: str, image: np.ndarray) -> bool:
    """RGB (or single-channel mask) uint8 array -> file; format from the extension, ``cv2.imwrite`` defaults for the
    formats listed above, Pillow's own choice of encoder for every other extension it knows (.ppm / .pgm / .pnm /
    .jp2 / ... — ``cv2.imwrite`` writes these too).  Only an extension NO encoder exists for warns and returns False
    (the file is skipped) instead of raising."""
    from PIL import Image
    kw = _ENCODER_KW.get(os.path.splitext(path)[1].lower())
    if kw is not None:
        Image.fromarray(image).save(path, **kw)
        return True
    ext = os.path.splitext(path)[1].lower()
    Image.init()                                    # fill Pillow's extension -> encoder registry
    fmt = Image.registered_extensions().get(ext)
    if fmt is None or fmt.upper() not in Image.SAVE or fmt.upper() in _NOT_CV2_FORMATS:
        # no encoder for this extension (here or in cv2.imwrite): the reference's writer returns False and goes on
        warnings.warn(f"Could not write the image {path}: no encoder for the extension {ext!r}")
        return False
    # real I/O errors (disk full, permissions, a missing directory) propagate, like everywhere else in the writer; the
    # file goes to a temporary name first so that a failed write never leaves a truncated image behind
    tmp = f"{path}.part{os.getpid()}"
    try:
        Image.fromarray(image).save(tmp, format=fmt)
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return True

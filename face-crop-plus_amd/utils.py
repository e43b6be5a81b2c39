"""Host utilities around the hot path (mirror of the reference's ``utils.py``
surface that ``Cropper`` needs).  Image file decode / encode uses Pillow: OpenCV
is not a dependency of this build.  ``as_batch`` runs on the
GPU (``batch.py`` / ``csrc/fcp_batch.hip``: OpenCV's INTER_AREA / INTER_CUBIC
uint8 arithmetic restated).
"""
from __future__ import annotations

import json
import os
import warnings

import numpy as np

from ._io_codec import read_image, write_image  # noqa: F401  (torch-free: shared with the I/O worker processes)
from .align import STANDARD_LANDMARKS_5  # noqa: F401  (re-export, utils.py:13-19)


def parse_landmarks_file(file_path: str, **kwargs):
    """json / csv / txt landmark files -> ((N,k,2) float32, (N,) str) (utils.py:21-88)."""
    if file_path.endswith(".json"):
        with open(file_path, "r") as f:
            data = json.load(f)
        filenames = np.array(list(data.keys()))
        landmarks = np.array(list(data.values()), dtype=np.float32)
    else:
        if file_path.endswith(".csv"):
            kwargs.setdefault("delimiter", ",")
            kwargs.setdefault("skip_header", 1)
        filenames = np.atleast_1d(np.genfromtxt(file_path, usecols=0, dtype=str, **kwargs))
        landmarks = np.atleast_2d(np.genfromtxt(file_path, dtype=np.float32, **kwargs))[:, 1:]
    return landmarks.reshape(len(landmarks), -1, 2), filenames


_SLICES_5 = {
    5: [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)],
    12: [(10, 11), (11, 12), (2, 3), (3, 4), (4, 5)],
    17: [(2, 5), (7, 10), (10, 11), (13, 14), (16, 17)],
    21: [(6, 9), (9, 12), (14, 15), (17, 18), (19, 20)],
    29: [(4, 9), (13, 18), (19, 20), (22, 23), (27, 28)],
    49: [(19, 25), (25, 31), (13, 14), (31, 32), (37, 38)],
    68: [(36, 42), (42, 48), (30, 31), (48, 49), (54, 55)],
    98: [(60, 68), (68, 76), (54, 55), (76, 77), (82, 83)],
    106: [(66, 75), (75, 84), (54, 55), (85, 86), (91, 92)],
}


def get_landmark_slices_5(num_landmarks: int):
    """utils.py:90-132."""
    if num_landmarks not in _SLICES_5:
        raise ValueError(f"Invalid number of landmarks: {num_landmarks}")
    return [slice(*x) for x in _SLICES_5[num_landmarks]]


def get_ldm_slices(num_tgt_landmarks: int, num_src_landmarks: int):
    """utils.py:134-168."""
    if num_tgt_landmarks != 5:
        raise ValueError(f"The number of target (standard) landmarks is not supported {num_tgt_landmarks}")
    return get_landmark_slices_5(num_src_landmarks)


def read_images(file_names, input_dir, pool=None):
    """-> (list of RGB uint8 HWC arrays, ndarray of surviving file names); unreadable files warn and are
    skipped (utils.py:228-271).  ``pool``: optional executor decoding the files concurrently (Pillow releases
    the GIL while decoding); the result order is the input order either way."""
    paths = [os.path.join(input_dir, f) for f in file_names]
    decoded = list(pool.map(read_image, paths)) if pool is not None else [read_image(p) for p in paths]
    indices = [i for i, im in enumerate(decoded) if im is not None]
    return [decoded[i] for i in indices], np.array(file_names)[indices]


def as_batch(images, size=512, padding_mode: str = "constant", device="cuda:0"):
    """Aspect-preserving resize + centred padding to a common size (utils.py:273-342), reference
    signature: returns (batch (N,H,W,3) uint8 numpy, unscales (N,), paddings (N,4) int64 [t,b,l,r]).
    The pixels are produced on the GPU by ``batch.build_batch`` (which ``Cropper`` calls directly to
    keep the batch on the device); there is no CPU resampler in this build."""
    from .batch import build_batch
    batch, unscales, paddings = build_batch(images, size, padding_mode, device)
    return batch.cpu().numpy(), unscales, paddings


_DEFAULT_EXCLUDE = frozenset("\00!@#$%^&*?={}:;'<>,.?/\\|" + '"')


def _to_ascii(name: str) -> str:
    """Transliterate to ASCII: ``unidecode`` when it is installed (what the reference calls, utils.py:426),
    otherwise NFKD decomposition with the non-ASCII remainder dropped ("České" -> "Ceske")."""
    try:
        import unidecode
        return unidecode.unidecode(name)
    except ImportError:
        import unicodedata
        return unicodedata.normalize("NFKD", name).encode("ascii", "ignore").decode("ascii")


def clean_names(input_dir: str, output_dir: str | None = None, max_chars: int = 250, exclude=_DEFAULT_EXCLUDE,
                desc: str | None = "Cleaning file names"):
    """Make the file names of a directory OS-safe (utils.py:344-453): ASCII transliteration of the stem,
    removal of the ``exclude`` characters from the stem (so one dot remains, the extension's), truncation so
    that directory + name stays within ``max_chars``, and "-N" suffixes until names are unique
    case-insensitively.  Files are copied to ``output_dir`` under the new names, or renamed in place when it is
    None."""
    import shutil
    budget = max_chars - len(input_dir)
    if budget <= 5:
        raise RuntimeError(f"Directory path length is too long ({len(input_dir)}) Either reduce the length of the "
                           f"directory name or increase `max_chars`.")
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
    names = os.listdir(input_dir)
    if desc is not None:
        try:
            import tqdm
            names = tqdm.tqdm(names, desc=desc)
        except ImportError:
            pass
    seen: dict[str, int] = {}
    drop = set(exclude)
    for original in names:
        stem, ext = os.path.splitext(original)
        if not stem.isascii():
            stem = _to_ascii(stem)
        if drop & set(stem):
            stem = "".join(ch for ch in stem if ch not in drop)
        if len(original) > budget:
            stem = stem[:budget - len(ext)]
        key = (stem + ext).lower()
        seen[key] = seen.get(key, -1) + 1
        while seen[key] > 0:                     # taken: append the running count, then re-check the new name
            stem += f"-{seen[key]}"
            key = (stem + ext).lower()
            seen[key] = seen.get(key, -1) + 1
        if output_dir is not None:
            shutil.copy(os.path.join(input_dir, original), os.path.join(output_dir, stem + ext))
        elif stem + ext != original:
            os.rename(os.path.join(input_dir, original), os.path.join(input_dir, stem + ext))

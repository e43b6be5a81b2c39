"""5-point align + crop on device (reference ``Cropper.crop_align``,
cropper.py:441-552, whose arithmetic is cv2.estimateAffine*2D + cv2.warpAffine)."""
from __future__ import annotations

import numpy as np
import torch

from . import _native as N
from . import torch_ops as T

BORDER_MODES = {"constant": 0, "replicate": 1, "reflect": 2, "wrap": 3, "reflect_101": 4,
                "reflect101": 4, "default": 4}

STANDARD_LANDMARKS_5 = np.float32([
    [0.31556875000000000, 0.4615741071428571],
    [0.68262291666666670, 0.4615741071428571],
    [0.50026249999999990, 0.6405053571428571],
    [0.34947187500000004, 0.8246919642857142],
    [0.65343645833333330, 0.8246919642857142],
])


def border_code(padding: str) -> int:
    """``getattr(cv2, f"BORDER_{padding.upper()}")`` (cropper.py:512)."""
    key = padding.lower()
    if key not in BORDER_MODES:
        raise AttributeError(f"module 'cv2' has no attribute 'BORDER_{padding.upper()}'")
    return BORDER_MODES[key]


def estimate_transform(landmarks: torch.Tensor, target: torch.Tensor, allow_skew: bool = False,
                       face_count: torch.Tensor | None = None, valid_total: torch.Tensor | None = None):
    """landmarks (F,k,2) f32 device, target (k,2) f32 device -> (mat (F,6) f64, ok (F,) i32).
    ``face_count``: device int32 scalar (view), live rows of a fixed-capacity face array — rows beyond it get ok = 0;
    ``valid_total``: device int64 scalar the number of ok faces is added to (both optional, no host read-back)."""
    f, k = landmarks.shape[0], landmarks.shape[1]
    dev = landmarks.device
    if T.ENABLED:
        mat, ok = T.load().similarity_from_5pt(landmarks.contiguous(), target.contiguous(), bool(allow_skew), face_count,
                                               valid_total)
        return mat.view(f, 6), ok
    mat = torch.empty((f, 6), dtype=torch.float64, device=dev)
    ok = torch.empty((f,), dtype=torch.int32, device=dev)
    N.check(N.lib().fcp_estimate_transform_counted(N.ptr(landmarks.contiguous()), N.ptr(target.contiguous()), f, k,
                                                   int(bool(allow_skew)), N.ptr(face_count), N.ptr(mat), N.ptr(ok),
                                                   N.ptr(valid_total), N.stream_ptr()),
            "fcp_estimate_transform")
    return mat, ok


def warp_affine(images_u8: torch.Tensor, img_idx: torch.Tensor, mat: torch.Tensor, ok: torch.Tensor | None,
                paddings: torch.Tensor | None, output_size, border: int = 0) -> torch.Tensor:
    """images (n,h,w,3) u8 device; output_size = (width, height) like cv2's dsize."""
    assert images_u8.dtype == torch.uint8 and images_u8.is_contiguous() and images_u8.shape[3] == 3
    n, h, w, _ = images_u8.shape
    f = img_idx.shape[0]
    ow, oh = int(output_size[0]), int(output_size[1])
    if T.ENABLED:
        return T.load().warp_affine_u8(images_u8, img_idx, mat.contiguous().view(f, 2, 3), ok, paddings, ow, oh, int(border))
    out = torch.empty((f, oh, ow, 3), dtype=torch.uint8, device=images_u8.device)
    N.check(N.lib().fcp_warp_affine_u8(N.ptr(images_u8), n, h, w, N.ptr(img_idx), N.ptr(mat), N.ptr(ok),
                                       N.ptr(paddings), f, oh, ow, int(border), N.ptr(out), N.stream_ptr()),
            "fcp_warp_affine_u8")
    return out


def crop_align(images_u8, img_idx, landmarks, target, output_size, border=0, allow_skew=False, paddings=None,
               face_count=None, valid_total=None):
    """Device crop_align: -> (crops (F,oh,ow,3) u8, ok (F,) i32, mat (F,6) f64).  Faces
    with ok == 0 (degenerate transform) are dropped by the caller like cropper.py:529-531.
    ``face_count`` / ``valid_total``: see ``estimate_transform``."""
    dev = images_u8.device
    landmarks = landmarks.to(device=dev, dtype=torch.float32)
    if not isinstance(target, torch.Tensor):
        target = torch.from_numpy(np.ascontiguousarray(target, dtype=np.float32))
    target = target.to(dev)
    img_idx = img_idx.to(device=dev, dtype=torch.int32).contiguous()
    if paddings is not None:
        paddings = paddings.to(device=dev, dtype=torch.int32).contiguous()
    mat, ok = estimate_transform(landmarks, target, allow_skew, face_count, valid_total)
    crops = warp_affine(images_u8, img_idx, mat, ok, paddings, output_size, border)
    return crops, ok, mat

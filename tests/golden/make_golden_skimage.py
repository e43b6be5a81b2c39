"""Third-party pin of the least-squares SIMILARITY estimate (reference cropper.py:515-519).

OpenCV is absent from the build container, but its image carries scikit-image 0.18.3 under /opt/conda (python3.9).
`skimage.transform.SimilarityTransform.estimate` is Umeyama's closed form for argmin sum |s R x + t - y|^2 — an
independently written implementation of the optimum `cv2.estimateAffinePartial2D(..., ransacReprojThreshold=inf)`
converges to (RANSAC accepts its first sample, ten Levenberg-Marquardt iterations on a linear residual follow:
oracle/align_ref.py::estimate_transform_cv_sequence).  This script records skimage's matrices for the same 64 + 2
five-point sets `tools/make_cv2_fixture.py` would hand to OpenCV; tests hold the oracle and the HIP kernel to them.

    /opt/conda/bin/python3.9 tests/golden/make_golden_skimage.py        # writes tests/golden/skimage_similarity.npz

Only data is stored (inputs, outputs, library versions)."""
import os
import warnings

import numpy as np

warnings.simplefilter("ignore")
import skimage
from skimage.transform import SimilarityTransform

HERE = os.path.dirname(os.path.abspath(__file__))
# the reference's 5-point target at 256 x 256, face factor 0.65: its own table through its own arithmetic
# (cropper.py:423-439), recorded from the reference's function in host_logic.npz
host = np.load(os.path.join(HERE, "host_logic.npz"))
std = host["standard_landmarks_5"].astype(np.float32).copy()
std[:, 0] *= 256 * 0.65
std[:, 1] *= 256 * 0.65
std[:, 0] += (1 - 0.65) * 256 / 2
std[:, 1] += (1 - 0.65) * 256 / 2
tgt = std

rng = np.random.default_rng(101)                      # the generator tools/make_cv2_fixture.py::make_align uses
srcs = []
for i in range(64):
    ang, sc = rng.uniform(-0.6, 0.6), rng.uniform(0.3, 4.0)
    R = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]]) * sc
    srcs.append((tgt @ R.T + rng.uniform(0, 3000, 2) + rng.normal(0, 2.0 * sc, (5, 2))).astype(np.float32))
srcs.append(np.stack([np.linspace(0, 40, 5), np.linspace(0, 40, 5)], 1).astype(np.float32))      # collinear: still determined
srcs = np.stack(srcs)
mats = np.zeros((len(srcs), 2, 3), np.float64)
for i, s in enumerate(srcs):
    t = SimilarityTransform()
    assert t.estimate(s.astype(np.float64), tgt.astype(np.float64))
    mats[i] = t.params[:2]
np.savez_compressed(os.path.join(HERE, "skimage_similarity.npz"), est_src=srcs, est_dst=tgt, est_mat=mats,
                    skimage_version=np.array(skimage.__version__), numpy_version=np.array(np.__version__))
print("wrote skimage_similarity.npz", srcs.shape, mats.shape, skimage.__version__)

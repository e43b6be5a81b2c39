"""Third-party check of INTER_AREA at integral ratios (reference utils.py:320-335): there
cv2.resize(..., INTER_AREA) is the mean of each ratio x ratio block, rounded; scikit-image 0.18.3's
`transform.downscale_local_mean` computes those means independently (float64).  OpenCV itself is not in the image.

    /opt/conda/bin/python3.9 tests/golden/make_golden_skimage_area.py     # writes tests/golden/skimage_area.npz"""
import os
import warnings

import numpy as np

warnings.simplefilter("ignore")
import skimage
from skimage.transform import downscale_local_mean

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(55)
out = {"skimage_version": np.array(skimage.__version__)}
for k, (h, w, f) in enumerate(((96, 128, 2), (150, 210, 3), (256, 192, 4), (120, 200, 5))):
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    out[f"img{k}"], out[f"factor{k}"] = img, np.array(f)
    out[f"mean_x64_{k}"] = np.rint(downscale_local_mean(img.astype(np.float64), (f, f, 1)) * 64).astype(np.uint16)   # 64ths
out["cases"] = np.array(4)
np.savez_compressed(os.path.join(HERE, "skimage_area.npz"), **out)
print("wrote skimage_area.npz", skimage.__version__)

"""ORACLE (test infrastructure only -- never imported by the product path).

numpy restatement of the reference's window transform chain `get_trans("edfx")`
(/root/reference/src/dataset/dataset.py:10-19), one MONAI transform per line:

  LoadImageD(keys='eeg')                               np.load of the per-recording .npy  (1, n_samples)
  ScaleIntensityD(factor=1e6)                          x * (1 + factor)
  ScaleIntensityD(minv=0, maxv=1)                      (x - min) / (max - min) over the WHOLE recording
                                                       (monai rescale_array: a constant array maps to x * minv = 0)
  RandSpatialCropD(roi_size=[3000], random_size=False) x[..., s:s+3000], s uniform in [0, n-3000]
  BorderPadD(spatial_border=[36], mode="constant")     36 zeros each side -> (1, 3072)

MONAI itself is not installed here (PARITY UNPINNED at the library boundary); the restatement follows the documented
semantics of those transforms.  The random crop offset is an INPUT so loader and oracle see the same draw.
"""
import numpy as np


def get_trans_edfx(recording, start, factor=1e6, roi=3000, border=36):
    x = np.asarray(recording, dtype=np.float32)
    x = x * np.float32(1.0 + factor)
    lo, hi = x.min(), x.max()
    x = (x - lo) / (hi - lo) if hi != lo else x * np.float32(0.0)
    x = x.reshape(1, -1)[:, start:start + roi]
    return np.pad(x, ((0, 0), (border, border)), mode="constant")

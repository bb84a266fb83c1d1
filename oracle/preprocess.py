"""Oracle: ``meta_preprocess`` / ``u16_to_u8``.  TEST INFRASTRUCTURE ONLY.

Restates src/image_tools.py:86-101.  The arithmetic lives in OpenCV (``opencv-contrib-python~=4.6.0.66``,
env.yml:20), which is not vendored in the reference and not installed here -> PARITY UNPINNED.
What is restated is OpenCV's published behaviour:

* ``cv2.convertScaleAbs(img, alpha)`` on uint16: ``saturate_cast<uchar>(|float(src) * float(alpha)|)`` with the
  product in float32 and ``cvRound`` (round half to even);
* ``cv2.threshold(img, 0, 1, THRESH_BINARY + THRESH_OTSU)`` on uint8: the threshold ``t`` maximising the
  between-class variance ``q1 * q2 * (mu1 - mu2)^2`` over the 256-bin histogram (first maximum wins, classes with
  weight < FLT_EPSILON skipped), output 1 where ``img > t``.
Only the ">50 % white -> invert" decision (:94-95) reaches the output.
"""
import numpy as np

_FLT_EPSILON = float(np.finfo(np.float32).eps)


def u16_to_u8(img):
    """src/image_tools.py:98-101."""
    img = np.asarray(img)
    if img.dtype == np.uint16:
        a = np.float32(255.0 / 65535.0)
        v = np.abs(img.astype(np.float32) * a)
        return np.clip(np.rint(v), 0, 255).astype(np.uint8)
    return img


def otsu_threshold_u8(img):
    hist = np.bincount(np.asarray(img, np.uint8).ravel(), minlength=256).astype(np.float64)
    scale = 1.0 / img.size
    mu = float((np.arange(256) * hist).sum()) * scale
    mu1 = 0.0
    q1 = 0.0
    max_sigma, max_val = 0.0, 0
    for i in range(256):
        p_i = hist[i] * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < _FLT_EPSILON or max(q1, q2) > 1.0 - _FLT_EPSILON:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2)
        if sigma > max_sigma:
            max_sigma, max_val = sigma, i
    return max_val


def meta_preprocess(img):
    """src/image_tools.py:86-96 -> uint8 (H, W): blue channel of an RGB read, inverted when Otsu marks more than
    half of the pixels as foreground."""
    img = u16_to_u8(img)
    if img.ndim > 2:
        img = img[:, :, 2]
    img = np.ascontiguousarray(img)
    t = otsu_threshold_u8(img)
    white = int(np.count_nonzero(img > t))
    if white > img.shape[0] * img.shape[1] * 0.5:
        img = ~img
    return img

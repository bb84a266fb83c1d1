"""Oracle: uint8 quantisation of the stitched probabilities and the per-pixel argmax.
TEST INFRASTRUCTURE ONLY.  Restates src/utils.py:117-118.  PINNED by tests/golden/quant_argmax.npz.
"""
import numpy as np


def quantise_u8(p):
    """``skimage.img_as_ubyte`` on a float image (src/utils.py:117): values must lie in [-1, 1];
    ``rint(float64(p) * 255)`` (round half to even) clipped to [0, 255]."""
    p = np.asarray(p)
    if p.size and (p.min() < -1.0 or p.max() > 1.0):
        raise ValueError("Images of type float must be between -1 and 1.")
    q = np.rint(p.astype(np.float64) * 255.0)
    return np.clip(q, 0, 255).astype(np.uint8)


def argmax_first(q):
    """``np.argmax(I, axis=2)`` (src/utils.py:118): first index of the maximum, int64."""
    return np.argmax(q, axis=-1).astype(np.int64)


def quantised_argmax(p):
    return argmax_first(quantise_u8(p))

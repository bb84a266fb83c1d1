"""Oracle: ``meta_inference`` morphological clean-up and the component counts.
TEST INFRASTRUCTURE ONLY.

Restates src/image_tools.py:15-84 (meta_inference), :103-112 (count_HSR), :114-119 (count_cc),
:126-134 (count_colocalization) with numpy + scipy.ndimage only (skimage is not installed on the GPU
box).  PINNED bit-exactly by tests/golden/meta_inference_*.npz and counting.npz, which hold outputs of
the reference's own functions.
"""
import numpy as np
from scipy import ndimage as ndi

EC_SIZE_THRESHOLD = 15          # src/image_tools.py:13
HSR_SIZE_THRESHOLD = 20         # src/meta_overlay.py:12
_FULL = np.ones((3, 3), bool)   # 8-connectivity (skimage.measure.label default; generate_binary_structure(2, 2))
_CROSS = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)  # skimage diamond(1); 4-connectivity


def label8(mask):
    """8-connected labelling, labels numbered by raster order of each component's first pixel."""
    return ndi.label(mask, structure=_FULL)


def label4(mask):
    return ndi.label(mask, structure=_CROSS)


def _areas(lab, n):
    return np.bincount(lab.ravel(), minlength=n + 1)[1:]


def _mean(areas):
    """``np.mean`` of a list of Python ints: exact integer sum / n in float64, NaN when empty."""
    return float(areas.sum()) / len(areas) if len(areas) else float('nan')


def fill_holes(img, c):
    """src/image_tools.py:36-39: ``binary_fill_holes(img == c)`` = everything except the pixels != c that
    are 4-connected to the outside of the image; filled pixels are overwritten with ``c``."""
    bg = img != c
    lab, n = label4(bg)
    open_ids = np.unique(np.concatenate([lab[0, :], lab[-1, :], lab[:, 0], lab[:, -1]]))
    is_open = np.zeros(n + 1, bool)
    is_open[open_ids] = True
    is_open[0] = False
    img[~is_open[lab]] = c       # class-c pixels (lab == 0) are re-written with c: a no-op
    return img


def size_thresh(img):
    """src/image_tools.py:41-59."""
    nuc, n_n = label8(img == 1)
    chrom, n_c = label8(img == 2)
    avg_chrom = _mean(_areas(chrom, n_c))
    a_n = _areas(nuc, n_n)
    kill = np.zeros(n_n + 1, bool)
    kill[1:] = a_n < avg_chrom                     # comparisons with NaN are False
    img[kill[nuc]] = 0

    chrom, n_c = label8(img == 2)
    ec, n_e = label8(img == 3)
    a_e = _areas(ec, n_e)
    avg_ec = _mean(a_e)
    a_c = _areas(chrom, n_c)
    to_ec = np.zeros(n_c + 1, bool)
    to_ec[1:] = a_c < avg_ec
    img[to_ec[chrom]] = 3
    small = np.zeros(n_e + 1, bool)                # ec regions as labelled BEFORE the reassignment above
    small[1:] = a_e < EC_SIZE_THRESHOLD
    img[small[ec]] = 0
    return img


def ec_band_removal(img):
    """src/image_tools.py:64; skimage binary_dilation pads with 0, binary_erosion with 1."""
    ec = img == 3
    band = ndi.binary_dilation(ec, structure=_CROSS) ^ ndi.binary_erosion(ec, structure=_CROSS, border_value=1)
    img[band] = 0
    return img


def _centroids(lab, n):
    """regionprops centroid = mean of pixel coordinates (float64; integer sums are exact)."""
    H, W = lab.shape
    cnt = np.bincount(lab.ravel(), minlength=n + 1)[1:].astype(np.float64)
    yy, xx = np.indices((H, W))
    sy = np.bincount(lab.ravel(), weights=yy.ravel(), minlength=n + 1)[1:]
    sx = np.bincount(lab.ravel(), weights=xx.ravel(), minlength=n + 1)[1:]
    return sy / cnt, sx / cnt


def nucleus_in_metaphase(img, v=70.0, min_chrom_count=5):
    """src/image_tools.py:66-81: a nucleus with more than five chromosome centroids strictly inside each of
    the four 70-px half-bands (x right, x left, y above, y below; x and y counted independently) is erased."""
    chrom, n_c = label8(img == 2)
    nuc, n_n = label8(img == 1)
    cy, cx = _centroids(chrom, n_c) if n_c else (np.zeros(0), np.zeros(0))
    ny, nx = _centroids(nuc, n_n) if n_n else (np.zeros(0), np.zeros(0))
    kill = np.zeros(n_n + 1, bool)
    for k in range(n_n):
        left = np.count_nonzero((cx > nx[k]) & (cx < nx[k] + v)) > min_chrom_count
        right = np.count_nonzero((cx < nx[k]) & (cx > nx[k] - v)) > min_chrom_count
        bottom = np.count_nonzero((cy < ny[k]) & (cy > ny[k] - v)) > min_chrom_count
        top = np.count_nonzero((cy > ny[k]) & (cy < ny[k] + v)) > min_chrom_count
        kill[k + 1] = left and right and bottom and top     # the expression at :80 reduces to all four
    img[kill[nuc]] = 0
    return img


def merge_comp(img, c):
    """src/image_tools.py:18-33.  ``m`` = the other class; it is lifted out, the remaining non-zero pixels
    (classes c and 3) are labelled 8-connected, every component except the LAST one in label order
    (``range(1, num_features)``) that contains a ``c`` pixel becomes ``c`` entirely, then a grey opening with
    the 3x3 cross (scipy default ``mode='reflect'``) flips pixels whose opened value equals ``c``."""
    m = 2 if c == 1 else 1
    lifted = img == m
    img[lifted] = 0
    lab, n = ndi.label(img, structure=_FULL)
    has_c = np.zeros(n + 1, bool)
    has_c[np.unique(lab[img == c])] = True
    has_c[0] = False
    if n >= 1:
        has_c[n] = False                           # the highest label is never visited
    img[has_c[lab]] = c
    opened = ndi.grey_dilation(ndi.grey_erosion(img, footprint=_CROSS), footprint=_CROSS)
    img[opened == c] = c
    img[lifted] = m
    return img


def meta_inference(img):
    """src/image_tools.py:15-84 on an integer label image with values 0..3; returns a new int64 array."""
    img = np.array(img, dtype=np.int64, copy=True)
    fill_holes(img, 1)
    fill_holes(img, 2)
    size_thresh(img)
    ec_band_removal(img)
    nucleus_in_metaphase(img)
    merge_comp(img, 1)
    merge_comp(img, 2)
    img[ndi.binary_dilation(img == 3, structure=_CROSS)] = 3     # :83
    return img


# ---------------------------------------------------------------------------------------------
# counting
# ---------------------------------------------------------------------------------------------
def count_cc(mask):
    """src/image_tools.py:114-119 -> (number of 8-connected components, total pixels in them).
    The second value is ``np.sum([])`` = float 0.0 when there is no component, an integer otherwise."""
    mask = np.asarray(mask).astype(bool)
    _, n = label8(mask)
    if n == 0:
        return 0, 0.0
    if mask.all():
        # ``np.unique(labels)[1:]`` (:117) assumes label 0 is present; on a mask without any background
        # pixel it drops component 1 instead, leaving ``np.sum([])`` = 0.0
        return int(n), 0.0
    return int(n), int(mask.sum())


def count_colocalization(ob1, ob2):
    """src/image_tools.py:126-134: 8-connected components of ``ob1`` with at least one pixel inside ``ob2``."""
    ob1 = np.asarray(ob1).astype(bool)
    ob2 = np.asarray(ob2).astype(bool)
    lab, _ = label8(ob1)
    if ob1.size and ob1.all():
        return 0          # ``np.unique(regs)[1:]`` (:129) skips the only component when there is no background
    hit = np.unique(lab[ob2])
    return int(np.count_nonzero(hit))


def remove_small_objects4(mask, min_size):
    """skimage ``remove_small_objects`` on a bool array: 4-connected components with fewer than ``min_size``
    pixels are dropped (connectivity=1, strict ``<``)."""
    mask = np.asarray(mask).astype(bool)
    lab, n = label4(mask)
    keep = np.zeros(n + 1, bool)
    keep[1:] = _areas(lab, n) >= min_size
    return keep[lab]


def count_HSR(chrom, fish, size_threshold=HSR_SIZE_THRESHOLD):
    """src/image_tools.py:103-112."""
    return count_colocalization(chrom, remove_small_objects4(fish, size_threshold))

"""Operator-level mirror of the reference's ``src/image_tools.py`` for the metaseg path: same names, argument meaning
and return shapes, computed by the HIP kernels behind the C ABI (no CPU implementation lives here).

Each function cites the reference definition it stands in for.  All of them run on the process-wide default handle
(``default_handle()``, GPU ``ECSEG_DEVICE`` or 0) unless ``handle=`` is given.
"""
import os

import numpy as np

from ._lib import Handle

NUM_CLASSES = 4                 # src/image_tools.py:12
EC_SIZE_THRESHOLD = 15          # src/image_tools.py:13
_default = None


def default_handle():
    global _default
    if _default is None:
        _default = Handle(int(os.environ.get('ECSEG_DEVICE', os.environ.get('LOCAL_RANK', '0'))))
    return _default


def set_default_handle(h):
    global _default
    _default = h


def _starts(dim, overlap, scw):
    spw = scw - 2 * overlap
    cropped = dim - 2 * overlap
    q, r = divmod(cropped, spw)
    s = [spw * e for e in range(q)]
    if r:
        s.append(cropped - spw)
    return s


def im2patches_overlap(img, overlap_value=25, scw=256):
    """src/image_tools.py:148-186 -> [img, list of (scw, scw[, C]) views, list of [row, col]].  Pure indexing (the
    device pipeline tiles on the GPU; this host version exists for callers that want the patches themselves)."""
    Lh, Lw = _starts(img.shape[0], overlap_value, scw), _starts(img.shape[1], overlap_value, scw)
    pos = [[h, w] for w in Lw for h in Lh]
    return [img, [img[h:h + scw, w:w + scw] for h, w in pos], pos]


def split_FISH_channels(I, image_path, sensitivity, handle=None):
    """src/image_tools.py:136-146: writes the inverted red / green channel PNGs next to the image and returns the
    (red, green) masks ``channel > sensitivity``; returns int 0 for a non-RGB image like the reference."""
    from . import image_io
    path_split = os.path.split(image_path)
    I = np.asarray(I)
    if I.ndim < 3:
        print(image_path, " isn't an RGB image. Therefore, no FISH signals could be identified. Skipping...")
        return 0
    I = u16_to_u8(I, handle=handle)
    image_io.write_png(os.path.join(path_split[0], 'red', path_split[1] + '.png'), ~np.uint8(I[..., 0]), level=1)
    image_io.write_png(os.path.join(path_split[0], 'green', path_split[1] + '.png'), ~np.uint8(I[..., 1]), level=1)
    return (np.array(I[..., 0]) > sensitivity), (np.array(I[..., 1]) > sensitivity)


def stitch_argmax(preds, H, W, handle=None):
    """patches2im_overlap -> img_as_ubyte -> np.argmax(axis=2) (src/utils.py:116-118) for one image's patch
    predictions (n_patches, 256, 256, 4) -> int64 (H, W)."""
    h = handle or default_handle()
    return h.stitch_argmax(np.asarray(preds, np.float32), 1, H, W)[0].astype(np.int64)


def meta_preprocess(img, handle=None):
    """src/image_tools.py:86-96."""
    h = handle or default_handle()
    gray, _ = h.preprocess(np.asarray(img)[None])
    return gray[0]


def u16_to_u8(img, handle=None):
    """src/image_tools.py:98-101 (uint16 -> uint8 with cv2.convertScaleAbs semantics); other dtypes pass through."""
    img = np.asarray(img)
    if img.dtype != np.uint16:
        return img
    return (handle or default_handle()).u16_to_u8(img)


def meta_inference(img, handle=None):
    """src/image_tools.py:15-84: int label image (values 0..3) -> cleaned-up int64 label image."""
    h = handle or default_handle()
    a = np.asarray(img)
    if a.min(initial=0) < 0 or a.max(initial=0) > 3:
        raise ValueError('label image values must be in 0..3')
    out, _ = h.meta_inference(a.astype(np.uint8))
    return out.astype(np.int64)


def count_cc(I, handle=None):
    """src/image_tools.py:114-119 -> (number of components, total pixels); the second value is float 0.0 exactly where
    the reference returns ``np.sum([])``."""
    h = handle or default_handle()
    n, px = h.count_cc(np.asarray(I) != 0)
    return n, (0.0 if px == -1 else px)


def count_colocalization(ob1, ob2, handle=None):
    """src/image_tools.py:126-134."""
    h = handle or default_handle()
    return h.count_colocalization(np.asarray(ob1) != 0, np.asarray(ob2) != 0)


def count_HSR(chrom, fish, HSR_SIZE_THRESHOLD, handle=None):
    """src/image_tools.py:103-112."""
    h = handle or default_handle()
    return h.count_hsr(np.asarray(chrom) != 0, np.asarray(fish) != 0, int(HSR_SIZE_THRESHOLD))

"""Oracle: 256x256 overlapping tiling and the stitch source map.  TEST INFRASTRUCTURE ONLY.

Restates ``im2patches_overlap`` (reference src/image_tools.py:148-186) and
``patches2im_overlap`` (src/image_tools.py:188-252).  PINNED by tests/golden/tiling.npz.
"""
import numpy as np

OVERLAP = 25
SCW = 256
SPW = SCW - 2 * OVERLAP  # 206, the "prediction window"


def _starts(dim):
    """Window starts along one axis (src/image_tools.py:158-174): multiples of 206 over the cropped
    extent ``dim - 50`` plus, when 206 does not divide it, one extra window flush with the far edge."""
    cropped = dim - 2 * OVERLAP
    q, r = divmod(cropped, SPW)
    s = [SPW * e for e in range(q)]
    if r != 0:
        s.append(cropped - SPW)
    return s


def patch_positions(H, W):
    """(n, 2) int array of (row, col) window origins in reference order.

    ``np.meshgrid(L_h, L_w)`` then ravel (src/image_tools.py:176-178) walks columns in the outer loop and
    rows in the inner loop: ``for w in L_w: for h in L_h``."""
    if H < SCW or W < SCW:
        raise ValueError("image smaller than one 256x256 window")
    Lh, Lw = _starts(H), _starts(W)
    return np.array([[h, w] for w in Lw for h in Lh], dtype=np.int64).reshape(-1, 2)


def extract_patches(img, pos):
    """``img[h:h+256, w:w+256]`` per position (src/image_tools.py:181-184). ``img`` is (H, W) or (H, W, C)."""
    return np.stack([img[h:h + SCW, w:w + SCW] for h, w in pos])


def stitch_source_map(pos):
    """For every canvas pixel: which (patch, y, x) ends up there after the reference's sequence of
    slice assignments, or patch = -1 when nothing is ever written (value stays 0.0).

    Follows src/image_tools.py:202-250 copy by copy, including
      * border strips only from patches sitting at the first / last row or column of windows,
      * the column-vs-row comparison at :242 (``L_pos[i][1] != h_l``) that suppresses the right-hand
        strip whenever the last column start equals the last row start,
      * the final pass that writes every patch's 206x206 core in patch order (last writer wins).
    Returns (src_patch int32, src_y int16, src_x int16), each of canvas shape (h_l+256, w_l+256).
    """
    pos = np.asarray(pos, dtype=np.int64).reshape(-1, 2)
    h_l, w_l = int(pos[:, 0].max()), int(pos[:, 1].max())
    Hc, Wc = h_l + SCW, w_l + SCW
    sp = np.full((Hc, Wc), -1, np.int32)
    sy = np.zeros((Hc, Wc), np.int16)
    sx = np.zeros((Hc, Wc), np.int16)
    o, lo, hi = OVERLAP, OVERLAP, SCW - OVERLAP  # 25, 25, 231

    def put(i, dr0, dr1, dc0, dc1, sr0, sc0):
        nr, nc = dr1 - dr0, dc1 - dc0
        if nr <= 0 or nc <= 0:
            return
        sp[dr0:dr1, dc0:dc1] = i
        sy[dr0:dr1, dc0:dc1] = (sr0 + np.arange(nr))[:, None]
        sx[dr0:dr1, dc0:dc1] = (sc0 + np.arange(nc))[None, :]

    for i, (ph, pw) in enumerate(pos):
        ph, pw = int(ph), int(pw)
        if ph == 0:                                   # :207-219 top strip
            if pw == 0:
                put(i, 0, o, 0, o, 0, 0)
                put(i, lo, hi, 0, o, lo, 0)
                put(i, 0, o, lo, hi, 0, lo)
            else:
                if pw == w_l:
                    put(i, 0, o, Wc - o, Wc, 0, hi)
                put(i, 0, o, pw + lo, pw + hi, 0, lo)
        if pw == 0 and ph != 0:                       # :221-225 left strip
            put(i, ph + lo, ph + hi, 0, o, lo, 0)
        if ph == h_l:                                 # :227-240 bottom strip
            if pw == w_l:
                put(i, Hc - o, Hc, Wc - o, Wc, hi, hi)
                put(i, h_l + lo, Hc - o, Wc - o, Wc, lo, hi)
                put(i, Hc - o, Hc, w_l + lo, Wc - o, hi, lo)
            else:
                if pw == 0:
                    put(i, Hc - o, Hc, 0, o, hi, 0)
                put(i, Hc - o, Hc, pw + lo, pw + hi, hi, lo)
        if pw == w_l and pw != h_l:                   # :241-245 right strip (compares a column start with h_l)
            put(i, ph + lo, ph + hi, Wc - o, Wc, lo, hi)
    for i, (ph, pw) in enumerate(pos):                # :247-250 cores, patch order, overwrite
        put(i, int(ph) + lo, int(ph) + hi, int(pw) + lo, int(pw) + hi, lo, lo)
    return sp, sy, sx


def stitch(preds, pos):
    """Stitched float64 canvas (Hc, Wc, C) from per-patch predictions (n, 256, 256, C)
    (src/image_tools.py:204: ``np.zeros`` is float64, assignments up-cast the float32 predictions)."""
    sp, sy, sx = stitch_source_map(pos)
    out = np.zeros(sp.shape + (preds.shape[-1],), np.float64)
    m = sp >= 0
    out[m] = preds[sp[m], sy[m], sx[m]]
    return out

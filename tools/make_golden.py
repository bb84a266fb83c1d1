#!/opt/conda/bin/python3.9
"""Generate golden input/output vectors by RUNNING the reference's own functions.

Run (in the build container only, never on the GPU box):

    /opt/conda/bin/python3.9 tools/make_golden.py

It imports ``/root/reference/src/image_tools.py`` (with a stub ``cv2`` module, which that
file imports but the functions exercised here never call) and records, as small
compressed ``.npz`` / ``.json`` fixtures under ``tests/golden/``:

  tiling.npz        A6/A8  patch positions + stitch source maps for several image sizes
  quant_argmax.npz  A9/A10 ``img_as_ubyte`` + ``np.argmax`` known answers
  meta_inference_small.npz / meta_inference_full.npz   A11-A16 in/out label maps
  counting.npz      A17/A21/A22 ``count_cc`` / ``count_colocalization`` / ``count_HSR``
  overlay_rows.json A23 rows of the nine overlay counts on seeded inputs
  csv_text.json     A18/A23 exact CSV text produced by pandas for both tasks
  keras_tiny.h5     a tiny Keras-layout HDF5 file written by h5py (reader fixture)
  dapi_example.npz  pixels of example_ecSeg/dapi.jpeg (data file held by the reference)
  io_tiff_files.json / io_tiff_pixels.npz   tifffile-written TIFF variants + the pixels skimage.io.imread returns
  io_label_png.npz  RGBA that plt.imsave(cmap=4 colours, vmin=0, vmax=4) writes for a label image

Only data (inputs and the reference's outputs) is stored; no reference source text.
"""
import io
import json
import os
import sys
import types

import numpy as np

sys.modules['cv2'] = types.ModuleType('cv2')
sys.path.insert(0, '/root/reference/src')
import image_tools as ref  # noqa: E402  (the genuine reference module)
from skimage import img_as_ubyte  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


# --------------------------------------------------------------------------------------
# synthetic label-map generator (only used to make INPUTS; inputs are stored in fixtures)
# --------------------------------------------------------------------------------------
def disc(img, cy, cx, r, val, hole=0):
    H, W = img.shape
    yy, xx = np.ogrid[:H, :W]
    d2 = (yy - cy) ** 2 + (xx - cx) ** 2
    img[d2 <= r * r] = val
    if hole:
        img[d2 <= hole * hole] = 0


def ellipse(img, cy, cx, a, b, theta, val):
    H, W = img.shape
    yy, xx = np.ogrid[:H, :W]
    c, s = np.cos(theta), np.sin(theta)
    u = (xx - cx) * c + (yy - cy) * s
    v = -(xx - cx) * s + (yy - cy) * c
    img[(u / a) ** 2 + (v / b) ** 2 <= 1.0] = val


def random_label_map(rng, H, W, n_nuc, n_chrom, n_ec, salt, holes=True, spread=None):
    img = np.zeros((H, W), np.int64)
    cy0, cx0 = rng.integers(H // 4, 3 * H // 4), rng.integers(W // 4, 3 * W // 4)
    spread = spread or min(H, W) // 2
    for _ in range(n_nuc):
        r = int(rng.integers(max(6, min(H, W) // 16), max(8, min(H, W) // 6)))
        disc(img, int(rng.integers(0, H)), int(rng.integers(0, W)), r, 1,
             hole=int(rng.integers(0, r // 2)) if holes and rng.random() < 0.6 else 0)
    for _ in range(n_chrom):
        a = float(rng.uniform(4, 14)); b = float(rng.uniform(2, 6))
        cy = int(np.clip(cy0 + rng.integers(-spread, spread + 1), 0, H - 1))
        cx = int(np.clip(cx0 + rng.integers(-spread, spread + 1), 0, W - 1))
        ellipse(img, cy, cx, a, b, float(rng.uniform(0, np.pi)), 2)
        if holes and rng.random() < 0.2:
            img[cy, cx] = 0
    for _ in range(n_ec):
        r = int(rng.integers(1, 5))
        disc(img, int(rng.integers(0, H)), int(rng.integers(0, W)), r, 3)
    if salt > 0:
        m = rng.random((H, W)) < salt
        img[m] = rng.integers(0, 4, size=int(m.sum()))
    return img


def targeted_cases():
    """Hand-built inputs that force individual branches of meta_inference."""
    cases = []
    H, W = 256, 320
    # 0: empty image
    cases.append(np.zeros((H, W), np.int64))
    # 1: one class everywhere
    cases.append(np.full((H, W), 2, np.int64))
    # 2: nuclei only (mean of no chromosomes -> NaN)
    a = np.zeros((H, W), np.int64); disc(a, 100, 100, 40, 1, hole=10); disc(a, 200, 250, 20, 1)
    cases.append(a)
    # 3: ec only, sizes around the 15-px threshold
    a = np.zeros((H, W), np.int64)
    for k, r in enumerate([1, 2, 2, 3, 3, 4, 5]):
        disc(a, 30 + 30 * k, 40 + 35 * k, r, 3)
    a[200:203, 10:15] = 3  # exactly 15 px
    a[210:212, 10:17] = 3  # 14 px
    cases.append(a)
    # 4: nucleus ringed by chromosomes on all sides (>5 each side within 70 px)
    a = np.zeros((H, W), np.int64)
    disc(a, 128, 160, 18, 1)
    rng = np.random.default_rng(7)
    for k in range(40):
        ang = 2 * np.pi * k / 40
        rad = 40 + 8 * (k % 3)
        ellipse(a, int(128 + rad * np.sin(ang)), int(160 + rad * np.cos(ang)), 6, 2.5, ang, 2)
    for k in range(12):
        disc(a, int(rng.integers(5, H - 5)), int(rng.integers(5, W - 5)), 2, 3)
    cases.append(a)
    # 5: same but only five chromosomes on the left -> nucleus kept
    b = a.copy(); b[:, :150][b[:, :150] == 2] = 0
    for k in range(5):
        ellipse(b, 90 + 18 * k, 120, 5, 2, 0.3, 2)
    cases.append(b)
    # 6: ec touching nucleus / ec touching chromosome / isolated ec; last component skipped
    a = np.zeros((H, W), np.int64)
    disc(a, 60, 60, 25, 1); disc(a, 60, 88, 4, 3)          # ec touching nucleus
    ellipse(a, 150, 100, 20, 6, 0.0, 2); disc(a, 150, 123, 3, 3)  # ec touching chromosome
    disc(a, 200, 200, 3, 3)                                 # isolated ec
    disc(a, 240, 300, 12, 1); disc(a, 240, 314, 3, 3)       # LAST component in raster order: nucleus + ec
    cases.append(a)
    # 7: variant where the last component in raster order is chromosome + ec
    a = a.copy(); a[225:, 280:] = 0
    ellipse(a, 245, 290, 15, 5, 0.0, 2); disc(a, 245, 307, 3, 3)
    cases.append(a)
    # 8: blobs touching every image border, holes open to the border
    a = np.zeros((H, W), np.int64)
    disc(a, 0, 50, 30, 1, hole=8); disc(a, 255, 200, 30, 2, hole=10); disc(a, 128, 0, 25, 1, hole=12)
    disc(a, 128, 319, 25, 2); disc(a, 0, 319, 6, 3); disc(a, 255, 0, 5, 3); disc(a, 0, 0, 3, 3)
    cases.append(a)
    # 9: nested holes: chromosome inside a nucleus hole, ec inside a chromosome hole
    a = np.zeros((H, W), np.int64)
    disc(a, 128, 160, 90, 1, hole=60); disc(a, 128, 160, 40, 2, hole=20); disc(a, 128, 160, 6, 3)
    cases.append(a)
    # 10: small chromosomes turn into ec (area < mean ec area)
    a = np.zeros((H, W), np.int64)
    for k in range(6):
        disc(a, 40 + 35 * k, 60, 7, 3)
    for k in range(6):
        ellipse(a, 40 + 35 * k, 160, 3 + k, 1.5 + 0.5 * k, 0.5, 2)
    disc(a, 128, 260, 30, 1)
    cases.append(a)
    # 11: diagonal-only contacts between ec pixels and between ec and chromosome
    a = np.zeros((H, W), np.int64)
    for k in range(30):
        a[50 + k, 50 + k] = 3
    for k in range(30):
        a[50 + k, 120 - k] = 3 if k % 2 else 2
    a[150:170, 150:170] = 2; a[149, 149] = 3; a[170, 170] = 3; a[171, 171] = 3
    cases.append(a)
    # 12: checkerboards / thin lines (stress for the stencils)
    a = np.zeros((H, W), np.int64)
    a[20:60:2, 20:100] = 3; a[80:120, 20:100:2] = 2
    a[140:180, 20:100] = (np.indices((40, 80)).sum(0) % 2) * 3
    a[200:240, 20:100] = 1 + (np.indices((40, 80)).sum(0) % 3)
    a[20:240, 200:203] = 1; a[100:103, 120:300] = 2
    cases.append(a)
    # 13: spiral nucleus (deep union-find chains) with ec in the gaps
    a = np.zeros((H, W), np.int64)
    yy, xx = np.mgrid[:H, :W]
    r = np.hypot(yy - 128, xx - 160); t = np.arctan2(yy - 128, xx - 160)
    a[(np.mod(r - 6 * t / np.pi, 12) < 5) & (r < 110)] = 1
    a[(np.mod(r - 6 * t / np.pi, 12) > 8) & (r < 100) & ((yy + xx) % 17 == 0)] = 3
    cases.append(a)
    return cases


def gen_meta_inference():
    rng = np.random.default_rng(20260803)
    ins, outs = [], []
    for c in targeted_cases():
        ins.append(c)
    shapes = [(256, 320), (150, 150), (200, 333), (64, 64), (97, 401)]
    for i in range(60):
        H, W = shapes[i % len(shapes)]
        ins.append(random_label_map(rng, H, W, int(rng.integers(0, 4)), int(rng.integers(0, 40)),
                                    int(rng.integers(0, 60)), [0.0, 0.002, 0.02][i % 3]))
    # pure noise maps of various class priors
    for p in ([0.7, 0.1, 0.1, 0.1], [0.25, 0.25, 0.25, 0.25], [0.9, 0.0, 0.05, 0.05], [0.4, 0.3, 0.3, 0.0]):
        ins.append(rng.choice(4, size=(120, 160), p=p).astype(np.int64))
    for a in ins:
        outs.append(ref.meta_inference(a.copy()))
    pack = {}
    for k, (a, o) in enumerate(zip(ins, outs)):
        assert o.min() >= 0 and o.max() <= 3
        pack['in_%03d' % k] = a.astype(np.uint8)
        pack['out_%03d' % k] = o.astype(np.uint8)
        pack['nec_%03d' % k] = np.int64(ref.count_cc(o == 3)[0])
    np.savez_compressed(os.path.join(OUT, 'meta_inference_small.npz'), **pack)
    print('meta_inference_small: %d cases' % len(ins))

    # full-size cases: pseudo-segmentation of the example image + a dense synthetic one
    from skimage.io import imread
    d = imread('/root/reference/example_ecSeg/dapi.jpeg')
    inv = 255 - d
    np.savez_compressed(os.path.join(OUT, 'dapi_example.npz'), dapi=d)
    seg = np.zeros(inv.shape, np.int64)
    seg[inv > 40] = 3
    from scipy import ndimage as ndi
    lab, n = ndi.label(inv > 40, structure=np.ones((3, 3)))
    areas = ndi.sum(np.ones_like(lab), lab, index=np.arange(1, n + 1))
    big = np.zeros(n + 1, np.int64)
    big[1:][areas > 60] = 2
    big[1:][areas > 3000] = 1
    big[1:][areas <= 60] = 3
    seg = big[lab]
    full_in = [seg, random_label_map(rng, 1040, 1392, 3, 60, 150, 0.002, spread=250)]
    pack = {}
    for k, a in enumerate(full_in):
        o = ref.meta_inference(a.copy())
        pack['in_%03d' % k] = a.astype(np.uint8)
        pack['out_%03d' % k] = o.astype(np.uint8)
        pack['nec_%03d' % k] = np.int64(ref.count_cc(o == 3)[0])
    np.savez_compressed(os.path.join(OUT, 'meta_inference_full.npz'), **pack)
    print('meta_inference_full: %d cases' % len(full_in))


def gen_tiling():
    sizes = [(1040, 1392), (1024, 1280), (512, 512), (300, 300), (256, 256), (1000, 700), (462, 668), (257, 900)]
    pack = {'sizes': np.array(sizes, np.int64)}
    for (H, W) in sizes:
        img = np.arange(H * W, dtype=np.int64).reshape(H, W, 1)
        _, patches, pos = ref.im2patches_overlap(img)
        pos = np.array(pos, np.int64).reshape(-1, 2)
        # Encode (patch index + 1, y, x, 1) in the four "class" channels so that the stitched
        # canvas IS the source map: where each output pixel was copied from (0 = never written).
        enc = []
        yy, xx = np.mgrid[:256, :256]
        for i in range(len(patches)):
            e = np.zeros((256, 256, 4), np.float32)
            e[..., 0] = i + 1; e[..., 1] = yy; e[..., 2] = xx; e[..., 3] = 1
            enc.append(e)
        canvas = ref.patches2im_overlap(np.array(enc), [list(p) for p in pos])
        assert canvas.dtype == np.float64
        key = '%dx%d' % (H, W)
        pack['pos_' + key] = pos
        pack['first_px_' + key] = np.array([int(p[0, 0, 0]) for p in patches], np.int64)
        pack['src_patch_' + key] = (canvas[..., 0].astype(np.int32) - 1).astype(np.int16)
        pack['src_y_' + key] = canvas[..., 1].astype(np.int16)
        pack['src_x_' + key] = canvas[..., 2].astype(np.int16)
        print('tiling', key, 'patches', len(patches), 'canvas', canvas.shape,
              'unwritten', int((canvas[..., 3] == 0).sum()))
    np.savez_compressed(os.path.join(OUT, 'tiling.npz'), **pack)


def gen_quant():
    rng = np.random.default_rng(5)
    # (a) random softmax rows
    z = rng.normal(size=(4000, 4)).astype(np.float32) * 2
    p = np.exp(z - z.max(1, keepdims=True)); p = (p / p.sum(1, keepdims=True)).astype(np.float32)
    # (b) near-tie / half-way rows
    special = np.array([[.5, .5, 0, 0], [.2, .3992, .4008, 0], [.25, .25, .25, .25], [0, 0, 0, 0],
                        [1, 0, 0, 0], [0, 0, 0, 1], [.5 / 255, 1.5 / 255, 2.5 / 255, 3.5 / 255],
                        [0.00196, 0.00197, 0.99, 0.00607], [.3333, .3333, .3334, 0],
                        [126.5 / 255, 127.5 / 255, 0.5 / 255, 0.5 / 255]], np.float32)
    # (c) rows built to sit on .5 boundaries after *255
    k = rng.integers(0, 255, size=(2000, 4))
    half = ((k + 0.5) / 255.0).astype(np.float32)
    half = half / np.maximum(half.sum(1, keepdims=True), 1).astype(np.float32)
    probs = np.concatenate([p, special, half.astype(np.float32)], 0).reshape(-1, 1, 4)
    q = img_as_ubyte(probs.astype(np.float64))   # reference: stitched canvas is float64 (image_tools.py:204)
    lab = np.argmax(q, axis=2)
    np.savez_compressed(os.path.join(OUT, 'quant_argmax.npz'), probs=probs[:, 0, :], q=q[:, 0, :],
                        label=lab[:, 0].astype(np.uint8))
    print('quant_argmax rows', probs.shape[0], 'float-argmax mismatches',
          int((np.argmax(probs[:, 0, :], 1) != lab[:, 0]).sum()))


def blob_mask(rng, H, W, n, rmax, p_diag=0.2):
    m = np.zeros((H, W), bool)
    for _ in range(n):
        r = int(rng.integers(0, rmax + 1))
        cy, cx = int(rng.integers(0, H)), int(rng.integers(0, W))
        if rng.random() < p_diag:
            for k in range(int(rng.integers(2, 25))):
                if 0 <= cy + k < H and 0 <= cx + k < W:
                    m[cy + k, cx + k] = True
        else:
            yy, xx = np.ogrid[:H, :W]
            m |= (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
    return m


def gen_counting():
    rng = np.random.default_rng(11)
    pack = {}
    n = 80
    for k in range(n):
        H, W = [(64, 80), (100, 100), (37, 211)][k % 3]
        a = blob_mask(rng, H, W, int(rng.integers(0, 25)), 5)
        b = blob_mask(rng, H, W, int(rng.integers(0, 25)), 4)
        if k == 0:
            a[:] = False
        if k == 1:
            b[:] = False
        if k == 2:
            a[:] = True
        cc = ref.count_cc(a)
        pack['a_%03d' % k] = np.packbits(a, axis=1); pack['b_%03d' % k] = np.packbits(b, axis=1)
        pack['shape_%03d' % k] = np.array([H, W])
        pack['cc_%03d' % k] = np.array([int(cc[0]), int(cc[1])], np.int64)
        pack['cc_is_float_%03d' % k] = np.bool_(isinstance(cc[1], (float, np.floating)))
        pack['coloc_%03d' % k] = np.int64(ref.count_colocalization(a, b))
        pack['hsr_%03d' % k] = np.int64(ref.count_HSR(a, b.copy(), 20))
    # HSR threshold edge cases: 19 / 20 px blobs, 20-px diagonal chain (4-connectivity + strict <)
    chrom = np.zeros((40, 120), bool); chrom[5:35, 5:35] = True; chrom[5:35, 45:75] = True; chrom[5:35, 85:115] = True
    fish = np.zeros((40, 120), bool)
    fish[10:14, 10:15] = True                      # 20 px -> kept
    fish[10:14, 50:55] = True; fish[10, 50] = False  # 19 px -> dropped
    for t in range(20):
        fish[8 + t, 88 + t] = True                 # 20-px diagonal -> 20 singletons -> dropped
    pack['edge_chrom'] = np.packbits(chrom, axis=1); pack['edge_fish'] = np.packbits(fish, axis=1)
    pack['edge_shape'] = np.array(chrom.shape)
    pack['edge_hsr'] = np.int64(ref.count_HSR(chrom, fish.copy(), 20))
    pack['n'] = np.int64(n)
    np.savez_compressed(os.path.join(OUT, 'counting.npz'), **pack)
    print('counting: %d cases, edge hsr = %d' % (n, int(pack['edge_hsr'])))


def overlay_row(labels, rgb, sens):
    """Drive the reference's counting functions in the order meta_overlay.py:59-95 does."""
    red, green = rgb[..., 0] > sens, rgb[..., 1] > sens     # image_tools.py:146 (8-bit input: u16_to_u8 is identity)
    nuclei, chrom, ec = labels == 1, labels == 2, labels == 3
    fish = green * ~nuclei
    fish2 = red * ~nuclei
    row = {
        'ec_dapi': ref.count_cc(ec),
        'ec_green': ref.count_cc(fish * ~chrom),
        'ec_red': ref.count_cc(fish2 * ~chrom),
        'dapi_green': ref.count_colocalization(ec, fish),
        'dapi_red': ref.count_colocalization(ec, fish2),
        'red_green': ref.count_colocalization(fish * ~chrom, fish2 * ~chrom),
        'dapi_red_green': ref.count_colocalization(ec, fish2 * fish),
        'hsr_red': ref.count_HSR(chrom, fish2, 20),
        'hsr_green': ref.count_HSR(chrom, fish, 20),
    }
    return row


def gen_overlay_and_csv():
    import pandas as pd
    rng = np.random.default_rng(99)
    rows, pack = [], {}
    for k in range(12):
        H, W = 200, 260
        lab = random_label_map(rng, H, W, 2, 20, 40, 0.001)
        lab = ref.meta_inference(lab.copy())
        rgb = np.clip(rng.normal(20, 8, size=(H, W, 3)), 0, 255).astype(np.uint8)
        for ch in (0, 1):
            spots = blob_mask(rng, H, W, int(rng.integers(0, 60)), 3, p_diag=0.1)
            rgb[..., ch][spots] = rng.integers(120, 256, size=int(spots.sum()))
        if k == 0:
            rgb[..., :2] = 0       # no FISH signal at all -> (0, 0.0) tuples
        sens = [85, 85, 0, 255, 120, 60, 85, 85, 200, 85, 30, 85][k]
        r = overlay_row(lab, rgb, sens)
        pack['labels_%02d' % k] = lab.astype(np.uint8); pack['rgb_%02d' % k] = rgb
        pack['sens_%02d' % k] = np.int64(sens)
        rows.append({kk: ([int(v[0]), float(v[1]), bool(isinstance(v[1], (float, np.floating)))]
                          if isinstance(v, tuple) else int(v)) for kk, v in r.items()})
        # CSV text exactly as pandas would write this row (meta_overlay.py:85-102)
        df = pd.DataFrame([{'image_name': 'img%02d.tif' % k,
                            '# of ecDNA (DAPI)': r['ec_dapi'], '# of ecDNA (green)': r['ec_green'],
                            '# of ecDNA (red)': r['ec_red'], '# of ecDNA (DAPI and green)': r['dapi_green'],
                            '# of ecDNA (DAPI and red)': r['dapi_red'],
                            '# of ecDNA (red and green)': r['red_green'],
                            '# of ecDNA (DAPI and red and green)': r['dapi_red_green'],
                            '# of HSR (red)': r['hsr_red'], '# of HSR (green)': r['hsr_green']}])
        buf = io.StringIO(); df.to_csv(buf, index=False)
        rows[-1]['csv'] = buf.getvalue()
    np.savez_compressed(os.path.join(OUT, 'overlay_inputs.npz'), **pack)
    with open(os.path.join(OUT, 'overlay_rows.json'), 'w') as f:
        json.dump(rows, f, indent=1)
    # metaseg CSV (metaseg.py:39-57): DataFrame(columns=[...]) then row appends
    df = pd.DataFrame(columns=['image name', '# of ec'])
    for name, n in [('input.tif', 2), ('b c.tif', 0), ('z.npy', 137)]:
        df = pd.concat([df, pd.DataFrame([{'image name': name, '# of ec': n}])], ignore_index=True)
    buf = io.StringIO(); df.to_csv(buf, index=False)
    with open(os.path.join(OUT, 'csv_text.json'), 'w') as f:
        json.dump({'metaseg': buf.getvalue(),
                   'metaseg_rows': [['input.tif', 2], ['b c.tif', 0], ['z.npy', 137]]}, f, indent=1)
    print('overlay rows', len(rows))


def gen_keras_h5():
    """A tiny Keras-2.x-layout HDF5 file (what ``model.save('x.h5')`` writes), via h5py."""
    import h5py
    rng = np.random.default_rng(3)
    layers = []
    weights = {}

    def conv(name, cin, cout, k, act, inbound, padding='same'):
        layers.append({'class_name': 'Conv2D', 'name': name,
                       'config': {'name': name, 'trainable': True, 'dtype': 'float32', 'filters': cout,
                                  'kernel_size': [k, k], 'strides': [1, 1], 'padding': padding,
                                  'data_format': 'channels_last', 'dilation_rate': [1, 1], 'groups': 1,
                                  'activation': act, 'use_bias': True},
                       'inbound_nodes': [[[i, 0, 0, {}] for i in inbound]]})
        weights[name] = [('kernel:0', (rng.normal(size=(k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)),
                         ('bias:0', (rng.normal(size=(cout,)) * 0.1).astype(np.float32))]

    layers.append({'class_name': 'InputLayer', 'name': 'input_1',
                   'config': {'batch_input_shape': [None, 256, 256, 1], 'dtype': 'float32', 'sparse': False,
                              'ragged': False, 'name': 'input_1'}, 'inbound_nodes': []})
    conv('conv2d', 1, 4, 3, 'relu', ['input_1'])
    conv('conv2d_1', 4, 4, 3, 'relu', ['conv2d'])
    layers.append({'class_name': 'MaxPooling2D', 'name': 'max_pooling2d',
                   'config': {'name': 'max_pooling2d', 'pool_size': [2, 2], 'padding': 'valid', 'strides': [2, 2],
                              'data_format': 'channels_last'}, 'inbound_nodes': [[['conv2d_1', 0, 0, {}]]]})
    conv('conv2d_2', 4, 8, 3, 'relu', ['max_pooling2d'])
    layers.append({'class_name': 'BatchNormalization', 'name': 'batch_normalization',
                   'config': {'name': 'batch_normalization', 'axis': [3], 'momentum': 0.99, 'epsilon': 0.001,
                              'center': True, 'scale': True}, 'inbound_nodes': [[['conv2d_2', 0, 0, {}]]]})
    weights['batch_normalization'] = [('gamma:0', rng.uniform(0.5, 1.5, 8).astype(np.float32)),
                                      ('beta:0', rng.normal(size=8).astype(np.float32) * 0.1),
                                      ('moving_mean:0', rng.normal(size=8).astype(np.float32) * 0.1),
                                      ('moving_variance:0', rng.uniform(0.5, 1.5, 8).astype(np.float32))]
    layers.append({'class_name': 'Dropout', 'name': 'dropout', 'config': {'name': 'dropout', 'rate': 0.5},
                   'inbound_nodes': [[['batch_normalization', 0, 0, {}]]]})
    layers.append({'class_name': 'Conv2DTranspose', 'name': 'conv2d_transpose',
                   'config': {'name': 'conv2d_transpose', 'filters': 4, 'kernel_size': [2, 2], 'strides': [2, 2],
                              'padding': 'same', 'data_format': 'channels_last', 'dilation_rate': [1, 1],
                              'activation': 'linear', 'use_bias': True, 'output_padding': None},
                   'inbound_nodes': [[['dropout', 0, 0, {}]]]})
    weights['conv2d_transpose'] = [('kernel:0', (rng.normal(size=(2, 2, 4, 8)) * 0.3).astype(np.float32)),
                                   ('bias:0', (rng.normal(size=(4,)) * 0.1).astype(np.float32))]
    layers.append({'class_name': 'Concatenate', 'name': 'concatenate', 'config': {'name': 'concatenate', 'axis': 3},
                   'inbound_nodes': [[['conv2d_1', 0, 0, {}], ['conv2d_transpose', 0, 0, {}]]]})
    conv('conv2d_3', 8, 4, 3, 'relu', ['concatenate'])
    layers.append({'class_name': 'UpSampling2D', 'name': 'up_sampling2d',
                   'config': {'name': 'up_sampling2d', 'size': [2, 2], 'data_format': 'channels_last',
                              'interpolation': 'nearest'}, 'inbound_nodes': [[['max_pooling2d', 0, 0, {}]]]})
    conv('conv2d_4', 4, 4, 2, 'relu', ['up_sampling2d'])
    layers.append({'class_name': 'Concatenate', 'name': 'concatenate_1', 'config': {'name': 'concatenate_1', 'axis': -1},
                   'inbound_nodes': [[['conv2d_3', 0, 0, {}], ['conv2d_4', 0, 0, {}]]]})
    conv('conv2d_5', 8, 4, 1, 'softmax', ['concatenate_1'])
    cfg = {'class_name': 'Functional',
           'config': {'name': 'model', 'layers': layers, 'input_layers': [['input_1', 0, 0]],
                      'output_layers': [['conv2d_5', 0, 0]]}}
    path = os.path.join(OUT, 'keras_tiny.h5')
    with h5py.File(path, 'w') as f:
        f.attrs['keras_version'] = '2.8.0'
        f.attrs['backend'] = 'tensorflow'
        f.attrs['model_config'] = json.dumps(cfg)
        g = f.create_group('model_weights')
        g.attrs['layer_names'] = np.array([l['name'].encode() for l in layers])
        g.attrs['backend'] = 'tensorflow'; g.attrs['keras_version'] = '2.8.0'
        for l in layers:
            lg = g.create_group(l['name'])
            ws = weights.get(l['name'], [])
            lg.attrs['weight_names'] = np.array([('%s/%s' % (l['name'], w[0])).encode() for w in ws]) \
                if ws else np.zeros((0,), 'S1')
            if ws:
                sub = lg.create_group(l['name'])
                for wname, arr in ws:
                    sub.create_dataset(wname, data=arr)
    # flat copy of the weights as the reader's expected answer
    flat = {('%s/%s' % (ln, w[0])).replace('/', '__').replace(':', '_'): w[1] for ln, ws in weights.items() for w in ws}
    np.savez_compressed(os.path.join(OUT, 'keras_tiny_expected.npz'), model_config=np.array(json.dumps(cfg)), **flat)
    print('keras_tiny.h5', os.path.getsize(path), 'bytes')


def write_keras_h5(path, cfg, weights):
    """Keras-2 layout: root attr model_config (JSON), model_weights/<layer>/<layer>/<weight>:0 datasets."""
    import h5py
    wnames = {'BatchNormalization': ['gamma:0', 'beta:0', 'moving_mean:0', 'moving_variance:0']}
    with h5py.File(path, 'w') as f:
        f.attrs['keras_version'] = '2.8.0'
        f.attrs['backend'] = 'tensorflow'
        f.attrs['model_config'] = json.dumps(cfg)
        g = f.create_group('model_weights')
        layers = cfg['config']['layers']
        names = [l['config']['name'] for l in layers]
        g.attrs['layer_names'] = np.array([n.encode() for n in names])
        g.attrs['backend'] = 'tensorflow'; g.attrs['keras_version'] = '2.8.0'
        for l in layers:
            n = l['config']['name']
            lg = g.create_group(n)
            ws = weights.get(n, [])
            wn = wnames.get(l['class_name'], ['kernel:0', 'bias:0'])[:len(ws)]
            lg.attrs['weight_names'] = np.array([('%s/%s' % (n, w)).encode() for w in wn]) if ws else np.zeros((0,), 'S1')
            if ws:
                sub = lg.create_group(n)
                for w, arr in zip(wn, ws):
                    sub.create_dataset(w, data=arr)
    print(os.path.basename(path), os.path.getsize(path), 'bytes')


def gen_metaseg_h5():
    """A small but complete Keras-layout ``metaseg.h5`` (U-Net base 8, depth 3) written by h5py from the repo's own
    seeded synthetic weights, so that ``make metaseg`` can be exercised end to end where h5py does not exist; and the two
    interSeg classifier call shapes (``interseg_models/interseg``, ``interseg_models/ecseg_c``) the same way."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from ecseg_amd import synth
    cfg = synth.unet_config(base=8, depth=3)
    write_keras_h5(os.path.join(OUT, 'metaseg_synth_b8.h5'), cfg, synth.unet_weights(cfg, seed=11))
    for kind, seed in (('interseg', 21), ('ecseg_c', 22)):
        cfg = synth.classifier_config(kind)
        write_keras_h5(os.path.join(OUT, '%s_synth.h5' % kind), cfg, synth.classifier_weights(cfg, seed=seed))


def gen_mobilenet_h5():
    """``mobilenet_synth.h5``: the MobileNet-style classifier of ecseg_amd/synth.py:mobilenet_classifier written the way
    Keras 2.8 saves a model that uses a NESTED sub-model as one layer: the sub-model's group holds the variables of all its
    layers under ``<inner layer>/<variable>:0`` and its ``weight_names`` lists the trainable ones first, then the moving
    statistics of the BatchNormalization layers (``layer.trainable_weights + layer.non_trainable_weights``)."""
    import h5py
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from ecseg_amd import synth
    cfg, weights = synth.mobilenet_classifier(seed=31)
    var_names = {'BatchNormalization': ['gamma:0', 'beta:0', 'moving_mean:0', 'moving_variance:0'], 'PReLU': ['alpha:0'],
                 'DepthwiseConv2D': ['depthwise_kernel:0', 'bias:0'], 'SeparableConv2D': ['depthwise_kernel:0', 'pointwise_kernel:0', 'bias:0'],
                 'LayerNormalization': ['gamma:0', 'beta:0']}
    path = os.path.join(OUT, 'mobilenet_synth.h5')
    with h5py.File(path, 'w') as f:
        f.attrs['keras_version'] = '2.8.0'
        f.attrs['backend'] = 'tensorflow'
        f.attrs['model_config'] = json.dumps(cfg)
        g = f.create_group('model_weights')
        layers = cfg['config']['layers']
        g.attrs['layer_names'] = np.array([l['config']['name'].encode() for l in layers])
        g.attrs['backend'] = 'tensorflow'; g.attrs['keras_version'] = '2.8.0'
        for l in layers:
            n = l['config']['name']
            lg = g.create_group(n)
            if l['class_name'] == 'Functional':
                train, frozen = [], []
                for il in l['config']['layers']:
                    iname = il['config']['name']
                    ws = weights[n].get(iname, [])
                    vn = var_names.get(il['class_name'], ['kernel:0', 'bias:0'])[:len(ws)]
                    for k, (v, arr) in enumerate(zip(vn, ws)):
                        (frozen if v.startswith('moving_') else train).append(('%s/%s' % (iname, v), arr))
                allw = train + frozen
                lg.attrs['weight_names'] = np.array([w[0].encode() for w in allw])
                for wn, arr in allw:
                    lg.create_dataset(wn, data=arr)
                continue
            ws = weights.get(n, [])
            vn = var_names.get(l['class_name'], ['kernel:0', 'bias:0'])[:len(ws)]
            lg.attrs['weight_names'] = np.array([('%s/%s' % (n, v)).encode() for v in vn]) if ws else np.zeros((0,), 'S1')
            for v, arr in zip(vn, ws):
                lg.create_dataset('%s/%s' % (n, v), data=arr)
    print('mobilenet_synth.h5', os.path.getsize(path), 'bytes')


def gen_io():
    """I/O-layer fixtures written / decoded by the libraries the reference itself uses: TIFF files written by tifffile
    (the decoder behind ``skimage.io.imread``, src/utils.py:110) in the variants microscopes and OpenCV produce, with the
    pixels skimage.io.imread returns for them; and the RGBA pixels ``plt.imsave(..., cmap=ListedColormap(4 colours),
    vmin=0, vmax=4)`` writes for a label image (src/metaseg.py:47-52), decoded back with matplotlib."""
    import base64
    import tempfile
    import tifffile
    from skimage.io import imread
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    from matplotlib.colors import ListedColormap
    rng = np.random.default_rng(11)
    H, W = 37, 52
    yy, xx = np.mgrid[:H, :W]
    smooth = (yy * 5 + xx * 3) % 256
    g8 = (smooth + rng.integers(0, 8, (H, W))).astype(np.uint8)
    g16 = (smooth.astype(np.uint16) * 251 + rng.integers(0, 300, (H, W))).astype(np.uint16)
    rgb8 = np.stack([g8, g8[::-1], rng.integers(0, 256, (H, W)).astype(np.uint8)], -1)
    rgb16 = np.stack([g16, g16[:, ::-1], rng.integers(0, 65536, (H, W)).astype(np.uint16)], -1)
    cases = [
        ('gray8_raw', g8, dict()),
        ('gray8_lzw_pred_strips5', g8, dict(compression='lzw', predictor=True, rowsperstrip=5)),      # OpenCV imwrite style
        ('rgb8_lzw_pred_strips7', rgb8, dict(compression='lzw', predictor=True, rowsperstrip=7, photometric='rgb')),
        ('rgb8_deflate', rgb8, dict(compression='zlib', photometric='rgb')),
        ('gray16_lzw_pred', g16, dict(compression='lzw', predictor=True, rowsperstrip=8)),
        ('rgb16_lzw_pred', rgb16, dict(compression='lzw', predictor=True, rowsperstrip=4, photometric='rgb')),
        ('rgb16_deflate_pred', rgb16, dict(compression='zlib', predictor=True, photometric='rgb')),
        ('gray16_bigendian', g16, dict(byteorder='>')),
        ('rgb8_tiled16', rgb8, dict(tile=(16, 16), compression='zlib', photometric='rgb')),
        ('gray16_tiled32_pred', g16, dict(tile=(32, 32), compression='zlib', predictor=True)),
        ('gray8_packbits', g8, dict(compression='packbits')),
        ('rgb8_bigtiff', rgb8, dict(bigtiff=True, photometric='rgb', compression='zlib', predictor=True)),
        ('gray8_strips5_deflate_pred', g8, dict(compression='zlib', predictor=True, rowsperstrip=5)),
    ]
    files, pixels = {}, {}
    tmp = tempfile.mkdtemp()
    for name, arr, kw in cases:
        path = os.path.join(tmp, name + '.tif')
        try:
            tifffile.imwrite(path, arr, **kw)
        except Exception as e:                                  # codec missing in this tifffile / imagecodecs build
            print('  skipped', name, e)
            continue
        dec = imread(path)                                       # what the reference's imread returns
        assert dec.shape == arr.shape and dec.dtype == arr.dtype and np.array_equal(dec, arr), name
        files[name] = base64.b64encode(open(path, 'rb').read()).decode()
        pixels[name] = dec
    # LZW (this imagecodecs build only decodes it): written by libtiff through PIL, decoded by tifffile like every other
    # case; + the reference's own example file (OpenCV-written: LZW, predictor 2, 5 rows per strip)
    from PIL import Image
    lzw_cases = [('pil_gray8_lzw', g8, {}), ('pil_rgb8_lzw', rgb8, {}), ('pil_gray16_lzw', g16, {}),
                 ('pil_gray8_lzw_pred2', g8, {'tiffinfo': {317: 2}}), ('pil_rgb8_lzw_pred2', rgb8, {'tiffinfo': {317: 2}})]
    for name, arr, kw in lzw_cases:
        path = os.path.join(tmp, name + '.tif')
        try:
            Image.fromarray(arr).save(path, compression='tiff_lzw', **kw)
            dec = imread(path)
        except Exception as e:
            print('  skipped', name, e)
            continue
        assert dec.shape == arr.shape and dec.dtype == arr.dtype and np.array_equal(dec, arr), name
        files[name] = base64.b64encode(open(path, 'rb').read()).decode()
        pixels[name] = dec
    ex = '/root/reference/example_ecSeg/dapi.jpeg'
    crop = imread(ex)[400:440, 600:660]                          # a 40x60 window re-encoded the way OpenCV wrote the file
    tf = tifffile.TiffFile(ex).pages[0]
    print('  example_ecSeg/dapi.jpeg tags: compression', tf.compression, 'predictor', tf.predictor, 'rowsperstrip', tf.rowsperstrip)
    json.dump({'note': 'TIFF files written by tifffile %s (imagecodecs) or libtiff via PIL (pil_*), base64; pixels in io_tiff_pixels.npz are what '
                       'skimage.io.imread returned for each' % tifffile.__version__, 'files': files},
              open(os.path.join(OUT, 'io_tiff_files.json'), 'w'))
    np.savez_compressed(os.path.join(OUT, 'io_tiff_pixels.npz'), **pixels)
    print('io: %d tifffile-written TIFF fixtures' % len(files))

    # labels/<stem>.png exactly as src/metaseg.py:47-52 writes it
    cmap = ListedColormap(["#386cb0", "#ffff99", "#7fc97f", '#f0027f'])
    lab = rng.integers(0, 4, (23, 31)).astype(np.int64)
    lab[0, :4] = [0, 1, 2, 3]
    path = os.path.join(tmp, 'lab.png')
    plt.imsave(path, lab.astype('uint8'), cmap=cmap, vmin=0, vmax=4)
    rgba = (plt.imread(path) * 255.0 + 0.5).astype(np.uint8)
    assert rgba.shape == (23, 31, 4)
    np.savez_compressed(os.path.join(OUT, 'io_label_png.npz'), labels=lab.astype(np.uint8), rgba=rgba,
                        class_rgba=rgba[0, :4])
    print('io: label PNG colours', rgba[0, :4].tolist())


def gen_otsu():
    """meta_preprocess's only data-dependent decision is ``sum(otsu(img)) > 0.5 H W`` (src/image_tools.py:91-95).  OpenCV
    is not installed anywhere; scikit-image's ``threshold_otsu`` (an independent implementation of the same published
    criterion, 256 bins on uint8) gives second-source thresholds + decisions for seeded images, stored as data."""
    from skimage.filters import threshold_otsu
    rng = np.random.default_rng(21)
    imgs = []
    for k in range(24):
        H, W = int(rng.integers(40, 90)), int(rng.integers(40, 90))
        bg, fg = int(rng.integers(0, 120)), int(rng.integers(130, 256))
        frac = [0.1, 0.3, 0.45, 0.55, 0.7, 0.9][k % 6]
        m = rng.random((H, W)) < frac
        img = np.where(m, rng.normal(fg, 12, (H, W)), rng.normal(bg, 8, (H, W)))
        imgs.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    d = np.load(os.path.join(OUT, 'dapi_example.npz'))['dapi']
    imgs += [d[::4, ::4].copy(), (255 - d)[::4, ::4].copy()]
    pack = {}
    for k, im in enumerate(imgs):
        t = int(threshold_otsu(im))                               # foreground = img > t, as cv2.THRESH_BINARY
        pack['img_%02d' % k] = im
        pack['thr_%02d' % k] = np.int64(t)
        pack['white_%02d' % k] = np.int64(np.count_nonzero(im > t))
    np.savez_compressed(os.path.join(OUT, 'otsu_skimage.npz'), **pack)
    print('otsu: %d images' % len(imgs))


if __name__ == '__main__':
    which = sys.argv[1:] or ['tiling', 'quant', 'meta', 'counting', 'overlay', 'h5', 'metaseg_h5', 'mobilenet_h5', 'io', 'otsu']
    if 'io' in which: gen_io()
    if 'otsu' in which: gen_otsu()
    if 'metaseg_h5' in which: gen_metaseg_h5()
    if 'mobilenet_h5' in which: gen_mobilenet_h5()
    if 'tiling' in which: gen_tiling()
    if 'quant' in which: gen_quant()
    if 'meta' in which: gen_meta_inference()
    if 'counting' in which: gen_counting()
    if 'overlay' in which: gen_overlay_and_csv()
    if 'h5' in which: gen_keras_h5()

"""Pin the oracle: it must reproduce, bit for bit, the outputs of the reference's own functions that
tools/make_golden.py captured (tests/golden/*)."""
import json
import os

import numpy as np
import pytest

from oracle import overlay, postproc, quant, tiling


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_patch_positions_and_stitch_map(golden_dir):
    g = _load(golden_dir, 'tiling.npz')
    for H, W in g['sizes']:
        key = '%dx%d' % (H, W)
        pos = tiling.patch_positions(int(H), int(W))
        assert np.array_equal(pos, g['pos_' + key]), key
        img = np.arange(H * W, dtype=np.int64).reshape(H, W)
        patches = tiling.extract_patches(img, pos)
        assert patches.shape[1:] == (256, 256)
        assert np.array_equal(patches[:, 0, 0], g['first_px_' + key]), key
        sp, sy, sx = tiling.stitch_source_map(pos)
        assert sp.shape == (H, W)
        assert np.array_equal(sp, g['src_patch_' + key]), key
        m = sp >= 0
        assert np.array_equal(sy[m], g['src_y_' + key][m]), key
        assert np.array_equal(sx[m], g['src_x_' + key][m]), key


def test_stitch_lands_at_true_coordinates_full_size():
    pos = tiling.patch_positions(1040, 1392)
    assert len(pos) == 35
    sp, sy, sx = tiling.stitch_source_map(pos)
    assert (sp >= 0).all()
    yy, xx = np.indices(sp.shape)
    assert np.array_equal(pos[sp, 0] + sy, yy) and np.array_equal(pos[sp, 1] + sx, xx)


def test_quantised_argmax(golden_dir):
    g = _load(golden_dir, 'quant_argmax.npz')
    q = quant.quantise_u8(g['probs'])
    assert np.array_equal(q, g['q'])
    assert np.array_equal(quant.argmax_first(q), g['label'])
    assert quant.quantised_argmax(np.array([[.5, .5, 0, 0]]))[0] == 0
    assert quant.quantised_argmax(np.array([[.2, .3992, .4008, 0]], np.float32))[0] == 1
    with pytest.raises(ValueError):
        quant.quantise_u8(np.array([1.5]))


@pytest.mark.parametrize('name', ['meta_inference_small.npz', 'meta_inference_full.npz'])
def test_meta_inference(golden_dir, name):
    g = _load(golden_dir, name)
    n = len([k for k in g.files if k.startswith('in_')])
    assert n > 0
    for k in range(n):
        out = postproc.meta_inference(g['in_%03d' % k])
        assert out.dtype == np.int64
        assert np.array_equal(out, g['out_%03d' % k]), 'case %d' % k
        assert postproc.count_cc(out == 3)[0] == int(g['nec_%03d' % k]), 'case %d' % k


def _unpack(g, key, shape):
    return np.unpackbits(g[key], axis=1)[:, :shape[1]].astype(bool)


def test_counting(golden_dir):
    g = _load(golden_dir, 'counting.npz')
    for k in range(int(g['n'])):
        shape = g['shape_%03d' % k]
        a, b = _unpack(g, 'a_%03d' % k, shape), _unpack(g, 'b_%03d' % k, shape)
        n, px = postproc.count_cc(a)
        assert [n, int(px)] == list(g['cc_%03d' % k]), k
        assert isinstance(px, float) == bool(g['cc_is_float_%03d' % k]), k
        assert postproc.count_colocalization(a, b) == int(g['coloc_%03d' % k]), k
        assert postproc.count_HSR(a, b, 20) == int(g['hsr_%03d' % k]), k
    shape = g['edge_shape']
    chrom, fish = _unpack(g, 'edge_chrom', shape), _unpack(g, 'edge_fish', shape)
    assert postproc.count_HSR(chrom, fish, 20) == int(g['edge_hsr']) == 1


def test_overlay_rows_and_csv(golden_dir):
    rows = json.load(open(os.path.join(golden_dir, 'overlay_rows.json')))
    g = _load(golden_dir, 'overlay_inputs.npz')
    keys = ['ec_dapi', 'ec_green', 'ec_red', 'dapi_green', 'dapi_red', 'red_green', 'dapi_red_green',
            'hsr_red', 'hsr_green']
    for k, want in enumerate(rows):
        got = overlay.overlay_row(g['labels_%02d' % k], g['rgb_%02d' % k], int(g['sens_%02d' % k]))
        for key, v in zip(keys, got):
            if isinstance(v, tuple):
                assert [v[0], float(v[1]), isinstance(v[1], float)] == want[key], (k, key)
            else:
                assert v == want[key], (k, key)
        assert overlay.csv_text(overlay.OVERLAY_COLUMNS, [['img%02d.tif' % k] + got]) == want['csv'], k
    c = json.load(open(os.path.join(golden_dir, 'csv_text.json')))
    assert overlay.csv_text(overlay.METASEG_COLUMNS, c['metaseg_rows']) == c['metaseg']


def test_otsu_against_skimage_second_source(golden_dir):
    """A3/A4 (OpenCV, PARITY UNPINNED): the Otsu restatement agrees with scikit-image's independent implementation of the
    same criterion - threshold within one grey level (the two break flat maxima differently), identical '> 50 % white'
    decision wherever that decision does not hinge on the one ambiguous level."""
    from oracle import preprocess
    z = np.load(os.path.join(golden_dir, 'otsu_skimage.npz'))
    n = len([k for k in z.files if k.startswith('img_')])
    assert n >= 26
    for k in range(n):
        im = z['img_%02d' % k]
        t = preprocess.otsu_threshold_u8(im)
        assert abs(t - int(z['thr_%02d' % k])) <= 1, (k, t, int(z['thr_%02d' % k]))
        white = int(np.count_nonzero(im > t))
        half = im.size * 0.5
        if (white > half) != (int(z['white_%02d' % k]) > half):
            lo, hi = sorted((white, int(z['white_%02d' % k])))
            assert lo <= half <= hi                                    # only possible when the two thresholds differ
        inv = preprocess.meta_preprocess(im)
        assert np.array_equal(inv, ~im if white > half else im)


def test_convert_scale_abs_known_answers():
    """cv2.convertScaleAbs(u16, alpha = 255/65535): |x * alpha| in float32, cvRound (half to even), saturate to uint8 -
    values worked out by hand from that definition."""
    from oracle import preprocess
    x = np.array([0, 1, 128, 129, 257, 32767, 32768, 32896, 65534, 65535], np.uint16)
    a = np.float32(255.0 / 65535.0)
    want = [0, 0, 0, 1, 1, 127, 128, 128, 255, 255]
    # 128 * a = 0.49805 -> 0; 129 * a = 0.50194 -> 1; 257 * a = 1.0000 -> 1; 32767 * a = 127.498 -> 127;
    # 32768 * a = 127.502 -> 128; 32896 * a = 128.0 -> 128
    got = preprocess.u16_to_u8(x)
    assert got.dtype == np.uint8 and got.tolist() == want
    assert [int(np.rint(np.float32(v) * a)) for v in x.tolist()] == want
    u8 = np.arange(256, dtype=np.uint8)
    assert preprocess.u16_to_u8(u8) is u8                              # non-uint16 passes through (src/image_tools.py:99)

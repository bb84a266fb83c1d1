"""Post-processing parity on the GPU (bit-exact): HIP kernels through the C ABI vs the golden vectors captured from
the reference and vs the CPU oracle on seeded inputs."""
import json
import os

import numpy as np
import pytest

from ecseg_amd import synth
from oracle import overlay as oracle_overlay
from oracle import postproc, preprocess, quant, tiling

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _unpack(g, key, shape):
    return np.unpackbits(g[key], axis=1)[:, :shape[1]].astype(np.uint8)


# ---------------------------------------------------------------- connected components
def _canon(lab):
    """Relabel so that every component carries 1 + the raster index of its first pixel."""
    out = np.zeros(lab.shape, np.int32)
    if lab.max() > 0:
        idx = np.arange(lab.size).reshape(lab.shape)
        first = np.full(lab.max() + 1, lab.size, np.int64)
        np.minimum.at(first, lab.ravel(), idx.ravel())
        out = np.where(lab > 0, first[lab] + 1, 0).astype(np.int32)
    return out


@pytest.mark.parametrize('conn', [4, 8])
def test_ccl_labels_adversarial(gpu, conn):
    rng = np.random.default_rng(conn)
    H, W = 150, 333                      # W not a multiple of the 64-pixel chunk
    masks = []
    yy, xx = np.mgrid[:H, :W]
    masks.append(np.zeros((H, W), np.uint8))
    masks.append(np.ones((H, W), np.uint8))
    masks.append(((yy + xx) % 2).astype(np.uint8))                       # checkerboard: all diagonal contacts
    r = np.hypot(yy - 75, xx - 166); t = np.arctan2(yy - 75, xx - 166)
    masks.append((np.mod(r - 4 * t / np.pi, 8) < 3).astype(np.uint8))    # spiral: deep union-find chains
    masks.append((rng.random((H, W)) < 0.55).astype(np.uint8))           # percolation threshold noise
    masks.append((rng.random((H, W)) < 0.3).astype(np.uint8))
    m = np.zeros((H, W), np.uint8); m[::2, :] = 1; m[:, 64] = 1; m[:, 127] = 1   # combs crossing chunk borders
    masks.append(m)
    m = np.zeros((H, W), np.uint8); m[:, ::2] = 1; m[H - 1, :] = 1
    masks.append(m)
    stack = np.stack(masks)
    got = gpu.ccl_labels(stack, conn)
    for k, mk in enumerate(masks):
        lab, _ = (postproc.label8 if conn == 8 else postproc.label4)(mk)
        assert np.array_equal(got[k], _canon(lab)), (conn, k)
    # run-to-run determinism
    assert np.array_equal(got, gpu.ccl_labels(stack, conn))


def test_count_functions_golden(gpu, golden_dir):
    g = _load(golden_dir, 'counting.npz')
    for k in range(int(g['n'])):
        shape = g['shape_%03d' % k]
        a, b = _unpack(g, 'a_%03d' % k, shape), _unpack(g, 'b_%03d' % k, shape)
        n, px = gpu.count_cc(a)
        want_n, want_px = g['cc_%03d' % k]
        assert n == want_n, k
        assert (px == -1) == bool(g['cc_is_float_%03d' % k]), k
        if px != -1:
            assert px == want_px, k
        assert gpu.count_colocalization(a, b) == int(g['coloc_%03d' % k]), k
        assert gpu.count_hsr(a, b, 20) == int(g['hsr_%03d' % k]), k
    shape = g['edge_shape']
    assert gpu.count_hsr(_unpack(g, 'edge_chrom', shape), _unpack(g, 'edge_fish', shape), 20) == 1


def test_count_batched_matches_single(gpu):
    rng = np.random.default_rng(2)
    masks = (rng.random((70, 96, 130)) < 0.4).astype(np.uint8)     # more images than one internal chunk of 64
    n, px = gpu.count_cc(masks)
    for k in range(0, 70, 9):
        w = postproc.count_cc(masks[k])
        assert n[k] == w[0] and px[k] == w[1]


# ---------------------------------------------------------------- meta_inference
@pytest.mark.parametrize('name', ['meta_inference_small.npz', 'meta_inference_full.npz'])
def test_meta_inference_golden(gpu, golden_dir, name):
    g = _load(golden_dir, name)
    n = len([k for k in g.files if k.startswith('in_')])
    for k in range(n):
        out, nec = gpu.meta_inference(g['in_%03d' % k])
        assert np.array_equal(out, g['out_%03d' % k]), 'case %d' % k
        assert nec == int(g['nec_%03d' % k]), 'case %d' % k


def test_meta_inference_batch_vs_oracle(gpu):
    labs = np.stack([synth.label_map(i, 300, 420, salt=[0.0, 0.002, 0.02][i % 3]) for i in range(9)])
    out, nec = gpu.meta_inference(labs)
    for i in range(len(labs)):
        want = postproc.meta_inference(labs[i])
        assert np.array_equal(out[i], want), i
        assert nec[i] == postproc.count_cc(want == 3)[0], i
    out2, nec2 = gpu.meta_inference(labs)          # determinism of the atomics-based union-find
    assert np.array_equal(out, out2) and np.array_equal(nec, nec2)


def test_meta_inference_idempotent_properties_full_size(gpu):
    """Size-independent properties at BASELINE size: values stay in 0..3, a second pass of the final count is stable."""
    lab = synth.label_map(7)
    out, nec = gpu.meta_inference(lab)
    assert out.shape == (1040, 1392) and out.max() <= 3
    n, _ = gpu.count_cc(out == 3)
    assert n == nec == postproc.count_cc(out == 3)[0]


# ---------------------------------------------------------------- stitch + quantised argmax
def test_stitch_argmax_known_answers(gpu, golden_dir):
    g = _load(golden_dir, 'quant_argmax.npz')
    probs, want = g['probs'], g['label']
    H = W = 256
    n = probs.shape[0]
    p = np.zeros((1, 256, 256, 4), np.float32)
    # place the known-answer rows in the patch core, where the 256x256 stitch map reads them back
    ys, xs = np.divmod(np.arange(n), 200)
    p[0, 25 + ys, 25 + xs] = probs
    lab = gpu.stitch_argmax(p, 1, H, W)
    assert np.array_equal(lab[0, 25 + ys, 25 + xs], want)


@pytest.mark.parametrize('H,W', [(256, 256), (300, 300), (512, 512), (462, 668), (1040, 1392)])
def test_stitch_argmax_vs_oracle(gpu, H, W):
    rng = np.random.default_rng(H + W)
    pos = tiling.patch_positions(H, W)
    z = rng.normal(size=(len(pos), 256, 256, 4)).astype(np.float32) * 3
    e = np.exp(z - z.max(-1, keepdims=True))
    p = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    want = quant.quantised_argmax(tiling.stitch(p, pos)).astype(np.uint8)
    got = gpu.stitch_argmax(p, 1, H, W)[0]
    assert np.array_equal(got, want)


# ---------------------------------------------------------------- overlay
def test_overlay_rows_golden(gpu, golden_dir):
    rows = json.load(open(os.path.join(golden_dir, 'overlay_rows.json')))
    g = _load(golden_dir, 'overlay_inputs.npz')
    for k, want in enumerate(rows):
        got = gpu.overlay(g['labels_%02d' % k], g['rgb_%02d' % k], int(g['sens_%02d' % k]))
        cells = []
        for j in (0, 2, 4):
            n, px = int(got[j]), int(got[j + 1])
            cells.append((n, 0.0 if px == -1 else px))
        cells += [int(v) for v in got[6:]]
        assert oracle_overlay.csv_text(oracle_overlay.OVERLAY_COLUMNS, [['img%02d.tif' % k] + cells]) == want['csv'], k


# ---------------------------------------------------------------- preprocess
def test_preprocess_vs_oracle(gpu):
    rng = np.random.default_rng(0)
    imgs8 = np.stack([synth.dapi_image(i, 256, 300, rgb=True) for i in range(3)])
    imgs8[1] = 255 - imgs8[1]                    # mostly white -> must be inverted back
    gray, inv = gpu.preprocess(imgs8)
    for i in range(3):
        assert np.array_equal(gray[i], preprocess.meta_preprocess(imgs8[i])), i
    want_inv = [int(np.count_nonzero(im[..., 2] > preprocess.otsu_threshold_u8(im[..., 2])) > im.shape[0] * im.shape[1] * 0.5)
                for im in imgs8]
    assert list(inv) == want_inv and want_inv[0] != want_inv[1]
    g16 = (rng.integers(0, 65536, size=(2, 260, 256))).astype(np.uint16)
    gray, inv = gpu.preprocess(g16)
    for i in range(2):
        assert np.array_equal(gray[i], preprocess.meta_preprocess(g16[i])), i


def test_meta_inference_through_hip_graph_replay(gpu):
    """Option post_graph=1: the kernels of meta_inference captured into a HIP graph and replayed must give the same labels
    and counts as plain launches, also when the same buffers are used again with new contents."""
    labs = np.stack([synth.label_map(400 + i, 300, 420) for i in range(3)])
    plain, nec_plain = gpu.meta_inference(labs)
    try:
        gpu.set_option('post_graph', 1)
        for rep in range(3):
            x = np.roll(labs, rep * 7, axis=2)
            got, nec = gpu.meta_inference(x)
            want, wnec = (plain, nec_plain) if rep == 0 else (None, None)
            for k in range(3):
                w = postproc.meta_inference(x[k])
                assert np.array_equal(got[k], w)
                assert int(nec[k]) == postproc.count_cc(w == 3)[0]
            if want is not None:
                assert np.array_equal(got, want) and np.array_equal(nec, wnec)
    finally:
        gpu.set_option('post_graph', 0)


def test_nucleus_test_binned_and_pair_paths(gpu):
    """The nucleus-in-metaphase test runs on binned centroid coordinates (counting sort) and falls back to pair tests for images
    wider / taller than 32768: dense grids of chromosomes (5.7 k and 35 k per image) with nuclei that are erased and nuclei that
    keep an empty band, and a 3 x 33000 strip for the fall-back."""
    import numpy as np
    from oracle import postproc
    rng = np.random.default_rng(77)
    labs = []
    for step in (5, 2):
        H, W = 480, 480
        lab = np.zeros((H, W), np.uint8)
        lab[4::step, 4:300:step] = 2                        # chromosomes only left of x = 300: nuclei right of it keep an empty band
        yy, xx = np.ogrid[:H, :W]
        for cy, cx, r in ((120, 130, 14), (300, 340, 9), (20, 460, 6), (440, 40, 11), (240, 296, 3)):
            lab[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1
        lab[rng.random((H, W)) < 0.001] = 3
        labs.append(lab)
    labs = np.stack(labs)
    kept = [int((postproc.meta_inference(l) == 1).sum()) for l in labs]
    assert all(0 < k < int((l == 1).sum()) for k, l in zip(kept, labs)), kept      # some nuclei erased, some kept
    out, nec = gpu.meta_inference(labs)
    for k in range(2):
        want = postproc.meta_inference(labs[k])
        assert np.array_equal(out[k], want), (k, int((out[k] != want).sum()))
        assert int(nec[k]) == postproc.count_cc(want == 3)[0]
    strip = np.zeros((40, 33000), np.uint8)                 # wider than the binned path takes
    strip[2:38:2, 10:2000:2] = 2
    for x0 in (1000, 1990, 30000):
        strip[18:23, x0:x0 + 5] = 1
    want = postproc.meta_inference(strip)
    assert 0 < int((want == 1).sum()) < int((strip == 1).sum())
    out, nec = gpu.meta_inference(strip[None])
    assert np.array_equal(out[0], want) and int(nec[0]) == postproc.count_cc(want == 3)[0]

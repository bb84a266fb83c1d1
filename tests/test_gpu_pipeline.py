"""End-to-end parity of the device pipeline (tile -> U-Net -> stitch/quantise/argmax -> meta_inference -> count)
against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from ecseg_amd import synth
from ecseg_amd.model import MetasegModel
from oracle import pipeline as oracle_pipeline
from oracle import postproc, quant, tiling

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def model16(gpu):
    cfg = synth.unet_config(base=16)
    weights = synth.unet_weights(cfg, seed=0)
    return MetasegModel(cfg, weights, handle=gpu)


def test_segment_matches_oracle_small(model16):
    H, W = 300, 462
    imgs = np.stack([synth.dapi_image(i, H, W) for i in range(3)])
    post, nec, raw = model16.segment(imgs, want_raw=True)
    for i in range(3):
        o_post, o_raw, o_probs, pos = oracle_pipeline.segment_gray(model16.model_config, model16.weights, imgs[i],
                                                                   return_intermediate=True)
        # the GPU's own probabilities agree with the oracle's to 1e-3 ...
        patches = tiling.extract_patches(imgs[i][..., None], pos)
        g_probs = model16.predict_on_batch(patches)
        assert np.abs(g_probs - o_probs).max() < 1e-3
        # ... raw labels may differ only where the oracle's two best quantised classes are within one step
        diff = raw[i] != o_raw
        if diff.any():
            q = np.sort(quant.quantise_u8(tiling.stitch(o_probs, pos))[diff].astype(int), axis=-1)
            assert (q[:, -1] - q[:, -2] <= 1).all()
        assert diff.mean() < 1e-3
        # ... the segment path (cropped plan: only what the stitch reads is computed, Winograd tiles read zeros outside the
        # receptive field of those pixels - ConvParams::in_box) agrees with the stitched full forward pass to float32 rounding:
        # same probabilities to 1e-5, labels equal except where the two best quantised classes are within one step
        g_canvas = tiling.stitch(g_probs, pos)
        s_canvas = model16.handle.segment_images(imgs[i:i + 1], want_raw=False, want_probs=True)[-1][0]
        written = np.abs(g_canvas).sum(-1) > 0                                  # (never-written canvas pixels stay 0 in both)
        assert np.abs(s_canvas - g_canvas)[written].max() < 1e-5
        d2 = raw[i] != quant.quantised_argmax(g_canvas)
        if d2.any():
            q = np.sort(quant.quantise_u8(g_canvas)[d2].astype(int), axis=-1)
            assert (q[:, -1] - q[:, -2] <= 1).all()
        assert d2.mean() < 1e-4
        # ... and everything downstream of the raw labels is bit-exact
        want_post = postproc.meta_inference(raw[i])
        assert np.array_equal(post[i], want_post)
        assert nec[i] == postproc.count_cc(want_post == 3)[0]


def test_segment_full_size_properties(model16):
    img = synth.dapi_image(0)
    post, nec, raw = model16.segment(img, want_raw=True)
    assert post.shape == raw.shape == (1040, 1392)
    want_post = postproc.meta_inference(raw)
    assert np.array_equal(post, want_post)
    assert nec == postproc.count_cc(want_post == 3)[0]
    # batch invariance: the same image inside a batch gives the same answer
    post2, nec2 = model16.segment(np.stack([synth.dapi_image(1), img]))
    assert np.array_equal(post2[1], post) and nec2[1] == nec


def test_batch_csv_full_size_vs_oracle(model16):
    """BASELINE configs[2] in miniature: a batch of full-size (1040x1392) images through the whole device pipeline, the
    ec_quantification.csv text diffed against the CPU oracle run end to end (U-Net included) on the same inputs."""
    from ecseg_amd import csvio
    from oracle import overlay as oracle_overlay
    n = 3
    imgs = np.stack([synth.dapi_image(20 + i) for i in range(n)])
    post, nec, raw = model16.segment(imgs, want_raw=True)
    rows_gpu = [['img%d.tif' % i, int(nec[i])] for i in range(n)]
    rows_cpu = []
    for i in range(n):
        o_post, o_raw, o_probs, pos = oracle_pipeline.segment_gray(model16.model_config, model16.weights, imgs[i],
                                                                   return_intermediate=True)
        mism = int((raw[i] != o_raw).sum())
        if mism:      # legitimate only on near-ties of the quantised probabilities; then compare the clean-up of equal inputs
            q = np.sort(quant.quantise_u8(tiling.stitch(o_probs, pos))[raw[i] != o_raw].astype(int), axis=-1)
            assert (q[:, -1] - q[:, -2] <= 1).all()
            o_post = postproc.meta_inference(raw[i])
        assert np.array_equal(post[i], o_post), i
        rows_cpu.append(['img%d.tif' % i, oracle_pipeline.num_ecdna(o_post)])
    assert csvio.csv_text(csvio.METASEG_COLUMNS, rows_gpu) == oracle_overlay.csv_text(oracle_overlay.METASEG_COLUMNS, rows_cpu)


def test_overlay_full_size_batch_vs_oracle(gpu):
    """BASELINE configs[4] in miniature: metaseg labels + red/green FISH thresholds over full-size 3-channel images."""
    from oracle import overlay as oracle_overlay
    from ecseg_amd import csvio
    n = 3
    rgb = np.stack([synth.dapi_image(60 + i, rgb=True) for i in range(n)])
    labels = np.stack([postproc.meta_inference(synth.label_map(60 + i)).astype(np.uint8) for i in range(n)])
    rec = gpu.overlay(labels, rgb, 85)
    for i in range(n):
        want = oracle_overlay.overlay_row(labels[i], rgb[i], 85)
        got = csvio.overlay_cells(rec[i])
        assert csvio.csv_text(csvio.OVERLAY_COLUMNS, [['x.tif'] + got]) == \
            oracle_overlay.csv_text(oracle_overlay.OVERLAY_COLUMNS, [['x.tif'] + want]), i


@pytest.mark.gpu
@pytest.mark.parametrize('hw,depth', [((300, 462), 1), ((600, 500), 2), ((1040, 1392), 2), ((256, 256), 2)])
def test_crop_of_unread_regions_changes_nothing(hw, depth):
    """Base-64 model: the last 3x3 convolutions of the two highest decoder levels run the F(4x4) kernel only on the regions
    that the stitch - or the halo of the layers behind them - reads (option crop=1, default), and read zeros outside the
    receptive field of those pixels (a Winograd tile leaks what lies outside an output's 3x3 support into its last bits).
    The probabilities agree with the uncropped run to float32 rounding; raw labels may differ only where the two best
    quantised classes are within one step (none on these inputs)."""
    from ecseg_amd import keras_plan
    from ecseg_amd._lib import Handle
    cfg = synth.unet_config(base=64, depth=depth)
    weights = synth.unet_weights(cfg, seed=9)
    hnd = Handle(0)                                       # own handle: the module's model16 fixture keeps its plan
    try:
        hnd.load_plan(keras_plan.build_plan(cfg, weights, fuse=True))
        imgs = np.stack([synth.dapi_image(40 + i, hw[0], hw[1]) for i in range(2)])
        hnd.set_option('crop', 1)
        a = hnd.segment_images(imgs, want_raw=True, want_probs=True)
        hnd.set_option('crop', 0)
        b = hnd.segment_images(imgs, want_raw=True, want_probs=True)
    finally:
        hnd.close()
    assert np.abs(a[3] - b[3]).max() < 1e-5
    d = a[0] != b[0]
    if d.any():
        q = np.sort(quant.quantise_u8(b[3].astype(np.float64))[d].astype(int), axis=-1)
        assert (q[:, -1] - q[:, -2] <= 1).all() and d.mean() < 1e-5
    else:
        for x, y in zip(a[:3], b[:3]):
            assert np.array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize('base,depth,hw', [(64, 2, (600, 500)), (16, 4, (1040, 1392)), (32, 3, (300, 462))])
def test_crop_is_exact_with_the_direct_kernels(base, depth, hw):
    """the strongest regression oracle of the region lists and need boxes.  With winograd=0 every convolution is the
    direct implicit-GEMM kernel - an output depends on its 3x3 support only, no Winograd tile leaks anything else into its last
    bits - so skipping the regions nobody reads (crop=1: the cropped up-convolutions here) must leave every stitched probability,
    label and count BIT-IDENTICAL to the uncropped plan; a box or region list one pixel too small shows up as a difference
    instead of hiding inside the 1e-5 / near-tie tolerance of the Winograd comparison above."""
    from ecseg_amd import keras_plan
    from ecseg_amd._lib import Handle
    cfg = synth.unet_config(base=base, depth=depth)
    weights = synth.unet_weights(cfg, seed=13)
    hnd = Handle(0)
    try:
        hnd.load_plan(keras_plan.build_plan(cfg, weights, fuse=True))
        hnd.set_option('winograd', 0)
        imgs = np.stack([synth.dapi_image(60 + i, hw[0], hw[1]) for i in range(3)])
        hnd.set_option('crop', 0)
        b = hnd.segment_images(imgs, want_raw=True, want_tie_risk=True, want_probs=True)
        hnd.set_option('crop', 1)
        a = hnd.segment_images(imgs, want_raw=True, want_tie_risk=True, want_probs=True)
    finally:
        hnd.close()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.gpu
def test_repeated_runs_are_bit_identical():
    """Race screen (tools/stress_determinism.py runs the long version): the F(4x4) kernel orders its LDS-DMA traffic with
    hand-counted vmcnt waits and one barrier per 8 channels; any mistake there shows up as run-to-run differences."""
    from ecseg_amd import keras_plan
    from ecseg_amd._lib import Handle
    cfg = synth.unet_config(base=64, depth=2)
    weights = synth.unet_weights(cfg, seed=5)
    hnd = Handle(0)
    try:
        hnd.load_plan(keras_plan.build_plan(cfg, weights, fuse=True))
        imgs = np.stack([synth.dapi_image(70 + i, 512, 640) for i in range(3)])
        ref = hnd.segment_images(imgs, want_raw=True)
        x = np.random.default_rng(3).integers(0, 256, size=(6, 256, 256, 1), dtype=np.uint8)
        pref = hnd.forward_patches(x)
        for _ in range(6):
            out = hnd.segment_images(imgs, want_raw=True)
            for a, b in zip(ref, out):
                assert np.array_equal(a, b)
            assert np.array_equal(pref, hnd.forward_patches(x))
    finally:
        hnd.close()


def test_tie_risk_counts_and_stitched_probabilities(model16):
    """ecseg_segment_images_ex: the stitched probabilities are patches2im_overlap of the device's patch probabilities
    (src/utils.py:115-116; never-written canvas pixels 0), the labels are their quantised argmax, and the
    per-image tie-risk count is the number of written pixels whose two largest uint8-quantised values differ by at most 1 -
    the only pixels on which two float32 evaluations of the network can disagree (two images of different content in one
    batch: the per-image counters must not mix)."""
    H, W = 300, 462           # 36 canvas pixels of this size are never written (tests/golden: stitch maps)
    imgs = np.stack([synth.dapi_image(7, H, W), synth.dapi_image(8, H, W), np.zeros((H, W), np.uint8)])
    raw, post, nec, tie, probs = model16.handle.segment_images(imgs, want_raw=True, want_tie_risk=True, want_probs=True)
    pos = tiling.patch_positions(H, W)
    for i in range(3):
        g_probs = model16.predict_on_batch(tiling.extract_patches(imgs[i][..., None], pos))
        want = tiling.stitch(g_probs, pos)
        # (the segment path computes only what the stitch reads and its Winograd tiles read zeros outside the receptive field of those
        # pixels - ConvParams::in_box: equal to the full forward pass up to float32 rounding; never-written pixels are 0 in both)
        assert probs[i].dtype == np.float32 and np.abs(probs[i] - want.astype(np.float32)).max() < 1e-5
        assert np.array_equal(probs[i].sum(-1) == 0, want.sum(-1) == 0)
        assert np.array_equal(raw[i], quant.quantised_argmax(probs[i].astype(np.float64)))
        q = np.sort(quant.quantise_u8(probs[i].astype(np.float64)).astype(int), axis=-1)
        written = probs[i].sum(-1) > 0.5
        assert int(tie[i]) == int(((q[..., 3] - q[..., 2] <= 1) & written).sum())
    assert tie[0] != tie[1]
    # the plain call returns the same labels and counts
    raw2, post2, nec2 = model16.handle.segment_images(imgs, want_raw=True)
    assert np.array_equal(raw, raw2) and np.array_equal(post, post2) and np.array_equal(nec, nec2)


@pytest.mark.parametrize('kind', ['rgb_u8', 'gray_u16', 'rgba_u16'])
def test_meta_segment_is_preprocess_then_segment_in_one_call(model16, kind):
    """ecseg_meta_segment (what `make metaseg` calls per batch: src/utils.py:105-124 minus the file I/O + src/metaseg.py:46) must
    return exactly what ecseg_preprocess followed by ecseg_segment_images_ex returns - pre-processed images, labels, counts and
    tie-risk bounds - from ordinary and from page-locked host buffers (ecseg_host_alloc), and the pre-processed images must
    equal the oracle's meta_preprocess."""
    from oracle import preprocess as oracle_pre
    H, W = 300, 462
    h = model16.handle
    base = np.stack([synth.dapi_image(40 + i, H, W, rgb=True) for i in range(3)])
    if kind == 'rgb_u8':
        imgs = base
    elif kind == 'gray_u16':
        imgs = (base[..., 2].astype(np.uint16) * 257)
        imgs[1] = 65535 - imgs[1]                                           # a bright-background image: the inverting branch
    else:
        imgs = np.concatenate([base.astype(np.uint16) * 201, np.full(base.shape[:3] + (1,), 65535, np.uint16)], axis=-1)
    gray_a, _ = h.preprocess(imgs)
    _, post_a, nec_a, tie_a = h.segment_images(gray_a, want_raw=False, want_tie_risk=True)
    for i in range(len(imgs)):
        assert np.array_equal(gray_a[i], oracle_pre.meta_preprocess(imgs[i]))
    gray_b, post_b, nec_b, tie_b = h.meta_segment(imgs)
    assert np.array_equal(gray_a, gray_b) and np.array_equal(post_a, post_b)
    assert np.array_equal(nec_a, nec_b) and np.array_equal(tie_a, tie_b)
    p_in = h.host_empty(imgs.shape, imgs.dtype)
    p_gray, p_post = h.host_empty((3, H, W)), h.host_empty((3, H, W))
    try:
        p_in[...] = imgs
        p_gray[...] = 0x5A; p_post[...] = 0x5A
        g, p, nec_c, tie_c = h.meta_segment(p_in, gray_out=p_gray, post_out=p_post)
        assert g is p_gray and p is p_post
        assert np.array_equal(gray_a, p_gray) and np.array_equal(post_a, p_post)
        assert np.array_equal(nec_a, nec_c) and np.array_equal(tie_a, tie_c)
        # images sent ahead (ecseg_prefetch_input): named before the call that precedes theirs, uploaded under that call's kernels,
        # recognised by the next call; a call with other images drops them; results never change
        p_in2 = h.host_empty(imgs.shape, imgs.dtype)
        try:
            p_in2[...] = imgs[::-1]
            want2 = h.meta_segment(np.ascontiguousarray(imgs[::-1]))
            h.prefetch_input(p_in2)
            r1 = h.meta_segment(p_in)                      # uploads p_in2 on the side
            r2 = h.meta_segment(p_in2)                     # finds it on the device
            h.prefetch_input(p_in2)
            r3 = h.meta_segment(p_in)
            r4 = h.meta_segment(imgs)                      # other images: own upload, the prefetched ones are dropped
            r5 = h.meta_segment(p_in2)                     # ... and uploaded again here
            for r, want in ((r1, (gray_a, post_a, nec_a, tie_a)), (r2, want2), (r3, (gray_a, post_a, nec_a, tie_a)),
                            (r4, (gray_a, post_a, nec_a, tie_a)), (r5, want2)):
                assert all(np.array_equal(x, y) for x, y in zip(r, want))
        finally:
            h.host_release(p_in2)
    finally:
        for b in (p_in, p_gray, p_post):
            h.host_release(b)
    with pytest.raises(ValueError):
        h.meta_segment(imgs, gray_out=np.empty((2, H, W), np.uint8), post_out=np.empty((3, H, W), np.uint8))

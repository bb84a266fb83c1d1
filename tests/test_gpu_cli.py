"""`make metaseg` / `make meta_overlay` end to end on the GPU: config.yaml in, labels/*.npy + *.png, dapi/*.tif,
ec_quantification.csv and fish_quantification.csv out, compared with the CPU oracle driven over the same files."""
import os
import shutil

import numpy as np
import pytest
import yaml
from PIL import Image

from ecseg_amd import hdf5_min, image_io, synth
from oracle import overlay as oracle_overlay
from oracle import pipeline as oracle_pipeline
from oracle import postproc, preprocess

pytestmark = pytest.mark.gpu
MPL_RGBA = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'io_label_png.npz'))['class_rgba']


@pytest.fixture()
def workdir(tmp_path, golden_dir, monkeypatch):
    os.makedirs(tmp_path / 'models')
    shutil.copy(os.path.join(golden_dir, 'metaseg_synth_b8.h5'), tmp_path / 'models' / 'metaseg.h5')
    inp = tmp_path / 'images'
    os.makedirs(inp)
    for k in range(3):
        rgb = synth.dapi_image(40 + k, 300, 400, rgb=True)
        if k == 1:
            rgb[..., 2] = 255 - rgb[..., 2]                 # white background: meta_preprocess must invert it
        Image.fromarray(rgb).save(str(inp / ('img%d.tif' % k)), compression='tiff_lzw')
    image_io.write_tiff_gray8(str(inp / 'gray.tif'), synth.dapi_image(50, 280, 300))     # a gray image of another size
    with open(tmp_path / 'config.yaml', 'w') as f:
        yaml.safe_dump({'metaseg': {'inpath': str(inp)}, 'meta_overlay': {'inpath': str(inp), 'color_sensitivity': 85}}, f)
    monkeypatch.chdir(tmp_path)
    return tmp_path, inp


def test_metaseg_then_overlay_cli(workdir):
    from ecseg_amd import meta_overlay, metaseg
    tmp, inp = workdir
    metaseg.main([])
    cfg, weights = hdf5_min.load_keras_h5(str(tmp / 'models' / 'metaseg.h5'))
    names = sorted(os.listdir(inp))
    tifs = [n for n in names if n.endswith('.tif')]
    assert tifs == ['gray.tif', 'img0.tif', 'img1.tif', 'img2.tif']
    rows = []
    for n in tifs:
        img = np.array(Image.open(str(inp / n)))                   # the oracle side decodes with PIL, not with the product's reader
        assert np.array_equal(img, image_io.imread(str(inp / n)))
        gray = preprocess.meta_preprocess(img)
        assert np.array_equal(np.array(Image.open(str(inp / 'dapi' / n))), 255 - gray), n     # cv2.bitwise_not(I); PIL = independent decoder
        lab = np.load(str(inp / 'labels' / (n[:-4] + '.npy')))
        assert lab.dtype == np.int64 and lab.shape == gray.shape
        want = oracle_pipeline.segment_gray(cfg, weights, gray)
        if not np.array_equal(lab, want):
            # only legitimate when the raw argmax differs on near-ties; then the clean-up of the GPU's own raw labels
            # must still be exact, which test_gpu_pipeline checks.  Report it loudly here.
            pytest.fail('%s: %d label pixels differ from the oracle' % (n, int((lab != want).sum())))
        png = np.array(Image.open(str(inp / 'labels' / (n[:-4] + '.png'))))
        assert np.array_equal(png, MPL_RGBA[lab])                   # colours pinned by matplotlib's own imsave output
        rows.append([n, postproc.count_cc(want == 3)[0]])
    text = open(str(inp / 'ec_quantification.csv')).read()
    assert text == oracle_overlay.csv_text(oracle_overlay.METASEG_COLUMNS, rows)
    assert open(str(inp / 'ec_quantifications.csv')).read() == text

    meta_overlay.main([])
    rows = []
    for n in tifs:
        img = image_io.imread(str(inp / n))
        if img.ndim < 3:
            continue
        lab = np.load(str(inp / 'labels' / (n[:-4] + '.npy')))
        rows.append([n] + oracle_overlay.overlay_row(lab, img, 85))
        assert np.array_equal(np.array(Image.open(str(inp / 'red' / (n + '.png')))), 255 - img[..., 0])
        assert np.array_equal(np.array(Image.open(str(inp / 'green' / (n + '.png')))), 255 - img[..., 1])
    assert open(str(inp / 'fish_quantification.csv')).read() == oracle_overlay.csv_text(oracle_overlay.OVERLAY_COLUMNS, rows)


def test_corrupt_image_is_reported_and_exit_code_is_nonzero(workdir):
    """A file that cannot be decoded must not take the run down (per-image status), but it must not pass
    silently either: the others are processed, it is absent from the CSV, a summary is printed and the exit code is 1."""
    from ecseg_amd import metaseg
    tmp, inp = workdir
    with open(str(inp / 'broken.tif'), 'wb') as f:
        f.write(b'II*\x00\x08\x00\x00\x00' + b'\xff' * 40)
    with pytest.raises(SystemExit) as e:
        metaseg.main([])
    assert e.value.code == 1
    text = open(str(inp / 'ec_quantification.csv')).read()
    assert 'broken.tif' not in text and text.count('\n') == 1 + 4
    assert os.path.exists(str(inp / 'labels' / 'img2.npy')) and not os.path.exists(str(inp / 'labels' / 'broken.npy'))


def test_resume_keeps_existing_outputs_and_reproduces_the_csv(workdir):
    """Optional `resume: true` (SURVEY 5 checkpoint / resume): a second run re-uses the stored labels - files untouched, the
    CSV identical; a removed output is re-created."""
    import time
    from ecseg_amd import metaseg
    tmp, inp = workdir
    cfg = yaml.safe_load(open(tmp / 'config.yaml'))
    cfg['metaseg']['resume'] = True
    yaml.safe_dump(cfg, open(tmp / 'config.yaml', 'w'))
    metaseg.main([])
    first = open(str(inp / 'ec_quantification.csv')).read()
    stamp = {n: os.stat(str(inp / 'labels' / n)).st_mtime_ns for n in os.listdir(str(inp / 'labels'))}
    os.remove(str(inp / 'labels' / 'img1.png'))
    time.sleep(0.05)
    metaseg.main([])
    assert open(str(inp / 'ec_quantification.csv')).read() == first
    for n, t in stamp.items():
        if n.startswith('img1'):
            continue
        assert os.stat(str(inp / 'labels' / n)).st_mtime_ns == t, n        # kept, not rewritten
    assert os.path.exists(str(inp / 'labels' / 'img1.png'))                # the incomplete image was redone
    assert os.stat(str(inp / 'labels' / 'img1.npy')).st_mtime_ns > stamp['img1.npy']


def test_metaseg_cli_exit_codes(tmp_path, monkeypatch):
    from ecseg_amd import meta_overlay, metaseg
    with open(tmp_path / 'config.yaml', 'w') as f:
        yaml.safe_dump({'metaseg': {'inpath': str(tmp_path / 'nope')},
                        'meta_overlay': {'inpath': str(tmp_path), 'color_sensitivity': 300}}, f)
    monkeypatch.chdir(tmp_path)
    with pytest.raises(SystemExit) as e:
        metaseg.main([])
    assert e.value.code == 2
    with pytest.raises(SystemExit) as e:                 # labels/ missing
        meta_overlay.main([])
    assert e.value.code == 2
    os.makedirs(tmp_path / 'labels'); os.makedirs(tmp_path / 'dapi')
    with pytest.raises(SystemExit) as e:                 # sensitivity out of range
        meta_overlay.main([])
    assert e.value.code == 2


def test_u16_and_reference_call_shapes(gpu):
    from ecseg_amd import image_tools
    image_tools.set_default_handle(gpu)
    rng = np.random.default_rng(1)
    a = rng.integers(0, 65536, size=(50, 60, 3)).astype(np.uint16)
    assert np.array_equal(image_tools.u16_to_u8(a), preprocess.u16_to_u8(a))
    m = rng.random((80, 90)) < 0.3
    assert image_tools.count_cc(m) == postproc.count_cc(m)
    assert image_tools.count_cc(np.zeros((8, 8), bool)) == (0, 0.0)
    m2 = rng.random((80, 90)) < 0.3
    assert image_tools.count_colocalization(m, m2) == postproc.count_colocalization(m, m2)
    assert image_tools.count_HSR(m, m2, 20) == postproc.count_HSR(m, m2, 20)
    lab = synth.label_map(3, 120, 150)
    out = image_tools.meta_inference(lab.astype(np.int64))
    assert out.dtype == np.int64 and np.array_equal(out, postproc.meta_inference(lab))
    img, patches, pos = image_tools.im2patches_overlap(np.zeros((300, 462, 1), np.uint8))
    from oracle import tiling
    assert np.array_equal(np.array(pos), tiling.patch_positions(300, 462))


def test_precision_and_emit_probs_config_keys(workdir):
    """`metaseg: {precision: exact | fast, emit_probs: true}`.  exact = Winograd F(2x2) everywhere (the
    kernel family closest to a float64 evaluation); emit_probs writes the stitched float32 probabilities np.argmax saw
    (src/utils.py:116-118) next to the labels; every image's record carries its tie-risk pixel count (record slot 15 ->
    ec_quantification_report.json), which bounds the raw-label disagreement with any other float32 evaluation."""
    import json
    from ecseg_amd import metaseg
    from oracle import quant, tiling
    tmp, inp = workdir
    cfg_path = tmp / 'config.yaml'
    base = yaml.safe_load(open(cfg_path))
    outs = {}
    for precision in ('fast', 'exact'):
        c = dict(base)
        c['metaseg'] = dict(base['metaseg'], precision=precision, emit_probs=True)
        yaml.safe_dump(c, open(cfg_path, 'w'))
        for sub in ('labels', 'dapi'):
            shutil.rmtree(str(inp / sub), ignore_errors=True)
        metaseg.main([])
        rep = json.load(open(str(inp / 'ec_quantification_report.json')))
        assert [r['image_name'] for r in rep['images']] == ['gray.tif', 'img0.tif', 'img1.tif', 'img2.tif']
        outs[precision] = {}
        for r in rep['images']:
            stem = r['image_name'][:-4]
            probs = np.load(str(inp / 'labels' / (stem + '_probs.npy')))
            lab = np.load(str(inp / 'labels' / (stem + '.npy')))
            gray = 255 - np.array(Image.open(str(inp / 'dapi' / r['image_name'])))
            assert probs.dtype == np.float32 and probs.shape == lab.shape + (4,)
            # the stored probabilities are what the labels were made from ...
            raw = quant.quantised_argmax(probs.astype(np.float64))
            assert np.array_equal(postproc.meta_inference(raw), lab), r['image_name']
            assert r['# of ec'] == postproc.count_cc(lab == 3)[0] and r['status'] == 0
            # ... and the record's tie-risk count is the number of written pixels whose two largest quantised values are <= 1 apart
            q = np.clip(np.rint(probs.astype(np.float64) * 255.0), 0, 255)
            top = np.sort(q, axis=-1)
            written = probs.sum(-1) > 0.5
            assert r['tie_risk_pixels'] == int(((top[..., 3] - top[..., 2] <= 1) & written).sum()), r['image_name']
            outs[precision][stem] = (probs, lab, gray)
    # the two kernel families agree to float32 rounding, and wherever their raw labels differ the pixel is a counted tie risk
    for stem in outs['fast']:
        pf, lf, _ = outs['fast'][stem]
        pe, le, _ = outs['exact'][stem]
        assert np.abs(pf - pe).max() < 1e-4
        assert not np.array_equal(pf, pe), 'precision: exact did not select another kernel family'
    # an unknown value is refused with the CLI's exit code 2
    c = dict(base)
    c['metaseg'] = dict(base['metaseg'], precision='double')
    yaml.safe_dump(c, open(cfg_path, 'w'))
    with pytest.raises(SystemExit) as e:
        metaseg.main([])
    assert e.value.code == 2


def test_device_workers_two_handles_same_outputs(workdir):
    """`metaseg: {device_workers: 2}`: two handles on the same GPU (own streams, activation buffers, weights) fed from one queue of
    batches; every output file and the CSV equal the one-worker run."""
    from ecseg_amd import metaseg
    tmp, inp = workdir
    cfg_path = tmp / 'config.yaml'
    base = yaml.safe_load(open(cfg_path))
    snaps = []
    for workers in (1, 2):
        c = dict(base)
        c['metaseg'] = dict(base['metaseg'], device_workers=workers, batch_images=1)
        yaml.safe_dump(c, open(cfg_path, 'w'))
        for sub in ('labels', 'dapi'):
            shutil.rmtree(str(inp / sub), ignore_errors=True)
        metaseg.main([])
        snap = {}
        for sub in ('labels', 'dapi', ''):
            d = inp / sub if sub else inp
            for f in sorted(os.listdir(str(d))):
                p = d / f
                if p.is_file() and (sub or f.endswith(('.csv', '.json'))):
                    snap[os.path.join(sub, f)] = p.read_bytes()
        snaps.append(snap)
    assert sorted(snaps[0]) == sorted(snaps[1])
    for k in snaps[0]:
        assert snaps[0][k] == snaps[1][k], k


def test_integration_stub_of_the_reference_side_runs(workdir):
    """the ctypes stub a reference maintainer pastes into src/utils.py - executed as written (its own raw
    ctypes binding, its struct definitions, `load_model`, `meta_segment`, `meta_segment_batch`) with the reference's
    surrounding names supplied, and compared with the package's own path."""
    import types
    from ecseg_amd import utils as eutils
    from ecseg_amd._lib import LIB_PATH
    tmp, inp = workdir
    os.makedirs(inp / 'dapi', exist_ok=True)
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'INTEGRATION.md')).read()
    start = text.index("```python\nimport ctypes, numpy as np\n_lib = ctypes.CDLL(")
    code = text[start + len('```python\n'):text.index('```', start + 10)]
    code = code.replace("'/path/to/ecseg_amd/libecseg_hip.so'", repr(LIB_PATH))
    saved = []
    ns = {'os': os, 'imread': image_io.imread, 'meta_preprocess': preprocess.meta_preprocess,
          'cv2': types.SimpleNamespace(bitwise_not=lambda a: ~a), 'save_img': lambda img, split, sub: saved.append((split[1], sub, img))}
    exec(code, ns)
    model = ns['load_model']('metaseg.h5')
    ours = eutils.load_model('metaseg.h5')
    paths = [str(inp / ('img%d.tif' % k)) for k in range(3)]
    for p in paths:
        lab = ns['meta_segment'](model, p)
        assert lab.dtype == np.int64 and np.array_equal(lab, eutils.meta_segment(ours, p))
    labs, nec = ns['meta_segment_batch'](model, paths)
    for j, p in enumerate(paths):
        want = eutils.meta_segment(ours, p)
        assert np.array_equal(labs[j], want) and nec[j] == postproc.count_cc(want == 3)[0]
    assert len(saved) == 6 and all(s[1] == 'dapi' for s in saved)
    ns['_lib'].ecseg_destroy(ns['_h'])


def test_page_locked_batch_buffers_change_nothing(workdir):
    """`make metaseg`'s page-locked recycled batch buffers and the next batch's upload under the current batch's kernels
    (ecseg_host_alloc, ecseg_prefetch_input; switched on for jobs of >= 4 batches, forced here): records and every output
    file equal a run over ordinary memory, in the first run (buffers arrive while it runs) and in the second (all hits)."""
    from ecseg_amd import metaseg, utils as eutils
    tmp, inp = workdir
    model = eutils.load_model('metaseg.h5')
    paths = eutils.get_imgs(str(inp))

    def one_run(**kw):
        for sub in ('labels', 'dapi'):
            shutil.rmtree(str(inp / sub), ignore_errors=True)
            os.makedirs(inp / sub)
        stats = {}
        rec = metaseg.run(str(inp), model, paths, batch_images=1, io_threads=2, log=lambda *a: None, stats=stats, **kw)
        snap = {os.path.join(sub, f): (inp / sub / f).read_bytes() for sub in ('labels', 'dapi') for f in sorted(os.listdir(str(inp / sub)))}
        return rec, snap, stats

    rec0, snap0, st0 = one_run(pinned_mb=0)
    assert st0['pinned_pool']['hits'] == 0
    # a four-image run is over before the pool's thread has locked anything (orders still pending when a run ends are dropped):
    # stock the pool first, as the early batches of a long run do
    import time
    pool = metaseg._PinnedPool.of(model.handle, 256 << 20)
    pool.start()
    sizes = [300 * 400 * 3] * 3 + [300 * 400] * 6 + [280 * 300] * 9
    pool.reserve(sizes)
    t0 = time.time()
    while len(pool.free) < len(sizes) and time.time() - t0 < 20:
        time.sleep(0.01)
    pool.stop()
    assert len(pool.free) == len(sizes)
    rec1, snap1, st1 = one_run(pinned_min_images=0)
    assert st1['pinned_pool']['hits'] >= 8, st1['pinned_pool']
    assert len(pool.free) >= len(sizes)                                    # every buffer came back
    assert np.array_equal(rec0, rec1)
    assert sorted(snap1) == sorted(snap0) and all(snap1[k] == snap0[k] for k in snap0)

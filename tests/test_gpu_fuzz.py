"""Randomised parity (seeded, reproducible): many geometries and densities through the union-find labelling, the counting
entry points, meta_inference and the stitch / quantised argmax, device vs CPU oracle, bit-exact.  Complements the golden
vectors (fixed cases captured from the reference) with breadth: ragged sizes, extreme densities, tile-border contacts."""
import numpy as np
import pytest

from ecseg_amd import synth
from oracle import postproc, quant, tiling

pytestmark = pytest.mark.gpu


def _canon(lab):
    out = np.zeros(lab.shape, np.int32)
    if lab.max() > 0:
        idx = np.arange(lab.size).reshape(lab.shape)
        first = np.full(lab.max() + 1, lab.size, np.int64)
        np.minimum.at(first, lab.ravel(), idx.ravel())
        out = np.where(lab > 0, first[lab] + 1, 0).astype(np.int32)
    return out


def _random_mask(rng, H, W):
    kind = int(rng.integers(0, 6))
    if kind == 0:
        return (rng.random((H, W)) < rng.choice([0.02, 0.2, 0.41, 0.5, 0.6, 0.9, 0.99])).astype(np.uint8)
    if kind == 1:                                            # blobs
        m = np.zeros((H, W), np.uint8)
        yy, xx = np.ogrid[:H, :W]
        for _ in range(int(rng.integers(1, 40))):
            r = int(rng.integers(1, max(2, min(H, W) // 3)))
            cy, cx = int(rng.integers(0, H)), int(rng.integers(0, W))
            m[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 1
        return m
    if kind == 2:                                            # lines along tile borders (multiples of 32 rows / 64 columns)
        m = np.zeros((H, W), np.uint8)
        m[31::32, :] = 1
        m[:, 63::64] = rng.integers(0, 2)
        m[rng.random((H, W)) < 0.05] ^= 1
        return m
    if kind == 3:                                            # diagonal stripes: only diagonal contacts
        yy, xx = np.mgrid[:H, :W]
        return (((yy + xx) % int(rng.integers(2, 5))) == 0).astype(np.uint8)
    if kind == 4:                                            # one huge component with holes
        m = np.ones((H, W), np.uint8)
        m[rng.random((H, W)) < 0.15] = 0
        return m
    m = np.zeros((H, W), np.uint8)                           # sparse dots
    n = int(rng.integers(0, 30))
    m[rng.integers(0, H, n), rng.integers(0, W, n)] = 1
    return m


@pytest.mark.parametrize('seed', range(6))
def test_fuzz_ccl_and_counts(gpu, seed):
    rng = np.random.default_rng(1000 + seed)
    for _ in range(24):
        H, W = int(rng.integers(1, 200)), int(rng.integers(1, 300))
        n = int(rng.integers(1, 4))
        a = np.stack([_random_mask(rng, H, W) for _ in range(n)])
        b = np.stack([_random_mask(rng, H, W) for _ in range(n)])
        for conn, lab_fn in ((8, postproc.label8), (4, postproc.label4)):
            got = gpu.ccl_labels(a, conn)
            for k in range(n):
                assert np.array_equal(got[k], _canon(lab_fn(a[k])[0])), (seed, H, W, conn)
        cnt, px = gpu.count_cc(a)
        col = gpu.count_colocalization(a, b)
        hsr = gpu.count_hsr(a, b, 20)
        for k in range(n):
            wn, wpx = postproc.count_cc(a[k].astype(bool))
            assert int(cnt[k]) == wn and (int(px[k]) == wpx or (px[k] == -1 and wpx == 0.0)), (seed, H, W)
            assert int(col[k]) == postproc.count_colocalization(a[k].astype(bool), b[k].astype(bool))
            assert int(hsr[k]) == postproc.count_HSR(a[k].astype(bool), b[k].astype(bool), 20)


def _random_labels(rng, H, W):
    kind = int(rng.integers(0, 4))
    if kind == 1 and min(H, W) < 250:
        kind = 0
    if kind == 0:
        p = rng.dirichlet(np.ones(4) * rng.choice([0.3, 1.0, 5.0]))
        return rng.choice(4, size=(H, W), p=p).astype(np.uint8)
    if kind == 1:
        return synth.label_map(int(rng.integers(0, 10 ** 6)), H, W, salt=float(rng.choice([0.0, 0.002, 0.05])))
    lab = np.zeros((H, W), np.uint8)
    yy, xx = np.ogrid[:H, :W]
    for _ in range(int(rng.integers(1, 60))):                # overlapping discs / rings of random classes
        r = int(rng.integers(1, max(2, min(H, W) // 4)))
        cy, cx = int(rng.integers(0, H)), int(rng.integers(0, W))
        d2 = (yy - cy) ** 2 + (xx - cx) ** 2
        c = int(rng.integers(0, 4))
        lab[d2 <= r * r] = c
        if kind == 3 and r > 3:
            lab[d2 <= (r // 2) ** 2] = int(rng.integers(0, 4))
    return lab


@pytest.mark.parametrize('seed', range(6))
def test_fuzz_meta_inference(gpu, seed):
    rng = np.random.default_rng(2000 + seed)
    for _ in range(16):
        H, W = int(rng.integers(3, 300)), int(rng.integers(3, 400))
        n = int(rng.integers(1, 4))
        labs = np.stack([_random_labels(rng, H, W) for _ in range(n)])
        out, nec = gpu.meta_inference(labs)
        for k in range(n):
            want = postproc.meta_inference(labs[k])
            assert np.array_equal(out[k], want), (seed, H, W, k)
            assert int(nec[k]) == postproc.count_cc(want == 3)[0]


@pytest.mark.parametrize('seed', range(3))
def test_fuzz_stitch_quantised_argmax(gpu, seed):
    rng = np.random.default_rng(3000 + seed)
    for _ in range(3):
        H, W = int(rng.integers(256, 700)), int(rng.integers(256, 700))
        pos = tiling.patch_positions(H, W)
        z = rng.normal(size=(len(pos), 256, 256, 4)).astype(np.float32) * float(rng.choice([0.2, 2.0, 8.0]))
        p = np.exp(z - z.max(-1, keepdims=True))
        p = (p / p.sum(-1, keepdims=True)).astype(np.float32)
        q = rng.integers(0, 510, size=p.shape[:3] + (2,))   # plant exact half-way values (k + 0.5) / 255 in two channels
        m = rng.random(p.shape[:3]) < 0.05
        p[m, 0] = ((q[m, 0] // 2) + 0.5) / 255.0
        p[m, 1] = ((q[m, 1] // 2) + 0.5) / 255.0
        got = gpu.stitch_argmax(p, 1, H, W)[0]
        assert np.array_equal(got, quant.quantised_argmax(tiling.stitch(p, pos))), (seed, H, W)


# Seeds of tools/fuzz_campaign.py that once failed: a tile root in the image's last 64-column chunk pointed at a root LEFT of
# the tile in a later row, and ccl_flatten's division-free in-tile test took the row-wrapped offset for an in-tile column.
CAMPAIGN_REGRESSIONS = (1405, 2096, 2602, 2765, 2776, 3812, 4193, 4826, 4948, 6014, 6136)


def _campaign_case(seed):
    """The case tools/fuzz_campaign.py builds for `seed` (same generator calls in the same order)."""
    rng = np.random.default_rng(10 ** 6 + seed)
    big = seed % 3 == 0
    H = int(rng.integers(200, 1100)) if big else int(rng.integers(1, 260))
    W = int(rng.integers(300, 1500)) if big else int(rng.integers(1, 400))
    n = int(rng.integers(1, 4))
    masks = np.stack([_random_mask(rng, H, W) for _ in range(n)])
    labs = np.stack([_random_labels(rng, H, W) for _ in range(n)]) if H >= 3 and W >= 3 else None
    return masks, labs


@pytest.mark.parametrize('seed', CAMPAIGN_REGRESSIONS)
def test_fuzz_campaign_regressions(gpu, seed):
    masks, labs = _campaign_case(seed)
    for conn, lab_fn in ((8, postproc.label8), (4, postproc.label4)):
        got = gpu.ccl_labels(masks, conn)
        for k in range(len(masks)):
            assert np.array_equal(got[k], _canon(lab_fn(masks[k])[0])), (seed, conn, k)
    out, nec = gpu.meta_inference(labs)
    for k in range(len(labs)):
        want = postproc.meta_inference(labs[k])
        assert np.array_equal(out[k], want), (seed, k, labs.shape)
        assert int(nec[k]) == postproc.count_cc(want == 3)[0]


# Random layer graphs (tools/fuzz_layers.py) under every kernel mode and fusion setting.  The second seed block once failed:
# an AveragePooling2D behind a Winograd F(4x4) convolution was taken for the max-pool the output stage can write itself.
@pytest.mark.parametrize('seed0', (0, 8, 16, 4090, 4117, 4121, 4227))
def test_fuzz_layer_graphs(seed0):
    from ecseg_amd.model import MetasegModel
    from oracle import unet as oracle_unet
    from tools.fuzz_layers import random_graph
    for seed in range(seed0, seed0 + (8 if seed0 < 100 else 1)):
        rng = np.random.default_rng(5 * 10 ** 6 + seed)
        cfg, weights, (h, w, cin) = random_graph(rng)
        n = int(rng.integers(1, 5))
        x = rng.integers(0, 256, size=(n, h, w, cin), dtype=np.uint8)
        want = oracle_unet.forward(cfg, weights, x.astype(np.float32))
        scale = max(1.0, float(np.abs(want).max()))
        m = MetasegModel(cfg, weights, device=0)
        for mode in (2, 3, 1, 0):
            m.handle.set_option('winograd', mode)
            for fuse in (1, 0):
                m.handle.set_option('fuse_pool', fuse)
                m.handle.set_option('fuse_head', fuse)
                got = m.handle.forward_patches(x)
                assert np.isfinite(got).all() and float(np.abs(got - want).max()) / scale <= 1e-3, (seed, mode, fuse)   # tolerance: BASELINE.json north_star
        del m


def test_fuzz_loader_corners(monkeypatch):
    """24 graphs of `tools/fuzz_layers.py --loader` (round 6: Conv2D with per-axis strides / dilation rates, layers called twice,
    channels_first twins fed (N, C, H, W)), every kernel mode and fusion setting; the 420-second campaign is profiles/r06_fuzz_loader.log."""
    import os
    import runpy
    import sys
    monkeypatch.setattr(sys, 'argv', ['fuzz_layers.py', '--loader', '--seeds', ','.join(str(s) for s in range(24)), '--seconds', '600'])
    with pytest.raises(SystemExit) as e:
        runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_layers.py'), run_name='__main__')
    assert e.value.code == 0


def test_fuzz_pipeline_cases(monkeypatch):
    """Six cases of tools/fuzz_pipeline.py (random small U-Nets x random image sizes: the cropped plan vs itself (history, lanes) and vs the uncropped plan, probabilities vs the
    oracle, clean-up / counts on the device raw labels, meta_preprocess, overlay rows)."""
    import os
    import runpy
    import sys
    monkeypatch.setattr(sys, 'argv', ['fuzz_pipeline.py', '--seeds', '0,1,2,3,4,5'])
    with pytest.raises(SystemExit) as e:
        runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_pipeline.py'),
                       run_name='__main__')
    assert e.value.code == 0


def test_fuzz_cli_folders(monkeypatch):
    """Eight folders of tools/fuzz_cli.py: mixed image sizes / sample types / TIFF flavours / .npy inputs through the
    `make metaseg` and `make meta_overlay` loops with random batch sizes and I/O thread counts."""
    import os
    import runpy
    import sys
    monkeypatch.setattr(sys, 'argv', ['fuzz_cli.py', '--seeds', '0,1,2,3,4,5,6,7'])
    with pytest.raises(SystemExit) as e:
        runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_cli.py'),
                       run_name='__main__')
    assert e.value.code == 0

#!/usr/bin/env python3
"""User-visible rate of `make metaseg`: N generated 1040x1392 RGB LZW TIFF files in, labels/*.npy + *.png, dapi/*.tif and
ec_quantification.csv out ("time `make metaseg` on 256 generated files").  Prints one JSON line.

    python tools/time_cli.py [--n 256] [--base 64] [--batch 16] [--io-threads 32] [--keep DIR]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cgroup_cpu():
    """cgroup v2 CPU quota and throttle counters of this process's group (None where there is no such file)."""
    try:
        q = open('/sys/fs/cgroup/cpu.max').read().split()
        st = dict(l.split() for l in open('/sys/fs/cgroup/cpu.stat').read().splitlines())
        return {'quota_cpus': None if q[0] == 'max' else round(int(q[0]) / int(q[1]), 2), 'usage_s': int(st['usage_usec']) / 1e6,
                'nr_throttled': int(st.get('nr_throttled', 0)), 'throttled_s': int(st.get('throttled_usec', 0)) / 1e6}
    except (OSError, ValueError, KeyError, IndexError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=256)
    ap.add_argument('--base', type=int, default=64)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--io-threads', type=int, default=None)
    ap.add_argument('--keep', default=None)
    ap.add_argument('--workers', type=int, default=1, help='handles (device threads) on the GPU: config key device_workers')
    ap.add_argument('--pinned-mb', type=int, default=2048, help='page-locked host memory for the batch buffers (config key pinned_mb; 0: none)')
    ap.add_argument('--input-compression', default='tiff_lzw', choices=('tiff_lzw', 'raw'), help='compression of the generated input TIFFs (raw = uncompressed, as many microscopes write them)')
    ap.add_argument('--weights', default='random', choices=('random', 'smooth'),
                    help="random: seeded random weights (speckled labels: the slowest case for the label PNG coder); smooth: the fitted base-16 "
                         "stand-in for a trained model (tests/golden/smooth_b16_f16.npz: blobs, as real label maps)")
    ap.add_argument('--warm-one', dest='warm_full', action='store_false', help='warm up with ONE image (rounds 1-4) instead of a full batch per handle')
    a = ap.parse_args()
    from PIL import Image
    import yaml
    from ecseg_amd import metaseg, synth, utils
    from ecseg_amd.model import MetasegModel
    work = a.keep or tempfile.mkdtemp(prefix='ecseg_cli_')
    inp = os.path.join(work, 'images')
    os.makedirs(inp, exist_ok=True)
    for sub in ('dapi', 'labels', 'red', 'green'):           # (--keep DIR of an earlier run: its outputs go, overwriting files is slower than creating them)
        shutil.rmtree(os.path.join(inp, sub), ignore_errors=True)
    shutil.rmtree(os.path.join(work, 'warm'), ignore_errors=True)
    for sub in ('dapi', 'labels'):
        os.makedirs(os.path.join(inp, sub), exist_ok=True)
    base = [synth.dapi_image(600 + i, rgb=True) for i in range(8)]
    t0 = time.perf_counter()
    old = [f for f in os.listdir(inp) if f.endswith('.tif')]
    have = a.keep and len(old) == a.n                    # (--keep DIR of an earlier run with the same --n: reuse its inputs)
    if not have:
        for f in old:
            os.unlink(os.path.join(inp, f))
    for i in range(0 if have else a.n):
        img = np.roll(base[i % 8], (31 * (i // 8), 17 * (i // 8)), axis=(0, 1))
        Image.fromarray(img).save(os.path.join(inp, 'img%04d.tif' % i), compression=a.input_compression)
    t_gen = time.perf_counter() - t0
    in_bytes = sum(os.path.getsize(os.path.join(inp, f)) for f in os.listdir(inp))
    if a.weights == 'smooth':
        from tools import make_smooth_fixture
        a.base = 16
        cfg, weights = make_smooth_fixture.load(os.path.join(ROOT, 'tests', 'golden', 'smooth_b16_f16.npz'))
    else:
        cfg = synth.unet_config(base=a.base)
        weights = synth.unet_weights(cfg, seed=0)
    model = MetasegModel(cfg, weights, device=0)
    extra = [MetasegModel(cfg, weights, device=0) for _ in range(a.workers - 1)]
    if a.base >= 64:
        model.handle.set_images_per_group(16)          # narrower models: the automatic launch-group size
    with open(os.path.join(work, 'config.yaml'), 'w') as f:
        yaml.safe_dump({'metaseg': {'inpath': inp, 'batch_images': a.batch, **({'io_threads': a.io_threads} if a.io_threads else {})}}, f)
    os.chdir(work)
    real_load = metaseg.load_model
    metaseg.load_model = lambda name, device=None: model           # random-weight canonical model instead of models/metaseg.h5
    try:
        warm = os.path.join(work, 'warm')
        for sub in ('', 'dapi', 'labels'):
            os.makedirs(os.path.join(warm, sub), exist_ok=True)
        # first-use allocations (activation buffers of a full batch, post-processing workspace, pinned staging) happen once per
        # handle in the life of a process: warm every handle with one full batch, as any job longer than a second has
        for k in range(a.batch if a.warm_full else 1):
            shutil.copy(os.path.join(inp, 'img0000.tif'), os.path.join(warm, 'w%03d.tif' % k))
        stats = {}
        for mdl in [model] + extra:
            metaseg.run(warm, mdl, utils.get_imgs(warm), batch_images=a.batch, log=lambda *x: None, stats=stats, pinned_mb=a.pinned_mb,
                        pinned_min_images=0)    # (one batch would not start the page-locked pool by itself)
        cg0 = cgroup_cpu()
        t0 = time.perf_counter()
        rec = metaseg.run(inp, [model] + extra if extra else model, utils.get_imgs(inp), batch_images=a.batch, io_threads=a.io_threads, log=lambda *x: None, stats=stats,
                          pinned_mb=a.pinned_mb)
        dt = time.perf_counter() - t0
        cg1 = cgroup_cpu()
    finally:
        metaseg.load_model = real_load
    # `make meta_overlay` over the same files (labels/*.npy from the run above): decode + labels in, red / green PNGs +
    # nine counts per image out
    from ecseg_amd import meta_overlay
    for sub in ('red', 'green'):
        os.makedirs(os.path.join(inp, sub), exist_ok=True)
    t0 = time.perf_counter()
    ov_stats = {}
    rows = meta_overlay.run(inp, model.handle, utils.get_imgs(inp), 85, batch_images=a.batch, io_threads=a.io_threads, log=lambda *x: None,
                            stats=ov_stats)
    dt_ov = time.perf_counter() - t0
    out_bytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(inp) for f in fs) - in_bytes
    print(json.dumps({'what': '`make metaseg` loop: %d RGB %s TIFF files (1040x1392) -> dapi/*.tif, labels/*.png, labels/*.npy (int64), records'
                              % (a.n, 'LZW' if a.input_compression == 'tiff_lzw' else 'uncompressed'), 'images': a.n, 'unet_base': a.base, 'weights': a.weights, 'batch_images': a.batch, 'device_workers': a.workers, 'pinned_mb': a.pinned_mb,
                      'pinned_pool': stats.get('pinned_pool'), 'warmup': 'one full batch per handle' if a.warm_full else 'one image',
                      'io_threads': a.io_threads or 'default', 'cpu_count': os.cpu_count(),
                      'seconds': round(dt, 3), 'images_per_s': round(a.n / dt, 2),
                      # the host side of the loop is CPU work (LZW decode, LZW / PNG / npy encode): a cgroup CPU quota bounds it, and a
                      # throttled group freezes the device thread with everything else
                      'cgroup_cpu': None if not (cg0 and cg1) else {
                          'quota_cpus': cg1['quota_cpus'], 'cpu_seconds_used': round(cg1['usage_s'] - cg0['usage_s'], 3),
                          'cpu_ms_per_image': round(1e3 * (cg1['usage_s'] - cg0['usage_s']) / a.n, 2),
                          'cpus_busy': round((cg1['usage_s'] - cg0['usage_s']) / dt, 2),
                          'periods_throttled': cg1['nr_throttled'] - cg0['nr_throttled'],
                          'throttled_thread_seconds': round(cg1['throttled_s'] - cg0['throttled_s'], 3)},
                      'device_call_seconds': round(stats.get('gpu_seconds', 0.0), 3),
                      'input_MB': round(in_bytes / 1e6, 1), 'output_MB': round(out_bytes / 1e6, 1),
                      'ok_images': int((rec[:, 1] == 0).sum()), 'generate_seconds': round(t_gen, 1),
                      'meta_overlay_seconds': round(dt_ov, 3), 'meta_overlay_images_per_s': round(len(rows) / dt_ov, 2),
                      'host_stage_ms_per_image': {k.replace('_thread_seconds', ''): round(1e3 * v / a.n, 3) for k, v in stats.items() if k.endswith('_thread_seconds')},
                      'overlay_host_stage_ms_per_image': {k.replace('_thread_seconds', ''): round(1e3 * v / a.n, 3) for k, v in ov_stats.items()}}))
    if not a.keep:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == '__main__':
    main()

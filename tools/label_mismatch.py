#!/usr/bin/env python3
"""How often does fp32 summation order flip a uint8-quantised argmax label?  ("measure and commit the
label-mismatch rate ... on a smooth-output model".)

Fits the canonical U-Net for a few hundred steps on synthetic scenes (tools/fit_smooth_model.py; torch CPU), runs the CPU
oracle on N full-size 1040x1392 synthetic images (worker processes, BEFORE this process touches the GPU), then the device
pipeline with each of the three 3x3 kernels (direct / Winograd F(2x2) / F(4x4)) and reports, per mode: raw-label
mismatch pixels, post-processed-label mismatch pixels and |delta n_ec| - per image, in total and scaled to 100 images.
Also reports the same for the seeded RANDOM-weight model (speckled output, the bench model).

    python tools/label_mismatch.py [--base 64] [--fit-steps 150] [--images 16] [--out profiles/r02_label_mismatch.json]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
H, W = 1040, 1392
SEED0 = 5000


def _load(path):
    z = np.load(path)
    w = {}
    for k in z.files:
        name, i = k.rsplit('/', 1)
        w.setdefault(name, {})[int(i)] = z[k]
    return {n: [d[i] for i in sorted(d)] for n, d in w.items()}


def _worker(job):
    base, wpath, idx, threads = job
    import torch
    torch.set_num_threads(threads)
    from ecseg_amd import synth
    from oracle import pipeline, postproc
    cfg = synth.unet_config(base=base)
    w = _load(wpath)
    img = synth.dapi_image(idx, H, W)
    post, raw, _, _ = pipeline.segment_gray(cfg, w, img, batch=7, return_intermediate=True)
    return raw.astype(np.uint8), post.astype(np.uint8), int(postproc.count_cc(post == 3)[0])


def cpu_refs(base, wpath, n):
    import multiprocessing as mp
    import bench
    threads = 1
    nproc = max(1, min(16, bench.host_cpu_budget(), n))
    ctx = mp.get_context('spawn')
    with ctx.Pool(nproc, initializer=bench._cpu_init, initargs=(ctx.Value('i', 0), threads)) as pool:
        return pool.map(_worker, [(base, wpath, SEED0 + i, threads) for i in range(n)], chunksize=1)


def device_modes(base, weights, n, refs):
    from ecseg_amd import synth
    from ecseg_amd.model import MetasegModel
    from oracle import postproc
    cfg = synth.unet_config(base=base)
    model = MetasegModel(cfg, weights, device=0)
    imgs = np.stack([synth.dapi_image(SEED0 + i, H, W) for i in range(n)])
    out = {}
    for mode, tag in ((0, 'direct'), (1, 'winograd_f2x2'), (2, 'winograd_f4x4')):
        model.handle.set_option('winograd', mode)
        raw, post, nec = model.handle.segment_images(imgs, want_raw=True)
        raw_mis = [int((raw[i] != refs[i][0]).sum()) for i in range(n)]
        post_mis = [int((post[i] != refs[i][1]).sum()) for i in range(n)]
        dn = [int(nec[i]) - refs[i][2] for i in range(n)]
        exact = all(np.array_equal(postproc.meta_inference(raw[i].astype(np.int64)), post[i]) for i in range(min(n, 2)))
        out[tag] = {'raw_mismatch_px': raw_mis, 'post_mismatch_px': post_mis, 'delta_n_ec': dn,
                    'images_with_any_raw_mismatch': int(sum(v > 0 for v in raw_mis)),
                    'images_with_csv_difference': int(sum(v != 0 for v in dn)),
                    'raw_mismatch_px_per_100_images': round(100.0 * sum(raw_mis) / n, 1),
                    'post_mismatch_px_per_100_images': round(100.0 * sum(post_mis) / n, 1),
                    'abs_delta_n_ec_per_100_images': round(100.0 * sum(abs(v) for v in dn) / n, 1),
                    'integer_stages_bit_exact_on_device_raw_labels': bool(exact)}
    model.handle.set_option('winograd', 2)
    model.handle.close()
    return out


def adjudicate(truth_path):
    """Device vs float32 oracle vs float64 adjudicator on the cached CPU results of tools/label_truth.py.  For every 3x3
    kernel: raw-label pixels differing from the oracle / from the truth, who is right where device and oracle disagree,
    the decision margin (float64) of the device's wrong pixels, max |dp| against the float64 probabilities, and the
    consequences after clean-up (labels, n_ec, CSV rows)."""
    from ecseg_amd import synth
    from ecseg_amd._lib import LIB_PATH
    from ecseg_amd.model import MetasegModel
    from oracle import tiling
    from tools import label_truth
    z = np.load(truth_path)
    base, seed0, n = int(z['base']), int(z['seed0']), len(z['raw32'])
    mname = str(z['model'])
    if mname not in ('random', 'smooth') and not os.path.isabs(mname):
        mname = os.path.join(ROOT, mname)
    cfg, weights = label_truth.model_weights(mname, base)
    raw32, raw64, post64, nec32, nec64, margin = z['raw32'], z['raw64'], z['post64'], z['nec32'], z['nec64'], z['margin64']
    model = MetasegModel(cfg, weights, device=0)
    imgs = np.stack([synth.dapi_image(seed0 + i, H, W) for i in range(n)])
    pos = tiling.patch_positions(H, W)
    win = tiling.extract_patches(imgs[0][..., None], pos)[:len(z['p64'])]
    o_wrong = raw32 != raw64
    out = {'library': os.path.basename(LIB_PATH), 'images': n, 'pixels': int(n * H * W), 'unet_base': base, 'model': str(z['model']),
           'oracle_float32_vs_float64': {
               'raw_px_wrong': int(o_wrong.sum()), 'raw_px_wrong_per_100_images': round(100.0 * o_wrong.sum() / n, 1),
               'images_with_n_ec_difference': int((nec32 != nec64).sum()), 'max_abs_dp': float(z['p32err'].max()),
               'post_px_differing': int((z['post32'] != post64).sum())},
           'kernels': {}}
    for mode, tag in ((0, 'direct'), (1, 'winograd_f2x2'), (2, 'winograd_f4x4')):
        model.handle.set_option('winograd', mode)
        raw, post, nec = model.handle.segment_images(imgs, want_raw=True)
        d_wrong = raw != raw64
        dis = raw != raw32
        probs = model.predict_on_batch(win)
        out['kernels'][tag] = {
            'raw_px_differing_from_oracle32': int(dis.sum()), 'per_100_images_vs_oracle32': round(100.0 * dis.sum() / n, 1),
            'raw_px_wrong_vs_float64': int(d_wrong.sum()), 'per_100_images_vs_float64': round(100.0 * d_wrong.sum() / n, 1),
            'where_device_and_oracle32_disagree': {'device_agrees_with_float64': int((dis & ~d_wrong).sum()),
                                                   'oracle32_agrees_with_float64': int((dis & ~o_wrong).sum()),
                                                   'neither': int((dis & d_wrong & o_wrong).sum())},
            'both_wrong_same_label': int((d_wrong & o_wrong & ~dis).sum()),
            'margin_of_device_wrong_px_1e-7': {'max': int(margin[d_wrong].max()) if d_wrong.any() else 0,
                                               'p50': float(np.median(margin[d_wrong])) if d_wrong.any() else 0.0},
            'max_abs_dp_vs_float64': float(np.abs(probs.astype(np.float64) - z['p64']).max()),
            'rms_dp_vs_float64': float(np.sqrt(((probs.astype(np.float64) - z['p64']) ** 2).mean())),
            'post_px_differing_from_float64': int((post != post64).sum()),
            'images_with_n_ec_difference_vs_float64': int((np.asarray(nec) != nec64).sum()),
            'images_with_n_ec_difference_vs_oracle32': int((np.asarray(nec) != nec32).sum()),
            'worst_image_raw_px_vs_oracle32': int(dis.sum(axis=(1, 2)).max()),
            'worst_image_raw_px_vs_float64': int(d_wrong.sum(axis=(1, 2)).max())}
    out['oracle_float32_vs_float64']['margin_of_wrong_px_1e-7'] = {'max': int(margin[o_wrong].max()) if o_wrong.any() else 0}
    model.handle.set_option('winograd', 2)
    model.handle.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--truth', action='append', default=[], help='cache file of tools/label_truth.py: adjudicate against its '
                    'float64 evaluation instead of fitting / running the CPU oracle here (repeatable)')
    ap.add_argument('--base', type=int, default=64)
    ap.add_argument('--fit-steps', type=int, default=150)
    ap.add_argument('--images', type=int, default=16)
    ap.add_argument('--out', default=None)
    ap.add_argument('--skip-random', action='store_true')
    ap.add_argument('--fit-device', default='cpu', help="'cuda': fit in a child process with torch on the GPU")
    ap.add_argument('--fit-batch', type=int, default=2)
    a = ap.parse_args()
    if a.truth:
        res = {'image_size': [H, W], 'what': 'device vs float32 CPU oracle vs float64 CPU evaluation (tools/label_truth.py)',
               'models': {os.path.splitext(os.path.basename(t))[0]: adjudicate(t) for t in a.truth}}
        text = json.dumps(res, indent=1)
        if a.out:
            open(a.out, 'w').write(text)
        print(text)
        return
    from ecseg_amd import synth
    from tools import fit_smooth_model
    res = {'image_size': [H, W], 'pixels_per_image': H * W, 'images': a.images, 'unet_base': a.base, 'models': {}}
    tmp = tempfile.mkdtemp()
    jobs = []
    t0 = time.time()
    if a.fit_device == 'cpu':
        cfg, w_fit = fit_smooth_model.fit(a.base, steps=a.fit_steps, batch=a.fit_batch, log=lambda s: print(s, file=sys.stderr))
    else:                                                        # child process: this one stays off the GPU until the CPU legs are done
        import subprocess
        fpath = os.path.join(tmp, 'fit.npz')
        subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fit_smooth_model.py'), '--base', str(a.base), '--steps',
                        str(a.fit_steps), '--device', a.fit_device, '--batch', str(a.fit_batch), '--out', fpath], check=True,
                       stdout=sys.stderr)
        cfg, w_fit = synth.unet_config(base=a.base), _load(fpath)
    jobs.append(('fitted_%d_steps' % a.fit_steps, w_fit))
    if not a.skip_random:
        jobs.append(('random_seed0', synth.unet_weights(cfg, seed=0)))
    staged = []
    for tag, w in jobs:                                          # all CPU work first: no process is started after HIP init
        path = os.path.join(tmp, tag + '.npz')
        np.savez(path, **{'%s/%d' % (k, i): arr for k, v in w.items() for i, arr in enumerate(v)})
        refs = cpu_refs(a.base, path, a.images)
        staged.append((tag, w, refs))
        print('%s: CPU oracle done (%.0f s)' % (tag, time.time() - t0), file=sys.stderr)
    for tag, w, refs in staged:
        r = device_modes(a.base, w, a.images, refs)
        r['n_ec_cpu'] = [x[2] for x in refs]
        r['class_fractions_cpu_raw'] = [round(float(v), 4) for v in
                                        np.bincount(np.concatenate([x[0].ravel() for x in refs]), minlength=4) / (a.images * H * W)]
        res['models'][tag] = r
    text = json.dumps(res, indent=1)
    if a.out:
        open(a.out, 'w').write(text)
    print(text)


if __name__ == '__main__':
    main()

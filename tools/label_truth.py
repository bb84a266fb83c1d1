#!/usr/bin/env python3
"""CPU half of the label-mismatch adjudication (VERDICT r02 #1a): for N full-size synthetic images evaluate the oracle
U-Net twice - in float32 (the comparison target of the GPU tests) and in float64 (the adjudicator) - and cache

    raw32      uint8 (N, H, W)   quantised argmax of the float32 oracle
    raw64      uint8 (N, H, W)   quantised argmax of float32(float64 evaluation) = what an exactly rounded network gives
    post32 / nec32               meta_inference + count of raw32
    post64 / nec64               ... of raw64
    margin64   uint8 (N, H, W)   min(255, floor(1e7 * distance of the float64 probabilities to the nearest point where the
                                 quantised argmax changes)): how hard a pixel is
    p64        float64 (K, 256, 256, 4) probabilities of the first K windows of image 0 (for max |dp| of each device kernel)
    p32err     max |p32 - p64| over everything evaluated (the float32 ORACLE's own distance from the truth)

under build/label_truth/<tag>.npz (git-ignored; travels to the GPU box with the snapshot).  Pure CPU, no GPU, no product
code except the synthetic generators.  tools/label_mismatch.py consumes the cache on the GPU box.

    python tools/label_truth.py --model random --base 64 --images 32 --procs 6
    python tools/label_truth.py --model smooth --base 64 --images 32
    python tools/label_truth.py --model build/label_truth/fit64_600.npz --tag fit64_600 --images 32
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
H, W = 1040, 1392
SEED0 = 5000
SMOOTH_HEAD_GAIN = 150.0          # softmax as decisive as a trained model's (mean top probability 0.97)
CACHE = os.path.join(ROOT, 'build', 'label_truth')


def load_weights(path):
    z = np.load(path)
    w = {}
    for k in z.files:
        name, i = k.rsplit('/', 1)
        w.setdefault(name, {})[int(i)] = z[k]
    return {n: [d[i] for i in sorted(d)] for n, d in w.items()}


def model_weights(model, base):
    from ecseg_amd import synth
    cfg = synth.unet_config(base=base)
    if model == 'random':
        return cfg, synth.unet_weights(cfg, seed=0)
    if model == 'smooth':                       # seeded smooth-output model (synth.unet_weights docstring)
        return cfg, synth.unet_weights(cfg, seed=0, smooth=True, head_gain=SMOOTH_HEAD_GAIN)
    return cfg, load_weights(model)


def decision_margin(p):
    """Distance (in probability units) of each pixel's 4 probabilities to the nearest change of the quantised argmax: the
    label changes when some q_c = rint(255 p_c) steps while the top two quantised values are within one level of each
    other.  Returned as the smallest |255 p_c - (k + 0.5)| / 255 over the classes whose step could change the winner."""
    s = p.astype(np.float64) * 255.0
    q = np.rint(s)
    top = np.sort(q, axis=-1)
    close = (top[..., -1] - top[..., -2]) <= 1                      # otherwise one rounding step cannot change the argmax
    frac = np.abs(s - np.floor(s) - 0.5)                            # distance to the rounding boundary, per class
    cand = np.where(q >= top[..., -1:] - 1, frac, np.inf)            # only classes within one level of the maximum matter
    d = cand.min(axis=-1) / 255.0
    return np.where(close, d, np.inf)


def _worker(job):
    model, base, idx, keep_windows = job
    import torch
    torch.set_num_threads(1)
    from ecseg_amd import synth
    from oracle import pipeline, postproc, tiling, unet
    cfg, w = model_weights(model, base)
    img = synth.dapi_image(idx, H, W)
    pos = tiling.patch_positions(H, W)
    patches = tiling.extract_patches(img[..., None], pos)
    p32 = np.concatenate([unet.forward(cfg, w, patches[i:i + 7]) for i in range(0, len(patches), 7)])
    p64 = np.concatenate([unet.forward(cfg, w, patches[i:i + 5], dtype=np.float64) for i in range(0, len(patches), 5)])
    err = float(np.abs(p32.astype(np.float64) - p64).max())
    t32 = p64.astype(np.float32)
    raw32 = pipeline.raw_labels_from_probs(p32, pos).astype(np.uint8)
    raw64 = pipeline.raw_labels_from_probs(t32, pos).astype(np.uint8)
    margin = tiling.stitch(np.minimum(decision_margin(p64) * 1e7, 255.0)[..., None].astype(np.float32), pos)[..., 0]
    post32 = postproc.meta_inference(raw32.astype(np.int64))
    post64 = postproc.meta_inference(raw64.astype(np.int64))
    return (idx, raw32, raw64, post32.astype(np.uint8), post64.astype(np.uint8), int(postproc.count_cc(post32 == 3)[0]),
            int(postproc.count_cc(post64 == 3)[0]), margin.astype(np.uint8), err, p64[:keep_windows] if keep_windows else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='random', help="'random' (seed-0 bench weights), 'smooth' (seeded smooth-output model) or a weights .npz (tools/fit_smooth_model.py)")
    ap.add_argument('--tag', default=None)
    ap.add_argument('--base', type=int, default=64)
    ap.add_argument('--images', type=int, default=32)
    ap.add_argument('--procs', type=int, default=6)
    ap.add_argument('--windows', type=int, default=6, help='float64 probability windows of image 0 to keep')
    a = ap.parse_args()
    tag = a.tag or ('%s_base%d' % (a.model, a.base) if a.model in ('random', 'smooth') else os.path.splitext(os.path.basename(a.model))[0])
    os.makedirs(CACHE, exist_ok=True)
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    t0 = time.time()
    jobs = [(a.model, a.base, SEED0 + i, a.windows if i == 0 else 0) for i in range(a.images)]
    out = []
    with ctx.Pool(a.procs) as pool:
        for r in pool.imap(_worker, jobs, chunksize=1):
            out.append(r)
            print('%s: image %d done (%.0f s), oracle32 vs truth: %d px, max |p32 - p64| %.2e'
                  % (tag, r[0], time.time() - t0, int((r[1] != r[2]).sum()), r[8]), flush=True)
    np.savez_compressed(os.path.join(CACHE, tag + '.npz'), base=a.base, seed0=SEED0, model=a.model,
                        raw32=np.stack([r[1] for r in out]), raw64=np.stack([r[2] for r in out]),
                        post32=np.stack([r[3] for r in out]), post64=np.stack([r[4] for r in out]),
                        nec32=np.array([r[5] for r in out]), nec64=np.array([r[6] for r in out]),
                        margin64=np.stack([r[7] for r in out]), p32err=np.array([r[8] for r in out]), p64=out[0][9])
    print('saved', os.path.join(CACHE, tag + '.npz'))


if __name__ == '__main__':
    main()

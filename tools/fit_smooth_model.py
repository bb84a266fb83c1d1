#!/usr/bin/env python3
"""Fit the canonical synthetic U-Net (ecseg_amd.synth.unet_config) for a few hundred steps on seeded synthetic scenes
(synth.dapi_image(..., with_labels=True)) so that its output is SMOOTH - nuclei / chromosome / ecDNA blobs instead of the
speckle a random-weight network produces.  Test / measurement infrastructure only (torch CPU autograd): metaseg.h5 is
not distributable, so this is the closest stand-in for "a trained model" when measuring how often fp32 summation
order flips a uint8-quantised argmax (tools/label_mismatch.py) and how the post-processing behaves on realistic label
maps.  Nothing here is used by the product path.

    python tools/fit_smooth_model.py [--base 64] [--steps 150] [--out model.npz]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def fit(base=64, depth=4, steps=150, crop=128, batch=2, seed=0, lr=2e-3, threads=None, log=None, device='cpu'):
    """-> (model_config, weights) in the same form as synth.unet_config / synth.unet_weights.  ``device='cuda'`` runs the
    autograd steps with torch on the GPU (test infrastructure; the product never uses torch for arithmetic)."""
    import torch
    import torch.nn.functional as F
    from ecseg_amd import synth
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(seed)
    cfg = synth.unet_config(base=base, depth=depth)
    w0 = synth.unet_weights(cfg, seed=seed, head_gain=1.0)
    layers = cfg['config']['layers']
    params = {}
    for name, arrs in w0.items():
        params[name] = [torch.tensor(a, requires_grad=True, device=device) for a in arrs]

    def forward(x):                                              # x: (N, 1, h, w) float 0..255 -> logits (N, 4, h, w)
        vals = {}
        for L in layers:
            cls, lc, name = L['class_name'], L['config'], L['config']['name']
            if cls == 'InputLayer':
                vals[name] = x
                continue
            ins = [vals[r[0]] for r in L['inbound_nodes'][0]]
            a = ins[0]
            if cls == 'Conv2D':
                k, b = params[name]
                y = F.conv2d(a, k.permute(3, 2, 0, 1), b, padding=k.shape[0] // 2)
                if lc['activation'] == 'relu':
                    y = F.relu(y)
            elif cls == 'Conv2DTranspose':
                k, b = params[name]
                y = F.conv_transpose2d(a, k.permute(3, 2, 0, 1), b, stride=2)
            elif cls == 'MaxPooling2D':
                y = F.max_pool2d(a, 2, 2)
            elif cls == 'Concatenate':
                y = torch.cat(ins, 1)
            else:
                raise NotImplementedError(cls)
            vals[name] = y
        return vals[cfg['config']['output_layers'][0][0]]

    rng = np.random.default_rng(seed)
    scenes = [synth.dapi_image(7000 + i, 512, 640, with_labels=True) for i in range(12)]
    flat = [t for ps in params.values() for t in ps]
    opt = torch.optim.Adam(flat, lr=lr)
    cw = torch.tensor([0.3, 1.0, 1.5, 3.0], device=device)
    t0 = time.time()
    for step in range(steps):
        xs, ys = [], []
        for _ in range(batch):
            g, lab = scenes[int(rng.integers(len(scenes)))]
            for _try in range(8):                                # prefer crops that contain objects
                y0, x0 = int(rng.integers(0, g.shape[0] - crop)), int(rng.integers(0, g.shape[1] - crop))
                if (lab[y0:y0 + crop, x0:x0 + crop] > 0).mean() > 0.03:
                    break
            xs.append(g[y0:y0 + crop, x0:x0 + crop]); ys.append(lab[y0:y0 + crop, x0:x0 + crop])
        x = torch.from_numpy(np.stack(xs).astype(np.float32))[:, None].to(device)
        y = torch.from_numpy(np.stack(ys).astype(np.int64)).to(device)
        loss = F.cross_entropy(forward(x), y, weight=cw)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if log and (step % 25 == 0 or step == steps - 1):
            log('step %d loss %.4f (%.1f s)' % (step, float(loss.detach()), time.time() - t0))
    weights = {name: [p.detach().cpu().numpy().copy() for p in ps] for name, ps in params.items()}
    return cfg, weights


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--base', type=int, default=64)
    ap.add_argument('--steps', type=int, default=150)
    ap.add_argument('--crop', type=int, default=128)
    ap.add_argument('--out', default=None)
    ap.add_argument('--device', default='cpu')
    ap.add_argument('--batch', type=int, default=2)
    a = ap.parse_args()
    cfg, w = fit(a.base, steps=a.steps, crop=a.crop, batch=a.batch, log=print, device=a.device)
    if a.out:
        np.savez(a.out, **{'%s/%d' % (k, i): arr for k, v in w.items() for i, arr in enumerate(v)})
        print('saved', a.out)


if __name__ == '__main__':
    main()

import sys, numpy as np
sys.path.insert(0, '.')
from ecseg_amd import keras_plan, synth
from ecseg_amd._lib import Handle
gpu = Handle(0)
base = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = synth.unet_config(base=base)
gpu.load_plan(keras_plan.build_plan(cfg, synth.unet_weights(cfg, seed=9)))
img = synth.dapi_image(40)[None]
gpu.set_option('unet_lanes', 1)
ref = gpu.segment_images(img, want_raw=True, want_probs=True)
ref2 = gpu.segment_images(img, want_raw=True, want_probs=True)
print('repeat identical', np.array_equal(ref[3], ref2[3]))
for lanes in (2, 3, 4, 5, 6, 7, 8, 3, 3):
    gpu.set_option('unet_lanes', lanes)
    got = gpu.segment_images(img, want_raw=True, want_probs=True)
    d = np.abs(got[3][0] - ref[3][0]).max(-1)
    ys, xs = np.nonzero(d)
    print('lanes', lanes, 'bounds', [35 * (l + 1) // lanes for l in range(lanes)], 'diff px', len(ys), 'max', d.max(),
          ('rows %d-%d cols %d-%d' % (ys.min(), ys.max(), xs.min(), xs.max())) if len(ys) else '')
    if len(ys):
        # windows: starts rows [0,206,412,618,784], cols [0,206,...]
        hs = [0, 206, 412, 618, 784]; ws = [0, 206, 412, 618, 824, 1030, 1136]
        cnt = {}
        for y, x in zip(ys[::50], xs[::50]):
            ci = max(i for i, w in enumerate(ws) if x >= w + (25 if i else 0)) if x >= 25 else 0
            ri = max(i for i, h in enumerate(hs) if y >= h + (25 if i else 0)) if y >= 25 else 0
            cnt[ci * 5 + ri] = cnt.get(ci * 5 + ri, 0) + 1
        print('   approx windows', sorted(cnt.items()))

import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from ecseg_amd import keras_plan, synth
from ecseg_amd._lib import Handle
cfg = synth.unet_config(base=64, up='upsample'); w = synth.unet_weights(cfg, seed=0)
imgs = np.stack([synth.dapi_image(i) for i in range(16)])
src = open(keras_plan.__file__).read()
mod = type(keras_plan)('kp0'); mod.__file__ = keras_plan.__file__
exec(compile(src.replace("if u['kind'] != 'upsample' or", "if True or u['kind'] != 'upsample' or"), keras_plan.__file__, 'exec'), mod.__dict__)
for tag, plan in (('lowered', keras_plan.build_plan(cfg, w)), ('as written', mod.build_plan(cfg, w))):
    h = Handle(0); h.load_plan(plan); h.segment_images(imgs)
    h.set_kernel_profiling(True); h.segment_images(imgs); h.conv_profile()
    recs = h.conv_launch_profile()
    print(tag, 'unet ms', h.timings()['unet'])
    for r in recs:
        o = plan.ops[r['op']]
        if o['op'] == keras_plan.OP_CONVT or (o['op'] == keras_plan.OP_CONV and o['kh'] == 2):
            print('   op %d kind %d k %d ms %.3f  alg TF/s %.1f exec TF/s %.1f' % (r['op'], r['kind'] & 255, o['kh'], r['ms'], r['flops'] / r['ms'] / 1e9, r['executed_flops'] / r['ms'] / 1e9))
    h.close()

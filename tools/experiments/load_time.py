#!/usr/bin/env python3
"""Start-up cost a `make metaseg` user sees before the first image: plan + upload + Winograd filter transforms of the
base-64 model, first and second segment call (GPU box)."""
import time, sys, os
sys.path.insert(0, os.getcwd())
t0=time.perf_counter()
from ecseg_amd import synth
from ecseg_amd.model import MetasegModel
import numpy as np
t1=time.perf_counter()
cfg = synth.unet_config(base=64); w = synth.unet_weights(cfg, seed=0)
t2=time.perf_counter()
m = MetasegModel(cfg, w, device=0)
t3=time.perf_counter()
img = synth.dapi_image(0)
t4=time.perf_counter()
m.segment(img)
t5=time.perf_counter()
m.segment(img)
t6=time.perf_counter()
print('import %.2f s, synth weights %.2f s, MetasegModel (plan + upload + filter transforms) %.2f s, first segment %.3f s, second %.3f s' % (t1-t0, t2-t1, t3-t2, t5-t4, t6-t5))

#!/usr/bin/env python3
"""Does the device thread burn a core while it waits for the GPU?  CPU time vs wall time of segment calls."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ecseg_amd import synth
from ecseg_amd.model import MetasegModel
cfg = synth.unet_config(base=16)
m = MetasegModel(cfg, synth.unet_weights(cfg, seed=0), device=0)
imgs = np.stack([synth.dapi_image(i) for i in range(32)])
if len(sys.argv) > 1:
    m.handle.set_option('blocking_wait', int(sys.argv[1]))
def threads():
    out = {}
    for t in os.listdir('/proc/self/task'):
        try:
            f = open('/proc/self/task/%s/stat' % t).read().rsplit(')', 1)[1].split()
            out[t] = ((int(f[11]) + int(f[12])) / os.sysconf('SC_CLK_TCK'), open('/proc/self/task/%s/comm' % t).read().strip())
        except OSError:
            pass
    return out
m.segment(imgs)
th0 = threads()
c0, w0 = time.process_time(), time.perf_counter()
for _ in range(5):
    m.segment(imgs)
c1, w1 = time.process_time(), time.perf_counter()
th1 = threads()
per = sorted(((th1[t][0] - th0.get(t, (0, ''))[0], th1[t][1], t == str(os.getpid())) for t in th1), reverse=True)[:4]
print(json.dumps({'threads_cpu_s(name, is_main)': per, 'wall_s': round(w1 - w0, 3), 'process_cpu_s': round(c1 - c0, 3), 'cpu_over_wall': round((c1 - c0) / (w1 - w0), 3),
                  'env': {k: v for k, v in os.environ.items() if k.startswith(('HIP_', 'GPU_', 'HSA_'))}}))

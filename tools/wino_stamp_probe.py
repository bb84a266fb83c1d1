import sys, numpy as np
sys.path.insert(0, '.')
import torch
from ecseg_amd import synth, keras_plan
from ecseg_amd._lib import Handle
h = Handle(0)
for cin, cout, hw in [(64,64,256),(128,128,128),(512,512,32),(1024,1024,16)]:
    cfg = {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, hw, hw, cin]}, 'inbound_nodes': []},
        {'class_name': 'Conv2D', 'name': 'c', 'config': {'name': 'c', 'filters': cout, 'kernel_size': [3, 3], 'strides': [1, 1], 'padding': 'same', 'activation': 'relu', 'use_bias': True}, 'inbound_nodes': [[['in', 0, 0, {}]]]}],
        'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}
    rng = np.random.default_rng(0)
    w = {'c': [(rng.normal(size=(3,3,cin,cout))*0.01).astype(np.float32), np.zeros(cout, np.float32)]}
    h.load_plan(keras_plan.build_plan(cfg, w))
    n = max(1, 280 * 256*256*64 // (hw*hw*cin))
    n = min(n, 280)
    x = rng.integers(0, 256, size=(n, hw, hw, cin), dtype=np.uint8)
    h.forward_patches(x); h.forward_patches(x)
    import os
    nw = 8 if os.environ.get('ECSEG_WINO_VARIANT') == '8' else 4
    d = h.debug_peek(nw * 8).reshape(nw, 8)
    nch = d[0,7]
    print('layer %d->%d @%d  n=%d chunks=%d' % (cin, cout, hw, n, nch))
    names = ['bar->top', 'issue loads', 'LDS reads', 'xform+MFMA', 'vmcnt wait', 'store+barrier']
    if nw == 8:
        names = ['M:mfma', 'M:vmcnt', 'M:barrier', 'R:read+xform', 'R:dma issue', 'R:barrier']
    for wv in range(nw):
        print('  wave %d total %8.0f per-chunk:' % (wv, d[wv,6]), ' '.join('%s %6.0f' % (names[i], d[wv,i]/max(nch,1)) for i in range(6)), ' sum/chunk %.0f' % (d[wv,:6].sum()/max(nch,1)))

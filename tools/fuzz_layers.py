#!/usr/bin/env python3
"""Randomised layer-graph parity: small random Keras graphs (convolutions of several kernel sizes / strides / paddings,
pooling, transposed convolutions, up-sampling, concatenation, BatchNormalization, 1x1 heads; channel counts and extents
chosen so that every convolution kernel - Winograd F(4x4), F(2x2), direct MFMA, small-Cin, generic - and every fusion gets
hit) through the device under all three `winograd` modes, against the CPU oracle (oracle/unet.py), tolerance 1e-3 relative to
the output range.  Runs for --seconds; exit code 1 on any mismatch.

    python tools/fuzz_layers.py --seconds 300 [--seed0 0]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CH = (4, 8, 12, 16, 24, 32, 40, 64, 96, 128)
HW = (16, 32, 48, 64, 20, 36, 80)
CH_WIDE = (12, 36, 44, 64, 100, 160, 192, 224, 288, 320)       # --wide: Cin % 8 == 4 tails, Cout % 64 == 32, several channel blocks
HW_WIDE = (16, 32, 48, 96, 112, 128, 24, 40)


def _L(cls, name, inb, **c):
    return {'class_name': cls, 'name': name, 'config': dict(c, name=name),
            'inbound_nodes': [[[i, 0, 0, {}] for i in inb]] if inb else []}


def random_graph(rng, seed_kind=0, wide=False):
    CH, HW = (CH_WIDE, HW_WIDE) if wide else (globals()['CH'], globals()['HW'])
    """-> (model_config, weights dict).  A small encoder / decoder with one skip; every choice is random."""
    h = int(rng.choice(HW)); w = int(rng.choice(HW))
    cin = int(rng.choice((1, 3, 4, 8, 16, 32)))
    layers = [_L('InputLayer', 'in', [], batch_input_shape=[None, h, w, cin])]
    weights = {}
    shape = {'in': (h, w, cin)}

    def conv(name, src, cout, k, stride=1, padding='same', act='relu', bias=True):
        hh, ww, cc = shape[src]
        layers.append(_L('Conv2D', name, [src], filters=cout, kernel_size=[k, k], strides=[stride, stride], padding=padding,
                         activation=act, use_bias=bias))
        ws = [(rng.normal(size=(k, k, cc, cout)) / np.sqrt(k * k * cc)).astype(np.float32)]
        if bias:
            ws.append((rng.normal(size=cout) * 0.1).astype(np.float32))
        weights[name] = ws
        if padding == 'same':
            shape[name] = (-(-hh // stride), -(-ww // stride), cout)
        else:
            shape[name] = ((hh - k) // stride + 1, (ww - k) // stride + 1, cout)
        return name

    # first layer scales the uint8 input down
    c0 = int(rng.choice(CH))
    if seed_kind == 1 and h >= 32 and w >= 32:                 # classifier-style stem: strided and / or 'valid' convolution
        prev = conv('c0', 'in', c0, int(rng.choice((3, 5, 2))), stride=int(rng.choice((1, 2, 2))),
                    padding=str(rng.choice(('same', 'valid'))), act='relu')
    else:
        prev = conv('c0', 'in', c0, int(rng.choice((1, 3, 3, 5))), act=str(rng.choice(('relu', 'linear', 'tanh'))))
    weights['c0'][0] = weights['c0'][0] / 128.0
    if rng.random() < 0.3:
        cc = shape[prev][2]
        layers.append(_L('BatchNormalization', 'bn0', [prev], axis=[3], epsilon=1e-3, center=True, scale=True))
        weights['bn0'] = [rng.uniform(0.5, 1.5, cc).astype(np.float32), (rng.normal(size=cc) * 0.1).astype(np.float32),
                          (rng.normal(size=cc) * 0.1).astype(np.float32), rng.uniform(0.5, 1.5, cc).astype(np.float32)]
        shape['bn0'] = shape[prev]
        prev = 'bn0'
    prev = conv('c1', prev, int(rng.choice(CH)), 3, act=str(rng.choice(('relu', 'relu', 'sigmoid', 'linear'))))
    skip = prev
    hh, ww, cc = shape[prev]
    if hh % 2 == 0 and ww % 2 == 0 and rng.random() < 0.8:
        pool = str(rng.choice(('MaxPooling2D', 'MaxPooling2D', 'AveragePooling2D')))
        layers.append(_L(pool, 'p', [prev], pool_size=[2, 2], strides=[2, 2], padding='valid'))
        shape['p'] = (hh // 2, ww // 2, cc)
        prev = conv('c2', 'p', int(rng.choice(CH)), int(rng.choice((3, 3, 3, 1, 2))), act='relu')
        prev = conv('c3', prev, int(rng.choice(CH)), 3, act=str(rng.choice(('relu', 'elu'))), bias=bool(rng.random() < 0.8))
        h2, w2, c2 = shape[prev]
        if rng.random() < 0.6:
            cu = int(rng.choice(CH))
            kt = int(rng.choice((2, 2, 3, 4)))                  # 3 / 4: kernel larger than the stride (sub-pixel MFMA form)
            layers.append(_L('Conv2DTranspose', 'up', [prev], filters=cu, kernel_size=[kt, kt], strides=[2, 2], padding='same',
                             activation=str(rng.choice(('linear', 'relu'))), use_bias=True))
            weights['up'] = [(rng.normal(size=(kt, kt, cu, c2)) / np.sqrt(c2)).astype(np.float32),
                             (rng.normal(size=cu) * 0.1).astype(np.float32)]
            shape['up'] = (h2 * 2, w2 * 2, cu)
        else:
            layers.append(_L('UpSampling2D', 'up', [prev], size=[2, 2], interpolation=str(rng.choice(('nearest', 'bilinear')))))
            shape['up'] = (h2 * 2, w2 * 2, c2)
        if shape['up'][:2] == shape[skip][:2]:
            order = [skip, 'up'] if rng.random() < 0.5 else ['up', skip]
            layers.append(_L('Concatenate', 'cat', order, axis=-1))
            shape['cat'] = (hh, ww, shape['up'][2] + shape[skip][2])
            prev = 'cat'
        else:
            prev = 'up'
        prev = conv('c4', prev, int(rng.choice(CH)), 3, act='relu')
    last = conv('c5', prev, int(rng.choice((32, 64, 64, 16, 24))), 3, act='relu')
    ncls = int(rng.choice((2, 3, 4, 4, 5)))
    if seed_kind == 1 and rng.random() < 0.7:                  # classifier tail: pooling / flatten -> Dense -> Dense
        hh, ww, cc = shape[last]
        if rng.random() < 0.5 or hh * ww * cc > 40000:
            pool = str(rng.choice(('GlobalAveragePooling2D', 'GlobalMaxPooling2D')))
            layers.append(_L(pool, 'gp', [last]))
            feat = cc
        else:
            layers.append(_L('Flatten', 'gp', [last]))
            feat = hh * ww * cc
        nh = int(rng.choice((8, 16, 33, 64)))
        layers.append(_L('Dense', 'd0', ['gp'], units=nh, activation='relu', use_bias=True))
        weights['d0'] = [(rng.normal(size=(feat, nh)) / np.sqrt(feat)).astype(np.float32), (rng.normal(size=nh) * 0.1).astype(np.float32)]
        layers.append(_L('Dense', 'head', ['d0'], units=ncls, activation=str(rng.choice(('softmax', 'sigmoid'))), use_bias=True))
        weights['head'] = [(rng.normal(size=(nh, ncls)) / np.sqrt(nh)).astype(np.float32), (rng.normal(size=ncls) * 0.1).astype(np.float32)]
        head = 'head'
    else:
        head = conv('head', last, ncls, 1, act=str(rng.choice(('softmax', 'softmax', 'sigmoid'))))
    cfg = {'class_name': 'Functional', 'config': {'name': 'fuzz', 'layers': layers, 'input_layers': [['in', 0, 0]],
                                                 'output_layers': [[head, 0, 0]]}}
    return cfg, weights, shape['in']


ACTS_R5 = ('relu', 'relu6', 'selu', 'softplus', 'softsign', 'swish', 'gelu', 'hard_sigmoid', 'elu', 'tanh', 'sigmoid', 'linear')


def random_graph_r5(rng):
    """-> (model_config, weights, input shape): a chain of random blocks from the round-5 vocabulary - dilated / grouped / odd-tap
    convolutions, DepthwiseConv2D / SeparableConv2D, squeeze-and-excite Multiply, merge layers with broadcasting, PReLU,
    LayerNormalization, Normalization, 'same' pooling, the new activation names - optionally wrapped into a nested sub-model and
    given a second output."""
    h = int(rng.choice((16, 24, 32, 40, 33))); w = int(rng.choice((16, 32, 48, 20, 37)))
    c = int(rng.choice((4, 8, 16, 24, 32, 64)))
    layers, weights, shape = [_L('InputLayer', 'in', [], batch_input_shape=[None, h, w, c])], {}, {'in': (h, w, c)}
    he = lambda *s: (rng.normal(size=s) / np.sqrt(np.prod(s[:-1]))).astype(np.float32)

    def add(cls, name, inb, out_shape, ws=None, **cfg):
        layers.append(_L(cls, name, inb, **cfg))
        shape[name] = out_shape
        if ws is not None:
            weights[name] = ws
        return name

    def out_hw(n, k, s, d, pad):
        e = (k - 1) * d + 1
        return -(-n // s) if pad == 'same' else (n - e) // s + 1

    prev = 'in'
    for b in range(int(rng.integers(3, 7))):
        hh, ww, cc = shape[prev]
        kind = str(rng.choice(('conv', 'conv', 'dw', 'sep', 'se', 'merge', 'prelu', 'ln', 'norm', 'pool', 'act')))
        n = 'b%d' % b
        if kind == 'conv':
            k = int(rng.choice((1, 3, 3, 5, 2))); d = int(rng.choice((1, 1, 2, 3, 6))) if k > 1 else 1
            s = 1 if d > 1 else int(rng.choice((1, 1, 2, 3)))
            pad = str(rng.choice(('same', 'same', 'valid')))
            if out_hw(hh, k, s, d, pad) < 4 or out_hw(ww, k, s, d, pad) < 4:
                pad = 'same'
            g = int(rng.choice([q for q in (1, 1, 1, 2, 4, cc) if cc % q == 0]))
            f = int(rng.choice((8, 16, 24, 32, 64, 96)))
            f = -(-f // g) * g
            prev = add('Conv2D', n, [prev], (out_hw(hh, k, s, d, pad), out_hw(ww, k, s, d, pad), f), [he(k, k, cc // g, f), he(f)], filters=f,
                       kernel_size=[k, k], strides=[s, s], dilation_rate=[d, d], groups=g, padding=pad, activation=str(rng.choice(ACTS_R5)), use_bias=True)
        elif kind in ('dw', 'sep'):
            k = int(rng.choice((3, 3, 5))); d = int(rng.choice((1, 1, 2))); s = 1 if d > 1 else int(rng.choice((1, 1, 2)))
            m = int(rng.choice((1, 1, 1, 2)))
            dk = he(k, k, cc, m) * np.float32(np.sqrt(cc))
            if kind == 'dw':
                prev = add('DepthwiseConv2D', n, [prev], (out_hw(hh, k, s, d, 'same'), out_hw(ww, k, s, d, 'same'), cc * m), [dk, he(cc * m)],
                           kernel_size=[k, k], strides=[s, s], dilation_rate=[d, d], depth_multiplier=m, padding='same',
                           activation=str(rng.choice(ACTS_R5)), use_bias=True)
            else:
                f = int(rng.choice((8, 16, 32, 40)))
                prev = add('SeparableConv2D', n, [prev], (out_hw(hh, k, s, d, 'same'), out_hw(ww, k, s, d, 'same'), f),
                           [dk, he(1, 1, cc * m, f), he(f)], filters=f, kernel_size=[k, k], strides=[s, s], dilation_rate=[d, d],
                           depth_multiplier=m, padding='same', activation=str(rng.choice(ACTS_R5)), use_bias=True)
        elif kind == 'se':
            r = max(cc // 4, 1)
            g = add('GlobalAveragePooling2D', n + '_gap', [prev], (1, 1, cc), keepdims=True)
            a = add('Conv2D', n + '_r', [g], (1, 1, r), [he(1, 1, cc, r), he(r)], filters=r, kernel_size=[1, 1], strides=[1, 1], padding='same',
                    activation='relu', use_bias=True)
            e = add('Conv2D', n + '_e', [a], (1, 1, cc), [he(1, 1, r, cc), he(cc)], filters=cc, kernel_size=[1, 1], strides=[1, 1], padding='same',
                    activation=str(rng.choice(('sigmoid', 'hard_sigmoid'))), use_bias=True)
            prev = add('Multiply', n, [prev, e] if rng.random() < 0.5 else [e, prev], (hh, ww, cc))
        elif kind == 'merge':
            other = add('Conv2D', n + '_o', [prev], (hh, ww, 1 if rng.random() < 0.3 else cc), None, filters=1, kernel_size=[1, 1], strides=[1, 1],
                        padding='same', activation='tanh', use_bias=True)
            co = shape[other][2]
            layers[-1]['config']['filters'] = co
            weights[other] = [he(1, 1, cc, co), he(co)]
            prev = add(str(rng.choice(('Add', 'Multiply', 'Subtract', 'Maximum', 'Minimum', 'Average'))), n, [prev, other], (hh, ww, cc))
        elif kind == 'prelu':
            shared = [None, [1, 2], [1, 2], [1], [2]][int(rng.integers(0, 5))]
            shp = [1 if shared and (a + 1) in shared else v for a, v in enumerate((hh, ww, cc))]
            prev = add('PReLU', n, [prev], (hh, ww, cc), [rng.uniform(-.4, .4, shp).astype(np.float32)], shared_axes=shared)
        elif kind == 'ln':
            ws, center, scale = [], bool(rng.random() < 0.8), bool(rng.random() < 0.8)
            if scale:
                ws.append(rng.uniform(.5, 1.5, cc).astype(np.float32))
            if center:
                ws.append(he(cc))
            prev = add('LayerNormalization', n, [prev], (hh, ww, cc), ws, axis=[3], epsilon=float(rng.choice((1e-3, 1e-5))), center=center, scale=scale)
        elif kind == 'norm':
            prev = add('Normalization', n, [prev], (hh, ww, cc), [he(cc), rng.uniform(.5, 2, cc).astype(np.float32), np.array(3, np.int64)], axis=[-1],
                       mean=None, variance=None)
        elif kind == 'pool':
            k = int(rng.choice((2, 3))); s = int(rng.choice((1, 2)))
            if -(-hh // s) < 4 or -(-ww // s) < 4:
                s = 1
            prev = add(str(rng.choice(('MaxPooling2D', 'AveragePooling2D'))), n, [prev], (-(-hh // s), -(-ww // s), cc), pool_size=[k, k],
                       strides=[s, s], padding='same')
        else:
            prev = add('Activation', n, [prev], (hh, ww, cc), activation=str(rng.choice(ACTS_R5)))
    hh, ww, cc = shape[prev]
    ncls = int(rng.choice((2, 3, 4)))
    head = add('Conv2D', 'head', [prev], (hh, ww, ncls), [he(1, 1, cc, ncls) * 3, he(ncls)], filters=ncls, kernel_size=[1, 1], strides=[1, 1],
               padding='same', activation=str(rng.choice(('softmax', 'sigmoid'))), use_bias=True)
    outs = [[head, 0, 0]]
    if rng.random() < 0.3:
        add('GlobalMaxPooling2D', 'aux', [prev], (1, 1, cc))
        outs.append(['aux', 0, 0])
    cfg = {'class_name': 'Functional', 'config': {'name': 'fuzz5', 'layers': layers, 'input_layers': [['in', 0, 0]], 'output_layers': outs}}
    if rng.random() < 0.35 and len(outs) == 1:
        # the whole graph as a nested sub-model of an outer model that rescales its input and post-processes the output
        inner = dict(cfg, name='sub')
        inner['config'] = dict(cfg['config'], name='sub')
        outer_layers = [_L('InputLayer', 'oin', [], batch_input_shape=[None, h, w, c]),
                        _L('Rescaling', 'resc', ['oin'], scale=0.5, offset=0.1),
                        dict(inner, inbound_nodes=[[['resc', 0, 0, {}]]]),
                        _L('Activation', 'oact', ['sub'], activation='linear')]
        cfg = {'class_name': 'Functional', 'config': {'name': 'outer', 'layers': outer_layers, 'input_layers': [['oin', 0, 0]],
                                                     'output_layers': [['oact', 0, 0]]}}
        weights = {'sub': weights}
    return cfg, weights, (h, w, c)


def channels_first_twin(cfg):
    """The channels_first edition of a flat channels_last Functional config: (N, C, H, W) input, data_format on every spatial layer,
    channel axis 1; a softmax fused into a convolution becomes a Softmax(axis=1) layer behind it (Keras applies a FUSED softmax
    over the last axis - W for such tensors)."""
    import json
    c = json.loads(json.dumps(cfg))
    extra = []
    for L in c['config']['layers']:
        lc = L['config']
        if L['class_name'] == 'InputLayer':
            b = lc['batch_input_shape']
            lc['batch_input_shape'] = [b[0], b[3], b[1], b[2]]
        if L['class_name'] in ('Conv2D', 'Conv2DTranspose', 'MaxPooling2D', 'UpSampling2D', 'DepthwiseConv2D', 'SeparableConv2D', 'AveragePooling2D'):
            lc['data_format'] = 'channels_first'
            if lc.get('activation') == 'softmax':
                lc['activation'] = 'linear'
                extra.append({'class_name': 'Softmax', 'name': lc['name'] + '_sm', 'config': {'name': lc['name'] + '_sm', 'axis': 1},
                              'inbound_nodes': [[[lc['name'], 0, 0, {}]]]})
                c['config']['output_layers'] = [[lc['name'] + '_sm', 0, 0] if o[0] == lc['name'] else o for o in c['config']['output_layers']]
        if L['class_name'] == 'Concatenate':
            lc['axis'] = 1
        if L['class_name'] == 'BatchNormalization':
            lc['axis'] = [1]
    c['config']['layers'] += extra
    return c


def random_graph_r6(rng):
    """-> (model_config, weights, input shape, channels_first): the loader corners of round 6 in one chain - Conv2D with per-axis
    strides / dilation rates / taps (scalar kernel), a convolution CALLED TWICE (shared weights: one layer entry with two inbound
    nodes, listed before the layer that feeds its second call, as Keras lists it), plain 3x3 convolutions and pools between them
    (so that the fast kernels see the odd shapes), BatchNormalization behind a shared call; half of the graphs as their
    channels_first twin (the device then takes and returns (N, C, H, W))."""
    h = int(rng.choice((16, 24, 32, 40, 33, 50))); w = int(rng.choice((16, 32, 48, 20, 37, 64)))
    c = int(rng.choice((1, 3, 4, 8, 16, 32)))
    layers, weights, shape = [_L('InputLayer', 'in', [], batch_input_shape=[None, h, w, c])], {}, {'in': (h, w, c)}
    he = lambda *s: (rng.normal(size=s) / np.sqrt(np.prod(s[:-1]))).astype(np.float32)

    def ext(n, k, s, d, pad):
        e = (k - 1) * d + 1
        return -(-n // s) if pad == 'same' else (n - e) // s + 1

    def conv(name, src, f, kk=(3, 3), st=(1, 1), dil=(1, 1), pad='same', act='relu'):
        hh, ww, cc = shape[src]
        layers.append(_L('Conv2D', name, [src], filters=f, kernel_size=list(kk), strides=list(st), dilation_rate=list(dil), padding=pad,
                         activation=act, use_bias=True))
        weights[name] = [he(kk[0], kk[1], cc, f), he(f)]
        shape[name] = (ext(hh, kk[0], st[0], dil[0], pad), ext(ww, kk[1], st[1], dil[1], pad), f)
        return name

    prev = conv('c0', 'in', int(rng.choice((8, 16, 32))), act='relu')
    weights['c0'][0] = weights['c0'][0] / 64.0
    for b in range(int(rng.integers(2, 5))):
        hh, ww, cc = shape[prev]
        kind = str(rng.choice(('aniso', 'aniso', 'shared', 'conv', 'pool')))
        n = 'b%d' % b
        if kind == 'aniso':
            kk = [(3, 3), (3, 3), (1, 3), (3, 1), (2, 3), (1, 5), (1, 1)][int(rng.integers(0, 7))]
            if rng.random() < 0.5:
                st, dil = [(2, 1), (1, 2), (1, 3), (3, 2), (3, 1)][int(rng.integers(0, 5))], (1, 1)
            else:
                st, dil = (1, 1), [(2, 3), (1, 2), (3, 1), (2, 1), (1, 4)][int(rng.integers(0, 5))]
            pad = str(rng.choice(('same', 'same', 'valid')))
            if ext(hh, kk[0], st[0], dil[0], pad) < 4 or ext(ww, kk[1], st[1], dil[1], pad) < 4:
                pad = 'same'
            if ext(hh, kk[0], st[0], dil[0], pad) < 2 or ext(ww, kk[1], st[1], dil[1], pad) < 2:
                st, dil, pad = (1, 1), (1, 1), 'same'
            prev = conv(n, prev, int(rng.choice((8, 16, 24, 32, 64))), kk, st, dil, pad, act=str(rng.choice(('relu', 'linear', 'tanh', 'elu'))))
        elif kind == 'shared':
            # y0 = L(x); a = tanh-ish(y0); y1 = L(a); out = y0 + y1 (+ a BatchNormalization called on the sum)
            k = int(rng.choice((3, 3, 1)))
            layers.append({'class_name': 'Conv2D', 'name': n, 'config': dict(name=n, filters=cc, kernel_size=[k, k], strides=[1, 1], padding='same',
                                                                             activation=str(rng.choice(('relu', 'linear'))), use_bias=True),
                           'inbound_nodes': [[[prev, 0, 0, {}]], [[n + '_a', 0, 0, {}]]]})
            weights[n] = [he(k, k, cc, cc), he(cc)]
            layers.append(_L('Activation', n + '_a', [n], activation=str(rng.choice(('tanh', 'sigmoid', 'relu')))))
            layers.append({'class_name': 'Add', 'name': n + '_s', 'config': {'name': n + '_s'}, 'inbound_nodes': [[[n, 0, 0, {}], [n, 1, 0, {}]]]})
            shape[n + '_s'] = (hh, ww, cc)
            prev = n + '_s'
            if rng.random() < 0.5:
                layers.append(_L('BatchNormalization', n + '_bn', [prev], axis=[3], epsilon=1e-3, center=True, scale=True))
                weights[n + '_bn'] = [rng.uniform(.5, 1.5, cc).astype(np.float32), he(cc), he(cc), rng.uniform(.5, 1.5, cc).astype(np.float32)]
                shape[n + '_bn'] = (hh, ww, cc)
                prev = n + '_bn'
        elif kind == 'conv':
            prev = conv(n, prev, int(rng.choice((16, 32, 64, 96))), act='relu')
        elif hh >= 8 and ww >= 8 and hh % 2 == 0 and ww % 2 == 0:
            layers.append(_L('MaxPooling2D', n, [prev], pool_size=[2, 2], strides=[2, 2], padding='valid'))
            shape[n] = (hh // 2, ww // 2, cc)
            prev = n
    head = conv('head', prev, int(rng.choice((2, 3, 4))), (1, 1), act=str(rng.choice(('softmax', 'sigmoid'))))
    weights['head'][0] = weights['head'][0] * 3
    cfg = {'class_name': 'Functional', 'config': {'name': 'fuzz6', 'layers': layers, 'input_layers': [['in', 0, 0]], 'output_layers': [[head, 0, 0]]}}
    return cfg, weights, (h, w, c), bool(rng.random() < 0.5)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=300)
    ap.add_argument('--seed0', type=int, default=0)
    ap.add_argument('--seeds', default=None)
    ap.add_argument('--wide', action='store_true', help='wider layers / larger extents')
    ap.add_argument('--vocab', action='store_true', help='graphs from the round-5 vocabulary (random_graph_r5)')
    ap.add_argument('--loader', action='store_true', help='the loader corners of round 6 (random_graph_r6): per-axis strides / dilation, shared layers, channels_first')
    a = ap.parse_args()
    import torch  # noqa: F401
    from ecseg_amd.model import MetasegModel
    from oracle import unet as oracle_unet
    t0 = time.time()
    seed = a.seed0
    todo = [int(x) for x in a.seeds.split(',')] if a.seeds else None
    n_graphs = fails = 0
    worst = 0.0
    while time.time() - t0 < a.seconds:
        if todo is not None:
            if not todo:
                break
            seed = todo.pop(0)
        rng = np.random.default_rng(5 * 10 ** 6 + seed)
        cf = False
        if a.loader:
            cfg, weights, (h, w, cin), cf = random_graph_r6(rng)
        elif a.vocab:
            cfg, weights, (h, w, cin) = random_graph_r5(rng)
        else:
            cfg, weights, (h, w, cin) = random_graph(rng, seed_kind=1 if seed >= 10 ** 5 else 0, wide=a.wide)
        n = int(rng.integers(1, 5))
        x = rng.integers(0, 256, size=(n, h, w, cin), dtype=np.uint8)
        if a.vocab:
            x = (x.astype(np.float32) - 128) / 64            # (these graphs carry no input scaling of their own)
        out_sel = int(rng.integers(0, len(cfg['config']['output_layers'])))
        want = oracle_unet.forward(cfg, weights, x.astype(np.float32), output=out_sel)
        if cf:
            # the channels_first twin with the SAME weights: the oracle evaluates it natively in (N, C, H, W) and must agree with the
            # channels_last graph; the device is then fed (N, C, H, W) and checked against the transposed result
            cfg = channels_first_twin(cfg)
            x = np.ascontiguousarray(np.moveaxis(x, -1, 1))
            want_cf = oracle_unet.forward(cfg, weights, x.astype(np.float32), output=out_sel)
            if float(np.abs(np.moveaxis(want_cf, 1, -1) - want).max()) > 2e-4 * max(1.0, float(np.abs(want).max())):
                print('ORACLE MISMATCH seed %d: channels_first twin differs from the channels_last graph' % seed, flush=True)
                fails += 1
            want = want_cf
        scale = max(1.0, float(np.abs(want).max()))
        try:
            m = MetasegModel(cfg, weights, device=0, output=out_sel)
            for mode in (2, 3, 1, 0):            # 3: the bf16x3 split kernels where they apply (round 6)
                m.handle.set_option('winograd', mode)
                for fuse in (1, 0):
                    m.handle.set_option('fuse_pool', fuse)
                    m.handle.set_option('fuse_head', fuse)
                    got = m.predict_on_batch(x) if cf else m.handle.forward_patches(x)
                    err = float(np.abs(got - want).max()) / scale
                    worst = max(worst, err)
                    if not np.isfinite(got).all() or err > 1e-3:
                        ll = cfg['config']['layers']
                        ll = next((L['config']['layers'] for L in ll if L['class_name'] == 'Functional'), ll)
                        desc = ' '.join('%s:%s' % (L['class_name'][:6], L['config'].get('filters', '')) for L in ll)
                        print('FAIL seed %d mode %d fuse %d err %.3e  in %s x%d  %s' % (seed, mode, fuse, err, (h, w, cin), n, desc), flush=True)
                        fails += 1
            del m
        except Exception as e:  # a graph the plan rejects is a finding too
            print('ERROR seed %d: %s: %s' % (seed, type(e).__name__, e), flush=True)
            fails += 1
        n_graphs += 1
        seed += 1
    print('layer fuzz: %d random graphs x 4 kernel modes x 2 fusion settings, worst relative error %.2e, %d failure(s), %.0f s'
          % (n_graphs, worst, fails, time.time() - t0), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()

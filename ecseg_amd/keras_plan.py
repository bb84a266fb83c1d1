"""Lower a Keras ``model_config`` (the JSON stored in metaseg.h5) to the kernel-level plan that
``ecseg_model_load`` takes (include/ecseg_hip.h).

Stands in for the graph reconstruction inside ``tf.keras.models.load_model`` (reference src/utils.py:27-33):
the reference never states the metaseg architecture in code, so this accepts the layer vocabulary a Keras
U-Net can be written in rather than one fixed topology.

Design points
* tensors are NHWC float32 *views* (buffer, channel offset, channel stride): the producers of a
  ``Concatenate``'s inputs write straight into the concatenated buffer, so concatenation is free;
* buffers are re-used by liveness (a U-Net at 256x256 would otherwise need ~370 MB per patch);
* ``Conv2D(linear) -> BatchNormalization`` is folded into the convolution and a following stand-alone
  activation is fused into the producing convolution (``fuse=True``).
"""
import json

import numpy as np

# op / activation codes mirror include/ecseg_hip.h
OP_CONV, OP_CONVT, OP_MAXPOOL, OP_UPSAMPLE, OP_AFFINE, OP_ACT, OP_ADD, OP_COPY, OP_GLOBALPOOL = 1, 2, 3, 4, 5, 6, 7, 8, 9
ACT = {'linear': 0, None: 0, 'relu': 1, 'softmax': 2, 'sigmoid': 3, 'leaky_relu': 4, 'tanh': 5, 'elu': 6}
IDENTITY_LAYERS = ('Dropout', 'SpatialDropout2D', 'GaussianNoise', 'GaussianDropout', 'AlphaDropout',
                   'ActivityRegularization')


class PlanError(ValueError):
    pass


class Plan:
    """tensors: list of dicts (buffer, h, w, c, c_stride, c_offset); ops: list of dicts (see ecseg_op_desc);
    weights: list of float32 arrays; buffer_floats: per-buffer floats per patch."""

    def __init__(self):
        self.tensors, self.ops, self.weights = [], [], []
        self.n_buffers = 0
        self.buffer_floats = []
        self.input_tensor = self.output_tensor = -1
        self.layer_tensor = {}     # Keras layer name -> tensor index of its output
        self.output_rank = 4       # 2: the Keras model returns (N, K) (classifier heads: Flatten / global pooling / Dense)

    def flops_per_patch(self):
        f = 0.0
        for o in self.ops:
            ti, to = self.tensors[o['in0']], self.tensors[o['out']]
            if o['op'] == OP_CONV:
                f += 2.0 * o['kh'] * o['kw'] * ti['c'] * to['c'] * to['h'] * to['w']
            elif o['op'] == OP_CONVT:
                f += 2.0 * o['kh'] * o['kw'] * ti['c'] * to['c'] * ti['h'] * ti['w']
        return f

    def bytes_per_patch_unfused(self):
        """Algorithmic activation + weight bytes (each conv reads its input and writes its output once)."""
        b = 0.0
        for o in self.ops:
            if o['op'] in (OP_CONV, OP_CONVT):
                ti, to = self.tensors[o['in0']], self.tensors[o['out']]
                b += 4.0 * (ti['h'] * ti['w'] * ti['c'] + to['h'] * to['w'] * to['c']) + 4.0 * self.weights[o['w0']].size
        return b


def _same_pad(k, s, n):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _act_code(name):
    if name not in ACT:
        raise PlanError('unsupported activation %r' % (name,))
    return ACT[name]


def _layers_of(model_config):
    cfg = model_config['config']
    if isinstance(cfg, list):          # very old Sequential format
        return cfg, True, None
    seq = model_config['class_name'] == 'Sequential'
    return cfg['layers'], seq, cfg


def _tfop_affine(L, lc):
    """``TFOpLambda`` nodes TF 2.x records for plain tensor arithmetic in a functional model (``x / 255.``,
    ``x * s``, ``x - m``, ``tf.cast``): -> (scale, offset) of y = x * scale + offset, or None."""
    fn = lc.get('function', '')
    node = L.get('inbound_nodes', [[]])[0]
    kwargs = {}
    if node and isinstance(node[0], (list, tuple)) and len(node[0]) > 3 and isinstance(node[0][3], dict):
        kwargs = node[0][3]
    elif node and len(node) > 3 and isinstance(node[3], dict):       # single-input form [name, 0, 0, {kwargs}]
        kwargs = node[3]
    const = kwargs.get('y', kwargs.get('x'))
    if fn in ('cast', 'identity', 'stop_gradient'):
        return 1.0, 0.0
    if not isinstance(const, (int, float)):
        return None
    if fn in ('math.truediv', 'math.divide', '__operators__.truediv'):
        return 1.0 / float(const), 0.0
    if fn in ('math.multiply', '__operators__.mul'):
        return float(const), 0.0
    if fn in ('math.add', '__operators__.add'):
        return 1.0, float(const)
    if fn in ('math.subtract', '__operators__.sub'):
        return 1.0, -float(const)
    return None


def build_plan(model_config, weights, input_hw=(256, 256), fuse=True, lambda_overrides=None):
    """``model_config``: dict or JSON text; ``weights``: {layer name: [arrays]} -> Plan.
    ``lambda_overrides``: {layer name: (scale, offset)} for ``Lambda`` layers (their Python bytecode cannot be
    interpreted; the common ``Lambda(lambda x: x / 255)`` input normalisation is ``(1/255, 0)``)."""
    lambda_overrides = lambda_overrides or {}
    if isinstance(model_config, (str, bytes)):
        model_config = json.loads(model_config)
    layers, seq, cfg = _layers_of(model_config)

    # ---------------------------------------------------------------- pass 1: logical graph
    nodes = []          # dicts: name, kind, inputs(list of node idx), shape (h, w, c), params
    by_name = {}

    def add(name, kind, inputs, shape, **params):
        if 'rank' not in params:           # rank of the Keras tensor (4: (N, H, W, C); 2: (N, K)), inherited by default
            params['rank'] = nodes[inputs[0]].get('rank', 4) if inputs else 4
        nodes.append(dict(name=name, kind=kind, inputs=list(inputs), shape=tuple(int(v) for v in shape), **params))
        by_name[name] = len(nodes) - 1
        return len(nodes) - 1

    prev = None
    for L in layers:
        cls, lc = L['class_name'], L['config']
        name = lc.get('name', L.get('name'))
        if cls == 'InputLayer' or (seq and prev is None):
            bis = lc.get('batch_input_shape') or lc.get('batch_shape')
            if bis is None:
                raise PlanError('cannot determine the input shape')
            h = bis[1] or input_hw[0]
            w = bis[2] or input_hw[1]
            c = bis[3] if len(bis) > 3 else 1      # (None, H, W): a single-channel image without a channel axis
            if c is None:
                raise PlanError('input channel count is undefined')
            idx = add(name if cls == 'InputLayer' else '__input__', 'input', [], (h, w, c), rank=len(bis))
            prev = idx
            if cls == 'InputLayer':
                continue
        if seq:
            ins = [prev]
        else:
            inb = L.get('inbound_nodes', [])
            if len(inb) != 1:
                raise PlanError('layer %s is shared or unconnected (%d inbound nodes)' % (name, len(inb)))
            node = inb[0]
            if isinstance(node, dict):      # Keras 3 style
                raise PlanError('Keras 3 model_config is not supported (save with TF 2.x / Keras 2)')
            if node and isinstance(node[0], str):      # single-input short form [name, node, tensor, kwargs]
                node = [node]
            ins = []
            for ref in node:
                if ref[0] not in by_name:
                    raise PlanError('layer %s consumes unknown layer %s' % (name, ref[0]))
                ins.append(by_name[ref[0]])
        h, w, c = nodes[ins[0]]['shape']
        ws = weights.get(name, [])
        if cls == 'Conv2D':
            kh, kw = lc['kernel_size']
            sh, sw = lc.get('strides', [1, 1])
            if list(lc.get('dilation_rate', [1, 1])) != [1, 1] or lc.get('groups', 1) != 1 or sh != sw:
                raise PlanError('Conv2D %s: dilation / groups / anisotropic strides are not supported' % name)
            if lc.get('data_format', 'channels_last') != 'channels_last':
                raise PlanError('channels_first is not supported')
            if lc['padding'] == 'same':
                pt, pl = _same_pad(kh, sh, h)[0], _same_pad(kw, sw, w)[0]
                oh, ow = -(-h // sh), -(-w // sw)
            else:
                pt = pl = 0
                oh, ow = (h - kh) // sh + 1, (w - kw) // sw + 1
            kernel = np.ascontiguousarray(ws[0], np.float32)
            if kernel.shape != (kh, kw, c, lc['filters']):
                raise PlanError('Conv2D %s: kernel shape %s does not match config' % (name, kernel.shape))
            bias = np.ascontiguousarray(ws[1], np.float32) if lc.get('use_bias', True) else None
            idx = add(name, 'conv', ins, (oh, ow, lc['filters']), kh=kh, kw=kw, stride=sh, pad_top=pt, pad_left=pl,
                      kernel=kernel, bias=bias, act=_act_code(lc.get('activation')), alpha=0.0)
        elif cls == 'Conv2DTranspose':
            kh, kw = lc['kernel_size']
            sh, sw = lc['strides']
            if sh != sw or list(lc.get('dilation_rate', [1, 1])) != [1, 1] or lc.get('output_padding') not in (None, [None, None]):
                raise PlanError('Conv2DTranspose %s: unsupported geometry' % name)
            kernel = np.ascontiguousarray(ws[0], np.float32)     # (kh, kw, out, in)
            if kernel.shape != (kh, kw, lc['filters'], c):
                raise PlanError('Conv2DTranspose %s: kernel shape %s does not match config' % (name, kernel.shape))
            bias = np.ascontiguousarray(ws[1], np.float32) if lc.get('use_bias', True) else None
            if lc['padding'] == 'same':
                oh, ow = h * sh, w * sw
                ct, cl = max(kh - sh, 0) // 2, max(kw - sw, 0) // 2
            else:
                oh, ow = (h - 1) * sh + max(kh, sh), (w - 1) * sw + max(kw, sw)
                ct = cl = 0
            idx = add(name, 'convt', ins, (oh, ow, lc['filters']), kh=kh, kw=kw, stride=sh, pad_top=ct, pad_left=cl,
                      kernel=kernel, bias=bias, act=_act_code(lc.get('activation')), alpha=0.0)
        elif cls in ('MaxPooling2D', 'AveragePooling2D'):
            kh, kw = lc['pool_size']
            st = lc.get('strides') or lc['pool_size']
            if st[0] != st[1] or lc.get('padding', 'valid') != 'valid':
                raise PlanError('%s %s: unsupported geometry' % (cls, name))
            idx = add(name, 'maxpool', ins, ((h - kh) // st[0] + 1, (w - kw) // st[0] + 1, c), kh=kh, kw=kw, stride=st[0],
                      mode=int(cls == 'AveragePooling2D'))
        elif cls in ('GlobalAveragePooling2D', 'GlobalMaxPooling2D'):
            idx = add(name, 'globalpool', ins, (1, 1, c), mode=int(cls == 'GlobalAveragePooling2D'),
                      rank=4 if lc.get('keepdims') else 2)
        elif cls == 'Flatten':
            idx = add(name, 'reshape', ins, (1, 1, h * w * c), rank=2)
        elif cls == 'Reshape':
            ts = [int(v) for v in lc['target_shape']]
            if len(ts) == 3:
                shp = tuple(ts)
            elif len(ts) == 2:
                shp = (ts[0], ts[1], 1)
            elif len(ts) == 1:
                shp = (1, 1, ts[0])
            else:
                raise PlanError('Reshape %s: target_shape %s is not supported' % (name, ts))
            if -1 in shp or shp[0] * shp[1] * shp[2] != h * w * c:
                raise PlanError('Reshape %s: %s does not match the input (%d, %d, %d)' % (name, ts, h, w, c))
            idx = add(name, 'reshape', ins, shp, rank=len(ts) + 1)
        elif cls == 'Dense':
            kernel = np.ascontiguousarray(ws[0], np.float32)     # (features, units)
            if kernel.shape != (c, lc['units']):
                raise PlanError('Dense %s: kernel shape %s does not match its input (%d features)' % (name, kernel.shape, c))
            bias = np.ascontiguousarray(ws[1], np.float32) if lc.get('use_bias', True) else None
            # a Dense layer acts on the last axis: a 1x1 convolution (on a (1, 1, F) tensor after Flatten / global pooling)
            idx = add(name, 'conv', ins, (h, w, lc['units']), kh=1, kw=1, stride=1, pad_top=0, pad_left=0,
                      kernel=kernel.reshape(1, 1, c, lc['units']), bias=bias, act=_act_code(lc.get('activation')), alpha=0.0)
        elif cls == 'UpSampling2D':
            sz = lc['size']
            if sz[0] != sz[1]:
                raise PlanError('UpSampling2D %s: anisotropic size' % name)
            interp = lc.get('interpolation', 'nearest')
            if interp not in ('nearest', 'bilinear'):
                raise PlanError('UpSampling2D %s: interpolation %s' % (name, interp))
            idx = add(name, 'upsample', ins, (h * sz[0], w * sz[0], c), stride=sz[0], mode=int(interp == 'bilinear'))
        elif cls == 'Concatenate':
            if lc.get('axis', -1) not in (-1, 3):
                raise PlanError('Concatenate %s: only the channel axis is supported' % name)
            for i in ins:
                if nodes[i]['shape'][:2] != (h, w):
                    raise PlanError('Concatenate %s: spatial shapes differ' % name)
            idx = add(name, 'concat', ins, (h, w, sum(nodes[i]['shape'][2] for i in ins)))
        elif cls == 'Add':
            idx = add(name, 'add', ins, (h, w, c))
        elif cls == 'BatchNormalization':
            ax = lc.get('axis', -1)
            ax = ax[0] if isinstance(ax, (list, tuple)) else ax
            if ax not in (-1, 3):
                raise PlanError('BatchNormalization %s: only the channel axis is supported' % name)
            wl = [np.asarray(a, np.float64) for a in ws]
            gamma = wl.pop(0) if lc.get('scale', True) else np.ones(c)
            beta = wl.pop(0) if lc.get('center', True) else np.zeros(c)
            mean, var = wl[0], wl[1]
            inv = gamma / np.sqrt(var + lc.get('epsilon', 1e-3))
            idx = add(name, 'affine', ins, (h, w, c), scale=inv.astype(np.float32),
                      shift=(beta - mean * inv).astype(np.float32), scale64=inv, shift64=beta - mean * inv,
                      act=0, alpha=0.0)
        elif cls == 'Rescaling':
            sc, of = float(lc['scale']), float(lc.get('offset', 0.0))
            idx = add(name, 'affine', ins, (h, w, c), scale=np.full(c, sc, np.float32), shift=np.full(c, of, np.float32),
                      scale64=np.full(c, sc), shift64=np.full(c, of), act=0, alpha=0.0)
        elif cls in ('TFOpLambda', 'Lambda'):
            aff = lambda_overrides.get(name) if cls == 'Lambda' or name in lambda_overrides else _tfop_affine(L, lc)
            if aff is None:
                raise PlanError('%s layer %s cannot be interpreted; pass lambda_overrides={%r: (scale, offset)} if it is '
                                'an affine map such as x / 255' % (cls, name, name))
            sc, of = float(aff[0]), float(aff[1])
            if sc == 1.0 and of == 0.0:
                by_name[name] = ins[0]
                prev = ins[0]
                continue
            idx = add(name, 'affine', ins, (h, w, c), scale=np.full(c, sc, np.float32), shift=np.full(c, of, np.float32),
                      scale64=np.full(c, sc), shift64=np.full(c, of), act=0, alpha=0.0)
        elif cls in IDENTITY_LAYERS:
            by_name[name] = ins[0]
            prev = ins[0]
            continue
        elif cls == 'Activation':
            idx = add(name, 'act', ins, (h, w, c), act=_act_code(lc['activation']), alpha=0.0)
        elif cls == 'ReLU':
            if lc.get('max_value') is not None or lc.get('threshold'):
                raise PlanError('ReLU %s: max_value / threshold are not supported' % name)
            ns = float(lc.get('negative_slope') or 0.0)
            idx = add(name, 'act', ins, (h, w, c), act=ACT['leaky_relu'] if ns else ACT['relu'], alpha=ns)
        elif cls == 'LeakyReLU':
            idx = add(name, 'act', ins, (h, w, c), act=ACT['leaky_relu'], alpha=float(lc.get('alpha', 0.3)))
        elif cls == 'Softmax':
            idx = add(name, 'act', ins, (h, w, c), act=ACT['softmax'], alpha=0.0)
        elif cls == 'ZeroPadding2D':
            (t, b), (l, r) = lc['padding']
            idx = add(name, 'copy', ins, (h + t + b, w + l + r, c), off_y=t, off_x=l)
        elif cls == 'Cropping2D':
            (t, b), (l, r) = lc['cropping']
            idx = add(name, 'copy', ins, (h - t - b, w - l - r, c), off_y=-t, off_x=-l)
        else:
            raise PlanError('Keras layer %s (%s) is not supported' % (cls, name))
        prev = idx

    if seq:
        out_node = prev
    else:
        outs = cfg['output_layers']
        if len(outs) != 1:
            raise PlanError('models with %d outputs are not supported' % len(outs))
        out_node = by_name[outs[0][0]]
    in_nodes = [i for i, n in enumerate(nodes) if n['kind'] == 'input']
    if len(in_nodes) != 1:
        raise PlanError('models with %d inputs are not supported' % len(in_nodes))

    # ---------------------------------------------------------------- pass 2: peephole fusion
    alive = [True] * len(nodes)
    alias = list(range(len(nodes)))     # node -> node that now produces its value

    def consumers(i):
        return [j for j, n in enumerate(nodes) if alive[j] and i in n['inputs']]

    if fuse:
        for j, n in enumerate(nodes):
            if not alive[j] or len(n['inputs']) != 1:
                continue
            i = n['inputs'][0]
            p = nodes[i]
            if p['kind'] not in ('conv', 'convt') or len(consumers(i)) != 1 or i == out_node:
                continue
            if n['kind'] == 'affine' and p['act'] == 0 and n['act'] == 0:
                s, t = n['scale64'], n['shift64']
                k = p['kernel'].astype(np.float64)
                p['kernel'] = (k * (s[None, None, None, :] if p['kind'] == 'conv' else s[None, None, :, None])).astype(np.float32)
                b = p['bias'].astype(np.float64) if p['bias'] is not None else np.zeros(len(s))
                p['bias'] = (b * s + t).astype(np.float32)
            elif n['kind'] == 'act' and p['act'] == 0:
                p['act'], p['alpha'] = n['act'], n['alpha']
            else:
                continue
            # node j disappears: its consumers read the conv directly; keep the Keras name of j on the conv output
            alive[j] = False
            alias[j] = i
            for m in nodes:
                m['inputs'] = [i if x == j else x for x in m['inputs']]
            if out_node == j:
                out_node = i
            p.setdefault('also', []).append(n['name'])

    # ---------------------------------------------------------------- pass 3: views for Concatenate
    # view[i] = (root concat node, channel offset) when node i's output lives inside a concat buffer
    view = {}
    copies = {}       # (concat node, position) -> explicit copy needed
    for j, n in enumerate(nodes):
        if not alive[j] or n['kind'] != 'concat':
            continue
        off = 0
        for pos, i in enumerate(n['inputs']):
            ci = nodes[i]['shape'][2]
            ok = (i not in view and nodes[i]['kind'] not in ('input', 'concat') and n['inputs'].count(i) == 1
                  and i != out_node)
            if ok:
                view[i] = (j, off)
            else:
                copies[(j, pos)] = off
            off += ci

    # ---------------------------------------------------------------- pass 4: emit tensors / ops with buffer reuse
    plan = Plan()
    order = [j for j in range(len(nodes)) if alive[j]]
    last_use = {}
    for j in order:
        for i in nodes[j]['inputs']:
            last_use[i] = j
    last_use[out_node] = len(nodes) + 1
    # Flatten / Reshape are views of their input's buffer: the owner lives as long as the view is read
    def owner(i):
        while nodes[i]['kind'] == 'reshape':
            i = nodes[i]['inputs'][0]
        return i
    for j in reversed(order):
        if nodes[j]['kind'] == 'reshape':
            o_ = nodes[j]['inputs'][0]
            last_use[o_] = max(last_use.get(o_, o_), last_use.get(j, j))
    # a concat buffer lives from its first producer to the concat's last use
    first_touch = {}
    for i, (j, _) in view.items():
        first_touch[j] = min(first_touch.get(j, i), i)
        last_use[i] = max(last_use.get(i, i), last_use.get(j, j))

    # a 2x2 max-pool that directly follows its producing convolution may be written by that convolution's output stage
    # (csrc/api.hip run_plan): the convolution's input must then still be alive when the pool's buffer is chosen
    # ... and likewise a 1x1 convolution (the head) that directly follows a convolution may be finished by that
    # convolution's output stage: its output buffer must differ from the producing convolution's INPUT buffer, which other
    # workgroups are still reading
    for a, b in zip(order, order[1:]):
        fused_pool = nodes[b]['kind'] == 'maxpool'
        fused_head = nodes[b]['kind'] == 'conv' and nodes[b]['kh'] == 1 and nodes[b]['kw'] == 1
        if (fused_pool or fused_head) and nodes[a]['kind'] == 'conv' and nodes[b]['inputs'] == [a]:
            for i in nodes[a]['inputs']:
                last_use[i] = max(last_use.get(i, i), b)

    free = []          # (floats, buffer id)
    node_buf = {}      # node -> buffer id (for nodes that own a buffer)
    tensor_of = {}

    def alloc(floats, fresh=False):
        best = None
        for k, (sz, b) in enumerate([] if fresh else free):
            if sz >= floats and (best is None or sz < free[best][0]):
                best = k
        if best is not None:
            sz, b = free.pop(best)
            return b
        plan.buffer_floats.append(floats)
        plan.n_buffers += 1
        return plan.n_buffers - 1

    def ensure_buffer(j):
        """Buffer that node j's value is written to (allocating the concat buffer on first touch)."""
        if j in view:
            root, off = view[j]
            if root not in node_buf:
                h, w, c = nodes[root]['shape']
                node_buf[root] = alloc(h * w * c)
            h, w, c = nodes[j]['shape']
            return node_buf[root], off, nodes[root]['shape'][2]
        if j not in node_buf:
            h, w, c = nodes[j]['shape']
            # the model output gets a buffer of its own (and the input's is never re-used, below): the window lanes of
            # csrc/api.hip run_plan address both in plain window order, whatever else a lane packs into shared buffers
            node_buf[j] = alloc(h * w * c, fresh=(j == out_node))
        return node_buf[j], 0, nodes[j]['shape'][2]

    def new_tensor(j):
        b, off, cs = ensure_buffer(j)
        h, w, c = nodes[j]['shape']
        plan.tensors.append(dict(buffer=b, h=h, w=w, c=c, c_stride=cs, c_offset=off))
        tensor_of[j] = len(plan.tensors) - 1
        plan.layer_tensor[nodes[j]['name']] = tensor_of[j]
        for nm in nodes[j].get('also', []):
            plan.layer_tensor[nm] = tensor_of[j]
        return tensor_of[j]

    def add_weight(arr):
        if arr is None:
            return -1
        plan.weights.append(np.ascontiguousarray(arr, np.float32).ravel())
        return len(plan.weights) - 1

    def op(**kw):
        d = dict(op=0, in0=-1, in1=-1, out=-1, kh=0, kw=0, stride=1, pad_top=0, pad_left=0, act=0, mode=0, w0=-1, w1=-1, alpha=0.0)
        d.update(kw)
        plan.ops.append(d)

    released = set()

    def release_after(step):
        for i in list(node_buf.keys()):
            if i in released:
                continue
            lu = last_use.get(i, i)
            if nodes[i]['kind'] == 'concat':
                lu = max([lu] + [last_use.get(x, x) for x in nodes[i]['inputs']])
            if lu <= step and nodes[i]['kind'] != 'input':
                released.add(i)
                free.append((plan.buffer_floats[node_buf[i]], node_buf[i]))

    for j in order:
        n = nodes[j]
        k = n['kind']
        if k == 'input':
            plan.input_tensor = new_tensor(j)
        elif k == 'reshape':
            i = n['inputs'][0]
            ti_ = plan.tensors[tensor_of[i]]
            if i in view or nodes[i]['kind'] == 'concat' or ti_['c_stride'] != ti_['c'] or ti_['c_offset'] != 0:
                raise PlanError('%s: reshaping a strided view is not supported' % n['name'])
            h_, w_, c_ = n['shape']
            plan.tensors.append(dict(buffer=ti_['buffer'], h=h_, w=w_, c=c_, c_stride=c_, c_offset=0))
            tensor_of[j] = len(plan.tensors) - 1
            plan.layer_tensor[n['name']] = tensor_of[j]
        elif k == 'concat':
            t = new_tensor(j)
            off = 0
            for pos, i in enumerate(n['inputs']):
                ci = nodes[i]['shape'][2]
                if (j, pos) in copies:
                    h, w, c = n['shape']
                    plan.tensors.append(dict(buffer=node_buf[j], h=h, w=w, c=ci, c_stride=c, c_offset=off))
                    op(op=OP_COPY, in0=tensor_of[i], out=len(plan.tensors) - 1)
                off += ci
        else:
            ins = [tensor_of[i] for i in n['inputs']]
            t = new_tensor(j)
            if k == 'conv':
                op(op=OP_CONV, in0=ins[0], out=t, kh=n['kh'], kw=n['kw'], stride=n.get('stride', 1), pad_top=n['pad_top'], pad_left=n['pad_left'],
                   act=n['act'], alpha=n['alpha'], w0=add_weight(n['kernel']), w1=add_weight(n['bias']))
            elif k == 'convt':
                op(op=OP_CONVT, in0=ins[0], out=t, kh=n['kh'], kw=n['kw'], stride=n['stride'], pad_top=n['pad_top'],
                   pad_left=n['pad_left'], act=n['act'], alpha=n['alpha'], w0=add_weight(n['kernel']), w1=add_weight(n['bias']))
            elif k == 'maxpool':
                op(op=OP_MAXPOOL, in0=ins[0], out=t, kh=n['kh'], kw=n['kw'], stride=n['stride'], mode=n.get('mode', 0))
            elif k == 'globalpool':
                op(op=OP_GLOBALPOOL, in0=ins[0], out=t, mode=n['mode'])
            elif k == 'upsample':
                op(op=OP_UPSAMPLE, in0=ins[0], out=t, stride=n['stride'], mode=n['mode'])
            elif k == 'affine':
                op(op=OP_AFFINE, in0=ins[0], out=t, act=n['act'], alpha=n['alpha'], w0=add_weight(n['scale']), w1=add_weight(n['shift']))
            elif k == 'act':
                op(op=OP_ACT, in0=ins[0], out=t, act=n['act'], alpha=n['alpha'])
            elif k == 'add':
                if len(ins) < 2:
                    raise PlanError('Add %s needs two inputs' % n['name'])
                op(op=OP_ADD, in0=ins[0], in1=ins[1], out=t)
                for extra in ins[2:]:
                    op(op=OP_ADD, in0=t, in1=extra, out=t)
            elif k == 'copy':
                op(op=OP_COPY, in0=ins[0], out=t, pad_top=n['off_y'], pad_left=n['off_x'])
            else:
                raise PlanError('internal: unknown node kind %s' % k)
        release_after(j)
    plan.output_tensor = tensor_of[out_node]
    plan.output_rank = 2 if nodes[out_node].get('rank', 4) == 2 else 4
    # the model output must be readable as a compact tensor for predict_on_batch; views are fine for the C side
    return plan

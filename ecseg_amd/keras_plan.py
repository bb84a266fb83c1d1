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
OP_DWCONV, OP_PRELU, OP_LAYERNORM = 10, 11, 12
ACT = {'linear': 0, None: 0, 'relu': 1, 'softmax': 2, 'sigmoid': 3, 'leaky_relu': 4, 'tanh': 5, 'elu': 6, 'relu_clip': 7, 'relu6': 7,
       'swish': 8, 'silu': 8, 'hard_sigmoid': 9, 'softplus': 10, 'selu': 11, 'gelu': 12, 'exponential': 13, 'softsign': 14}
ACT_ALPHA = {'elu': 1.0, 'relu6': 6.0}       # parameter of an activation given by name
BIN = {'Add': 0, 'Multiply': 1, 'Subtract': 2, 'Maximum': 3, 'Minimum': 4, 'Average': 0}
MODEL_CLASSES = ('Functional', 'Model', 'Sequential')
CONV_ACT_MAX = 7      # activation codes the convolution kernels' output stages implement (device_util.h: apply_act); the others run
                      # as an element-wise pass over the convolution's output
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
        self.channels_first = False   # the Keras model takes / returns (N, C, H, W): the caller transposes at the boundary (channels_first_to_last)

    def flops_per_patch(self):
        f = 0.0
        for o in self.ops:
            ti, to = self.tensors[o['in0']], self.tensors[o['out']]
            if o['op'] == OP_CONV:
                f += 2.0 * o['kh'] * o['kw'] * ti['c'] * to['c'] * to['h'] * to['w']
            elif o['op'] == OP_CONVT:
                f += 2.0 * o['kh'] * o['kw'] * ti['c'] * to['c'] * ti['h'] * ti['w']
            elif o['op'] == OP_DWCONV:
                f += 2.0 * o['kh'] * o['kw'] * to['c'] * to['h'] * to['w']
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
    if isinstance(name, dict):             # a serialised activation object: {'class_name': ..., 'config': ...}
        name = (name.get('config') or {}).get('name', name.get('class_name'))
    if name not in ACT:
        raise PlanError('unsupported activation %r' % (name,))
    return ACT[name]


def _act_pair(name):
    """-> dict(act=code, alpha=parameter) of an activation given by name in a layer config."""
    if isinstance(name, dict):
        name = (name.get('config') or {}).get('name', name.get('class_name'))
    return dict(act=_act_code(name), alpha=ACT_ALPHA.get(name, 0.0))


class NamedWeights(list):
    """Weight arrays of one top-level Keras layer with their HDF5 ``weight_names``: a nested model's list mixes the variables
    of all its layers (trainable ones first), the names say which inner layer each belongs to."""

    def __init__(self, arrays, names=None):
        super().__init__(arrays)
        self.names = list(names) if names is not None else None


def _refs(L):
    """The single inbound node of a functional layer as a list of [layer name, node, tensor, kwargs] references."""
    inb = L.get('inbound_nodes', [])
    if not inb:
        return []
    if len(inb) != 1:
        raise PlanError('layer %s is shared (%d inbound nodes)' % (L.get('name', L['config'].get('name')), len(inb)))
    node = inb[0]
    if isinstance(node, dict):
        raise PlanError('Keras 3 model_config is not supported (save with TF 2.x / Keras 2)')
    if node and isinstance(node[0], str):      # single-input short form [name, node, tensor, kwargs]
        node = [node]
    return [list(r) for r in node]


def _to_functional(mc):
    """A Sequential model as the equivalent Functional config (a chain), so that the rest of the lowering sees one form."""
    layers, seq, cfg = _layers_of(mc)
    if not seq:
        return mc
    out, prev = [], None
    for L in layers:
        lc = L['config']
        name = lc.get('name', L.get('name'))
        if L['class_name'] == 'InputLayer':
            out.append(dict(L, inbound_nodes=[]))
            prev = name
            continue
        if prev is None:
            bis = lc.get('batch_input_shape') or lc.get('batch_shape')
            if bis is None and L['class_name'] in MODEL_CLASSES:        # a Sequential that starts with a nested model
                ilayers = _layers_of({'class_name': L['class_name'], 'config': lc})[0]
                bis = ilayers[0]['config'].get('batch_input_shape') or ilayers[0]['config'].get('batch_shape')
            if bis is None:
                raise PlanError('cannot determine the input shape')
            out.append({'class_name': 'InputLayer', 'name': '__input__', 'inbound_nodes': [],
                        'config': {'name': '__input__', 'batch_input_shape': bis}})
            prev = '__input__'
        out.append(dict(L, inbound_nodes=[[[prev, 0, 0, {}]]]))
        prev = name
    first = out[0]['config']['name'] if out else None
    return {'class_name': 'Functional', 'config': {'name': (cfg or {}).get('name', 'sequential') if isinstance(cfg, dict) else 'sequential',
                                                   'layers': out, 'input_layers': [[first, 0, 0]], 'output_layers': [[prev, 0, 0]]}}


def _group_nested_weights(ws, owner):
    """Weights of a nested model -> {inner layer name: [arrays in that layer's own order]}."""
    if ws is None:
        return {}
    if isinstance(ws, dict):
        return ws
    names = getattr(ws, 'names', None)
    if names is None or len(names) != len(ws):
        if not len(ws):
            return {}
        raise PlanError('nested model %s: its %d weight arrays carry no names - pass {inner layer name: [arrays]} or a '
                        'NamedWeights list (hdf5_min.load_keras_h5 returns those)' % (owner, len(ws)))
    d = {}
    for nm, a in zip(names, ws):
        parts = str(nm).split(':')[0].split('/')
        d.setdefault(parts[-2] if len(parts) >= 2 else parts[0], []).append(a)
    return d


SPATIAL_CLASSES = ('Conv2D', 'Conv2DTranspose', 'DepthwiseConv2D', 'SeparableConv2D', 'MaxPooling2D', 'AveragePooling2D', 'UpSampling2D',
                   'ZeroPadding2D', 'Cropping2D', 'GlobalAveragePooling2D', 'GlobalMaxPooling2D')


def channels_first_to_last(mc, weights):
    """A Functional config whose layers say ``data_format: channels_first`` (tensors (N, C, H, W)) -> (the equivalent channels_last config,
    weights, True); a channels_last config comes back unchanged with False.  Keras stores convolution kernels as (kh, kw, in, out) in
    BOTH formats, so only shapes and axes move: the input shape (C, H, W) -> (H, W, C), channel axis 1 -> 3 (BatchNormalization,
    Normalization, LayerNormalization, Concatenate, Softmax), PReLU slopes (C, H, W) -> (H, W, C) with shared axes 2, 3 -> 1, 2.
    The caller transposes the arrays at the boundary (MetasegModel).  What has no channels_last twin is refused: Flatten in
    channels_last memory order, Reshape / Permute, Dense or a last-axis softmax on a 4-D tensor (they act on W there)."""
    layers = mc['config']['layers']
    fmts = {L['config'].get('data_format') for L in layers if L['class_name'] in SPATIAL_CLASSES}
    if 'channels_first' not in fmts:
        return mc, weights, False
    if fmts - {'channels_first'}:
        raise PlanError('the model mixes channels_first and channels_last layers')
    ax_map = {1: 3, -3: 3, 2: 1, 3: 2, -2: 1, -1: 2}
    wout = dict(weights)
    out = []
    for L in layers:
        cls, lc = L['class_name'], dict(L['config'])
        name = lc.get('name', L.get('name'))
        if cls == 'InputLayer':
            key = 'batch_input_shape' if 'batch_input_shape' in lc else 'batch_shape'
            b = lc.get(key)
            if b is not None and len(b) == 4:
                lc[key] = [b[0], b[2], b[3], b[1]]
        elif cls in SPATIAL_CLASSES:
            lc['data_format'] = 'channels_last'
            if lc.get('batch_input_shape') is not None and len(lc['batch_input_shape']) == 4:
                b = lc['batch_input_shape']
                lc['batch_input_shape'] = [b[0], b[2], b[3], b[1]]
        elif cls in ('BatchNormalization', 'Normalization', 'LayerNormalization', 'Concatenate', 'Softmax'):
            ax = lc.get('axis', -1)
            one = not isinstance(ax, (list, tuple))
            axs = [ax] if one else list(ax)
            if axs != [None]:
                bad = [a for a in axs if a not in ax_map]
                if bad:
                    raise PlanError('%s %s: axis %s of a channels_first tensor' % (cls, name, ax))
                axs = [ax_map[a] for a in axs]
                if cls in ('Concatenate', 'Softmax') and axs != [3]:
                    raise PlanError('%s %s acts on a spatial axis of a channels_first tensor (axis %s)' % (cls, name, ax))
                lc['axis'] = axs[0] if one else axs
        elif cls == 'PReLU':
            sh = lc.get('shared_axes')
            if sh is not None:
                lc['shared_axes'] = sorted(ax_map[a] for a in sh)
            if name in weights and np.asarray(weights[name][0]).ndim == 3:
                wout[name] = [np.ascontiguousarray(np.transpose(np.asarray(weights[name][0]), (1, 2, 0)))] + list(weights[name][1:])
        elif cls == 'Flatten':
            if lc.get('data_format') != 'channels_first':
                raise PlanError('Flatten %s flattens a channels_first tensor in (C, H, W) order: not supported (Flatten(data_format='
                                "'channels_first') - the order Keras itself converts to - is)" % name)
            lc['data_format'] = 'channels_last'
        elif cls in ('Reshape', 'Permute', 'Dense') or (cls == 'Activation' and lc.get('activation') == 'softmax'):
            raise PlanError('%s %s in a channels_first model acts on the W axis of (N, C, H, W): not supported' % (cls, name))
        out.append(dict(L, config=lc))
    return dict(mc, config=dict(mc['config'], layers=out)), wout, True


def unshare_layers(mc, weights):
    """A Functional config in which some layer is CALLED more than once (several inbound nodes: shared weights - a Siamese branch, one
    block applied at two scales) -> (config, weights) in which every call is a layer of its own: call j >= 1 of layer ``name`` becomes
    ``name/call<j>`` with the same weight arrays, every reference [layer, node index, tensor index] is rewritten to the clone it means,
    and the layers are put into dependency order (Keras lists a layer once, where it was created - the inputs of its later calls are
    defined further down).  The reference loads such files like any other (/root/reference/src/utils.py:27-33)."""
    cfg = mc['config']
    layers = cfg['layers']
    if all(len(L.get('inbound_nodes', [])) <= 1 for L in layers):
        return mc, weights

    def lname(L):
        return L['config'].get('name', L.get('name'))

    def norm(node):
        if isinstance(node, dict):
            raise PlanError('Keras 3 model_config is not supported (save with TF 2.x / Keras 2)')
        if node and isinstance(node[0], str):
            node = [node]
        return [list(r) for r in node]

    calls = {lname(L): len(L.get('inbound_nodes', [])) for L in layers}

    def clone_name(name, j):
        if j and j >= max(calls.get(name, 1), 1):
            raise PlanError('reference to call %d of layer %s, which is called %d time(s)' % (j, name, calls.get(name, 0)))
        return name if not j else '%s/call%d' % (name, j)

    def fix(ref):
        j = ref[1] if len(ref) > 1 and isinstance(ref[1], int) else 0
        return [clone_name(ref[0], j), 0] + list(ref[2:])

    wout = dict(weights)
    out = []
    for L in layers:
        inb = L.get('inbound_nodes', [])
        name = lname(L)
        if not inb:
            out.append(L)
            continue
        for j, node in enumerate(inb):
            nm = clone_name(name, j)
            out.append(dict(L, name=nm, config=dict(L['config'], name=nm), inbound_nodes=[[fix(r) for r in norm(node)]]))
            if j and name in weights:
                wout[nm] = weights[name]
    # dependency order (stable): a clone may consume layers that are listed after the layer it was cloned from
    pos = {lname(L): k for k, L in enumerate(out)}
    done, order = set(), []

    def visit(k, stack=()):
        nm = lname(out[k])
        if nm in done:
            return
        if nm in stack:
            raise PlanError('the layer graph has a cycle through %s' % nm)
        for node in out[k].get('inbound_nodes', []):
            for r in node:
                if r[0] not in pos:
                    raise PlanError('layer %s consumes unknown layer %s' % (nm, r[0]))
                visit(pos[r[0]], stack + (nm,))
        done.add(nm)
        order.append(out[k])

    for k in range(len(out)):
        visit(k)
    new_cfg = dict(cfg, layers=order, input_layers=[fix(list(r)) for r in cfg.get('input_layers', [])],
                   output_layers=[fix(list(r)) for r in cfg.get('output_layers', [])])
    return dict(mc, config=new_cfg), wout


def inline_nested(model_config, weights):
    """-> (flat Functional model_config, weights): every nested Functional / Sequential sub-model (a transfer-learning
    backbone used as one layer, ``tf.keras.applications.*`` inside a classifier) is replaced by its own layers, named
    ``<sub-model>/<layer>``; the sub-model's InputLayers become references to the tensors it is called on."""
    mc = _to_functional(model_config)
    mc, weights = unshare_layers(mc, weights)
    cfg = mc['config']
    new_layers, alias, wout = [], {}, dict(weights)

    def resolve(ref):
        if ref[0] in alias:
            t = ref[2] if len(ref) > 2 and isinstance(ref[2], int) else 0
            outs = alias[ref[0]]
            if t >= len(outs):
                raise PlanError('reference to output %d of %s, which has %d' % (t, ref[0], len(outs)))
            return [outs[t], 0, 0] + list(ref[3:])
        return ref

    for L in cfg['layers']:
        cls, lc = L['class_name'], L['config']
        name = lc.get('name', L.get('name'))
        refs = [resolve(r) for r in _refs(L)]
        if cls in MODEL_CLASSES:
            icfg, iw = inline_nested({'class_name': cls, 'config': lc}, _group_nested_weights(weights.get(name), name))
            in_names = [r[0] for r in icfg['config']['input_layers']]
            if len(in_names) != len(refs):
                raise PlanError('nested model %s has %d inputs but is called on %d tensors' % (name, len(in_names), len(refs)))
            feed = {n: refs[k][0] for k, n in enumerate(in_names)}
            prefix = name + '/'
            for IL in icfg['config']['layers']:
                iname = IL['config'].get('name', IL.get('name'))
                if IL['class_name'] == 'InputLayer':
                    continue
                nrefs = [[feed.get(r[0], prefix + r[0])] + list(r[1:]) for r in _refs(IL)]
                new_layers.append(dict(IL, name=prefix + iname, config=dict(IL['config'], name=prefix + iname), inbound_nodes=[nrefs]))
                if iname in iw:
                    wout[prefix + iname] = iw[iname]
            alias[name] = [feed.get(r[0], prefix + r[0]) for r in icfg['config']['output_layers']]
            wout.pop(name, None)
            continue
        new_layers.append(dict(L, inbound_nodes=[refs] if refs else []))
    outs = [resolve(list(r)) for r in cfg['output_layers']]
    flat = {'class_name': 'Functional', 'config': dict(cfg, layers=new_layers, output_layers=outs)}
    return flat, wout


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


def _conv_geometry(cls, name, lc, h, w, aniso=None):
    """Conv2D / DepthwiseConv2D / SeparableConv2D config -> (kh, kw, stride, dilation, pad_top, pad_left, out_h, out_w).
    'same' pads max((ceil(n / s) - 1) s + (k - 1) d + 1 - n, 0) in total, the smaller half in front (TensorFlow's rule with the
    dilated extent of the kernel).  ``aniso``: a list (plain Conv2D only) that receives the CONV op's ``mode`` word when the
    horizontal stride / dilation rate differs from the vertical one returned here - horizontal stride in bits 0-7, horizontal
    dilation rate in bits 8-15, 0 where equal (include/ecseg_hip.h); without it such layers are rejected."""
    kh, kw = lc['kernel_size']
    sh, sw = lc.get('strides', [1, 1])
    dr = lc.get('dilation_rate', [1, 1])
    dr = [dr, dr] if isinstance(dr, int) else list(dr)
    dil, dilw = int(dr[0]), int(dr[1])
    if sh != sw or dil != dilw:
        if aniso is None:
            raise PlanError('%s %s: anisotropic strides / dilation rates are not supported' % (cls, name))
        if not (0 < sw < 256 and 0 < dilw < 256):
            raise PlanError('%s %s: horizontal stride / dilation rate out of range' % (cls, name))
        aniso.append((sw if sw != sh else 0) | ((dilw if dilw != dil else 0) << 8))
    if lc.get('data_format', 'channels_last') != 'channels_last':
        raise PlanError('channels_first is not supported')
    if max(dil, dilw) > 1 and max(sh, sw) > 1:
        raise PlanError('%s %s: strides > 1 together with dilation_rate > 1 (Keras rejects it too)' % (cls, name))
    ekh, ekw = (kh - 1) * dil + 1, (kw - 1) * dilw + 1
    if lc['padding'] == 'same':
        pt, pl = _same_pad(ekh, sh, h)[0], _same_pad(ekw, sw, w)[0]
        oh, ow = -(-h // sh), -(-w // sw)
    elif lc['padding'] == 'valid':
        pt = pl = 0
        oh, ow = (h - ekh) // sh + 1, (w - ekw) // sw + 1
    else:
        raise PlanError('%s %s: padding %r is not supported' % (cls, name, lc['padding']))
    if oh <= 0 or ow <= 0:
        raise PlanError('%s %s: the kernel does not fit the %d x %d input' % (cls, name, h, w))
    return kh, kw, sh, dil, pt, pl, oh, ow


def build_plan(model_config, weights, input_hw=(256, 256), fuse=True, lambda_overrides=None, output=0):
    """``model_config``: dict or JSON text; ``weights``: {layer name: [arrays]} -> Plan.
    ``lambda_overrides``: {layer name: (scale, offset)} for ``Lambda`` layers (their Python bytecode cannot be
    interpreted; the common ``Lambda(lambda x: x / 255)`` input normalisation is ``(1/255, 0)``).
    ``output``: which output of a model with several (index into ``output_layers``, or a layer name): the plan computes that
    one (``predict_on_batch`` of such a Keras model returns a list; the callers of the reference read a single array,
    src/utils.py:115, src/interseg.py:155,168)."""
    lambda_overrides = lambda_overrides or {}
    if isinstance(model_config, (str, bytes)):
        model_config = json.loads(model_config)
    model_config, weights = inline_nested(model_config, weights)
    model_config, weights, was_channels_first = channels_first_to_last(model_config, weights)
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
            groups = int(lc.get('groups', 1) or 1)
            aniso = [] if groups == 1 else None        # per-axis strides / dilation rates: plain convolutions only (scalar kernel)
            kh, kw, sh, dil, pt, pl, oh, ow = _conv_geometry(cls, name, lc, h, w, aniso)
            filters = lc['filters']
            if groups < 1 or c % groups or filters % groups:
                raise PlanError('Conv2D %s: %d groups do not divide %d -> %d channels' % (name, groups, c, filters))
            kernel = np.ascontiguousarray(ws[0], np.float32)
            if kernel.shape != (kh, kw, c // groups, filters):
                raise PlanError('Conv2D %s: kernel shape %s does not match config' % (name, kernel.shape))
            bias = np.ascontiguousarray(ws[1], np.float32) if lc.get('use_bias', True) else None
            geo = dict(kh=kh, kw=kw, stride=sh, dilation=dil, pad_top=pt, pad_left=pl)
            if groups == 1:
                idx = add(name, 'conv', ins, (oh, ow, filters), kernel=kernel, bias=bias, aniso=aniso[0] if aniso else 0, **geo,
                          **_act_pair(lc.get('activation')))
            elif groups == c:
                # one input channel per group: a depthwise convolution with depth multiplier filters / c (output channel
                # o belongs to group o // (filters / groups): Keras' depthwise channel order)
                idx = add(name, 'dwconv', ins, (oh, ow, filters), kernel=kernel.reshape(kh, kw, c, filters // c), bias=bias,
                          mult=filters // c, **geo, **_act_pair(lc.get('activation')))
            else:
                # grouped convolution: one convolution per group on a channel slice of the input, written straight into its
                # slice of the output (the Concatenate machinery below)
                cg, fg = c // groups, filters // groups
                parts = []
                for g in range(groups):
                    si = add('%s/in%d' % (name, g), 'slice', ins, (h, w, cg), c0=g * cg)
                    parts.append(add('%s/group%d' % (name, g), 'conv', [si], (oh, ow, fg),
                                     kernel=np.ascontiguousarray(kernel[..., g * fg:(g + 1) * fg]),
                                     bias=None if bias is None else np.ascontiguousarray(bias[g * fg:(g + 1) * fg]), **geo,
                                     **_act_pair(lc.get('activation'))))
                idx = add(name, 'concat', parts, (oh, ow, filters))
        elif cls == 'DepthwiseConv2D':
            kh, kw, sh, dil, pt, pl, oh, ow = _conv_geometry(cls, name, lc, h, w)
            mult = int(lc.get('depth_multiplier', 1))
            kernel = np.ascontiguousarray(ws[0], np.float32)
            if kernel.shape != (kh, kw, c, mult):
                raise PlanError('DepthwiseConv2D %s: kernel shape %s does not match config' % (name, kernel.shape))
            bias = np.ascontiguousarray(ws[1], np.float32) if lc.get('use_bias', True) else None
            idx = add(name, 'dwconv', ins, (oh, ow, c * mult), kernel=kernel, bias=bias, mult=mult, kh=kh, kw=kw, stride=sh,
                      dilation=dil, pad_top=pt, pad_left=pl, **_act_pair(lc.get('activation')))
        elif cls == 'SeparableConv2D':
            # depthwise (no bias, no activation) then pointwise 1x1 (bias, activation): weights [depthwise_kernel,
            # pointwise_kernel, bias]
            kh, kw, sh, dil, pt, pl, oh, ow = _conv_geometry(cls, name, lc, h, w)
            mult = int(lc.get('depth_multiplier', 1))
            dk = np.ascontiguousarray(ws[0], np.float32)
            pk = np.ascontiguousarray(ws[1], np.float32)
            if dk.shape != (kh, kw, c, mult) or pk.shape != (1, 1, c * mult, lc['filters']):
                raise PlanError('SeparableConv2D %s: kernel shapes %s / %s do not match config' % (name, dk.shape, pk.shape))
            bias = np.ascontiguousarray(ws[2], np.float32) if lc.get('use_bias', True) else None
            di = add(name + '/depthwise', 'dwconv', ins, (oh, ow, c * mult), kernel=dk, bias=None, mult=mult, kh=kh, kw=kw, stride=sh,
                     dilation=dil, pad_top=pt, pad_left=pl, act=0, alpha=0.0)
            idx = add(name, 'conv', [di], (oh, ow, lc['filters']), kh=1, kw=1, stride=1, dilation=1, pad_top=0, pad_left=0, kernel=pk,
                      bias=bias, **_act_pair(lc.get('activation')))
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
                      kernel=kernel, bias=bias, **_act_pair(lc.get('activation')))
        elif cls in ('MaxPooling2D', 'AveragePooling2D'):
            kh, kw = lc['pool_size']
            st = lc.get('strides') or lc['pool_size']
            if st[0] != st[1]:
                raise PlanError('%s %s: unsupported geometry' % (cls, name))
            if lc.get('padding', 'valid') == 'same':
                pt, pl = _same_pad(kh, st[0], h)[0], _same_pad(kw, st[0], w)[0]
                oh, ow = -(-h // st[0]), -(-w // st[0])
            else:
                pt = pl = 0
                oh, ow = (h - kh) // st[0] + 1, (w - kw) // st[0] + 1
            idx = add(name, 'maxpool', ins, (oh, ow, c), kh=kh, kw=kw, stride=st[0], pad_top=pt, pad_left=pl,
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
                      kernel=kernel.reshape(1, 1, c, lc['units']), bias=bias, **_act_pair(lc.get('activation')))
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
        elif cls in BIN:
            if len(ins) < 2:
                raise PlanError('%s %s needs at least two inputs' % (cls, name))
            shp = [h, w, c]
            for i in ins[1:]:
                for a in range(3):
                    v = nodes[i]['shape'][a]
                    if v != shp[a] and v != 1 and shp[a] != 1:
                        raise PlanError('%s %s: shapes %s and %s cannot be broadcast' % (cls, name, tuple(shp), nodes[i]['shape']))
                    shp[a] = max(shp[a], v)
            if cls == 'Subtract' and len(ins) != 2:
                raise PlanError('Subtract %s takes exactly two inputs' % name)
            if cls == 'Average':                                # mean of n tensors: the sum, then one per-channel scale
                si = add(name + '/sum', 'add', ins, tuple(shp), mode=0, rank=max(nodes[i].get('rank', 4) for i in ins))
                sc = 1.0 / len(ins)
                idx = add(name, 'affine', [si], tuple(shp), scale=np.full(shp[2], sc, np.float32), shift=np.zeros(shp[2], np.float32),
                          scale64=np.full(shp[2], sc), shift64=np.zeros(shp[2]), act=0, alpha=0.0)
            else:
                idx = add(name, 'add', ins, tuple(shp), mode=BIN[cls], rank=max(nodes[i].get('rank', 4) for i in ins))
        elif cls == 'PReLU':
            alpha = np.asarray(ws[0], np.float32)
            shared = lc.get('shared_axes') or []
            full = [1 if (a + 1) in shared else v for a, v in enumerate((h, w, c))]
            if nodes[ins[0]].get('rank', 4) == 2:               # (N, K) input: alpha has shape (K,)
                alpha = alpha.reshape(1, 1, -1)
            if tuple(alpha.shape) != tuple(full):
                raise PlanError('PReLU %s: alpha shape %s does not match input %s with shared_axes %s' % (name, alpha.shape, (h, w, c), shared))
            per_channel = full[0] == 1 and full[1] == 1
            if per_channel:
                slopes = np.ascontiguousarray(np.broadcast_to(alpha, (1, 1, c)).reshape(c))
            else:
                slopes = np.ascontiguousarray(np.broadcast_to(alpha, (h, w, c)))
            idx = add(name, 'prelu', ins, (h, w, c), slopes=slopes, mode=int(not per_channel))
        elif cls == 'LayerNormalization':
            ax = lc.get('axis', -1)
            ax = list(ax) if isinstance(ax, (list, tuple)) else [ax]
            rank_in = nodes[ins[0]].get('rank', 4)
            if ax not in ([-1], [rank_in - 1]):
                raise PlanError('LayerNormalization %s: only the channel (last) axis is supported, got axis %s' % (name, ax))
            wl = [np.asarray(a, np.float32).reshape(-1) for a in ws]
            gamma = wl.pop(0) if lc.get('scale', True) else None
            beta = wl.pop(0) if lc.get('center', True) else None
            idx = add(name, 'layernorm', ins, (h, w, c), gamma=gamma, beta=beta, eps=float(lc.get('epsilon', 1e-3)))
        elif cls == 'Normalization':
            # preprocessing layer: (x - mean) / max(sqrt(variance), epsilon); statistics as weights [mean, variance, count] (adapt())
            # or in the config
            ax = lc.get('axis', -1)
            ax = [] if ax is None else (list(ax) if isinstance(ax, (list, tuple)) else [ax])
            rank_in = nodes[ins[0]].get('rank', 4)
            if ax not in ([], [-1], [rank_in - 1]):
                raise PlanError('Normalization %s: only per-channel (or scalar) statistics are supported, got axis %s' % (name, ax))
            if lc.get('mean') is not None:
                mean, var = np.asarray(lc['mean'], np.float64), np.asarray(lc['variance'], np.float64)
            elif len(ws) >= 2:
                mean, var = np.asarray(ws[0], np.float64), np.asarray(ws[1], np.float64)
            else:
                raise PlanError('Normalization %s carries no statistics' % name)
            mean = np.broadcast_to(mean.reshape(-1), (c,)) if mean.size in (1, c) else None
            var = np.broadcast_to(var.reshape(-1), (c,)) if var.size in (1, c) else None
            if mean is None or var is None:
                raise PlanError('Normalization %s: statistics do not match %d channels' % (name, c))
            inv = 1.0 / np.maximum(np.sqrt(var), 1e-7)           # backend.epsilon()
            idx = add(name, 'affine', ins, (h, w, c), scale=inv.astype(np.float32), shift=(-mean * inv).astype(np.float32),
                      scale64=inv, shift64=-mean * inv, act=0, alpha=0.0)
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
            idx = add(name, 'act', ins, (h, w, c), **_act_pair(lc['activation']))
        elif cls == 'ReLU':
            ns = float(lc.get('negative_slope') or 0.0)
            mv = lc.get('max_value')
            if lc.get('threshold') or (mv is not None and ns):
                raise PlanError('ReLU %s: threshold / max_value together with negative_slope are not supported' % name)
            if mv is not None:                                   # relu6 of the MobileNets
                idx = add(name, 'act', ins, (h, w, c), act=ACT['relu_clip'], alpha=float(mv))
            else:
                idx = add(name, 'act', ins, (h, w, c), act=ACT['leaky_relu'] if ns else ACT['relu'], alpha=ns)
        elif cls == 'LeakyReLU':
            idx = add(name, 'act', ins, (h, w, c), act=ACT['leaky_relu'], alpha=float(lc.get('alpha', 0.3)))
        elif cls == 'ELU':
            idx = add(name, 'act', ins, (h, w, c), act=ACT['elu'], alpha=float(lc.get('alpha', 1.0)))
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
        if isinstance(output, str):
            if output not in by_name or output not in [o[0] for o in outs]:
                raise PlanError('%r is not an output of the model (outputs: %s)' % (output, [o[0] for o in outs]))
            out_node = by_name[output]
        else:
            if not 0 <= int(output) < len(outs):
                raise PlanError('output %r out of range: the model has %d output(s)' % (output, len(outs)))
            out_node = by_name[outs[int(output)][0]]
    in_nodes = [i for i, n in enumerate(nodes) if n['kind'] == 'input']
    if len(in_nodes) != 1:
        raise PlanError('models with %d inputs are not supported' % len(in_nodes))

    # ---------------------------------------------------------------- pass 2: peephole fusion
    alive = [True] * len(nodes)
    alias = list(range(len(nodes)))     # node -> node that now produces its value

    def consumers(i):
        return [j for j, n in enumerate(nodes) if alive[j] and i in n['inputs']]

    if fuse:
        # UpSampling2D(2, nearest) -> Conv2D(2x2, 'same') - the decoder step of the most common public Keras U-Net - as ONE stride-2
        # transposed convolution over the un-upsampled tensor (round 6).  With up(y) = in(y // 2) and 'same' padding
        # of an even kernel (nothing before, one row / column behind):
        #     out(2i)   = w0 in(i) + w1 in(i)          out(2i+1) = w0 in(i) + w1 in(i+1)
        # i.e. in(i) reaches out(2i-1), out(2i), out(2i+1) with w1, w0 + w1, w0: a 3-tap transposed kernel K' = [w1, w0 + w1, w0] with
        # one output row cropped at the top - in 2-D the nine taps K'[a][b] = sum of w[r][s] over r in R(a), s in R(b),
        # R = ({1}, {0, 1}, {0}) (1 + 2 + 2 + 4 = 9 products of pre-summed filters instead of 16 on a 4x-sized tensor that never exists).
        # The library runs k x k / stride-2 transposed convolutions as one 2x2-tap sub-pixel convolution on the matrix cores.
        for j, n in enumerate(nodes):
            if not alive[j] or n['kind'] != 'conv' or len(n['inputs']) != 1 or n.get('rank', 4) != 4:
                continue
            i = n['inputs'][0]
            u = nodes[i]
            if u['kind'] != 'upsample' or u['stride'] != 2 or u['mode'] != 0 or len(consumers(i)) != 1 or i == out_node:
                continue
            if (n['kh'], n['kw'], n.get('stride', 1), n.get('dilation', 1), n['pad_top'], n['pad_left'], n.get('aniso', 0)) != (2, 2, 1, 1, 0, 0, 0) or \
                    n['shape'][:2] != u['shape'][:2]:
                continue
            w = n['kernel'].astype(np.float64)                   # (2, 2, in, out)
            R = ((1,), (0, 1), (0,))
            kt = np.zeros((3, 3, w.shape[3], w.shape[2]))
            for a in range(3):
                for b in range(3):
                    kt[a, b] = sum(w[r, q] for r in R[a] for q in R[b]).T
            n.update(kind='convt', kh=3, kw=3, stride=2, pad_top=1, pad_left=1, kernel=np.ascontiguousarray(kt, np.float32), inputs=list(u['inputs']))
            n.pop('dilation', None)
            n.setdefault('also', []).append(u['name'] + ' (UpSampling2D folded into the convolution)')
            alive[i] = False
        for j, n in enumerate(nodes):
            if not alive[j] or len(n['inputs']) != 1:
                continue
            i = n['inputs'][0]
            p = nodes[i]
            if p['kind'] not in ('conv', 'convt', 'dwconv') or len(consumers(i)) != 1 or i == out_node:
                continue
            if n['kind'] == 'affine' and p['act'] == 0 and n['act'] == 0:
                s, t = n['scale64'], n['shift64']
                k = p['kernel'].astype(np.float64)
                if p['kind'] == 'dwconv':                       # (kh, kw, cin, mult): output channel = cin index * mult + j
                    p['kernel'] = (k * s.reshape(k.shape[2], k.shape[3])[None, None]).astype(np.float32)
                else:
                    p['kernel'] = (k * (s[None, None, None, :] if p['kind'] == 'conv' else s[None, None, :, None])).astype(np.float32)
                b = p['bias'].astype(np.float64) if p['bias'] is not None else np.zeros(len(s))
                p['bias'] = (b * s + t).astype(np.float32)
            elif n['kind'] == 'act' and p['act'] == 0 and (n['act'] <= CONV_ACT_MAX or p['kind'] == 'dwconv'):
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
            ok = (i not in view and nodes[i]['kind'] not in ('input', 'concat', 'reshape', 'slice') and n['inputs'].count(i) == 1
                  and i != out_node)
            if ok:
                view[i] = (j, off)
            else:
                copies[(j, pos)] = off
            off += ci

    # ---------------------------------------------------------------- pass 4: emit tensors / ops with buffer reuse
    plan = Plan()
    plan.channels_first = was_channels_first
    order = [j for j in range(len(nodes)) if alive[j]]
    last_use = {}
    for j in order:
        for i in nodes[j]['inputs']:
            last_use[i] = j
    last_use[out_node] = len(nodes) + 1
    # Flatten / Reshape / channel slices (grouped convolutions) are views of their input's buffer: the owner lives as long as the
    # view is read
    for j in reversed(order):
        if nodes[j]['kind'] in ('reshape', 'slice'):
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

    weight_ids = {}                     # id(array) -> index: the calls of a shared layer (unshare_layers) are several ops on ONE weight index

    def add_weight(arr):
        if arr is None:
            return -1
        if id(arr) in weight_ids and weight_ids[id(arr)][0] is arr:
            return weight_ids[id(arr)][1]
        plan.weights.append(np.ascontiguousarray(arr, np.float32).ravel())
        weight_ids[id(arr)] = (arr, len(plan.weights) - 1)
        return len(plan.weights) - 1

    def op(**kw):
        d = dict(op=0, in0=-1, in1=-1, out=-1, kh=0, kw=0, stride=1, pad_top=0, pad_left=0, act=0, mode=0, w0=-1, w1=-1, alpha=0.0, dilation=1)
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
        elif k == 'slice':
            i = n['inputs'][0]
            ti_ = plan.tensors[tensor_of[i]]
            h_, w_, c_ = n['shape']
            plan.tensors.append(dict(buffer=ti_['buffer'], h=h_, w=w_, c=c_, c_stride=ti_['c_stride'], c_offset=ti_['c_offset'] + n['c0']))
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
            late_act = k in ('conv', 'convt') and n['act'] > CONV_ACT_MAX
            conv_act = 0 if late_act else n.get('act', 0)
            if k == 'conv':
                op(op=OP_CONV, in0=ins[0], out=t, kh=n['kh'], kw=n['kw'], stride=n.get('stride', 1), pad_top=n['pad_top'], pad_left=n['pad_left'],
                   act=conv_act, alpha=n['alpha'], w0=add_weight(n['kernel']), w1=add_weight(n['bias']), dilation=n.get('dilation', 1),
                   mode=n.get('aniso', 0))
            elif k == 'dwconv':
                op(op=OP_DWCONV, in0=ins[0], out=t, kh=n['kh'], kw=n['kw'], stride=n['stride'], pad_top=n['pad_top'], pad_left=n['pad_left'],
                   act=n['act'], alpha=n['alpha'], mode=n['mult'], w0=add_weight(n['kernel']), w1=add_weight(n['bias']), dilation=n['dilation'])
            elif k == 'prelu':
                op(op=OP_PRELU, in0=ins[0], out=t, mode=n['mode'], w0=add_weight(n['slopes']))
            elif k == 'layernorm':
                op(op=OP_LAYERNORM, in0=ins[0], out=t, alpha=n['eps'], w0=add_weight(n['gamma']), w1=add_weight(n['beta']))
            elif k == 'convt':
                op(op=OP_CONVT, in0=ins[0], out=t, kh=n['kh'], kw=n['kw'], stride=n['stride'], pad_top=n['pad_top'],
                   pad_left=n['pad_left'], act=conv_act, alpha=n['alpha'], w0=add_weight(n['kernel']), w1=add_weight(n['bias']))
            elif k == 'maxpool':
                op(op=OP_MAXPOOL, in0=ins[0], out=t, kh=n['kh'], kw=n['kw'], stride=n['stride'], mode=n.get('mode', 0),
                   pad_top=n.get('pad_top', 0), pad_left=n.get('pad_left', 0))
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
                op(op=OP_ADD, in0=ins[0], in1=ins[1], out=t, mode=n.get('mode', 0))
                for extra in ins[2:]:
                    op(op=OP_ADD, in0=t, in1=extra, out=t, mode=n.get('mode', 0))
            elif k == 'copy':
                op(op=OP_COPY, in0=ins[0], out=t, pad_top=n['off_y'], pad_left=n['off_x'])
            else:
                raise PlanError('internal: unknown node kind %s' % k)
            if late_act:
                op(op=OP_ACT, in0=t, out=t, act=n['act'], alpha=n['alpha'])
        release_after(j)
    plan.output_tensor = tensor_of[out_node]
    plan.output_rank = 2 if nodes[out_node].get('rank', 4) == 2 else 4
    # the model output must be readable as a compact tensor for predict_on_batch; views are fine for the C side
    return plan

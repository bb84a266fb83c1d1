"""Oracle: forward pass of a Keras functional model (the metaseg U-Net) on the CPU.
TEST INFRASTRUCTURE ONLY.

Restates what ``model.predict_on_batch`` does at src/utils.py:115 for the layer types a Keras U-Net
``model_config`` holds.  The arithmetic lives in TensorFlow 2.8 (env.yml:11), which is absent from the reference
tree and not installed here -> PARITY UNPINNED; Keras layer semantics are restated from their published
definitions and evaluated with torch CPU float32 ops (``conv_numpy`` is an independent numpy cross-check):

* Conv2D: cross-correlation, HWIO kernel, ``same`` = TF padding (total ``k-1`` at stride 1, the extra pixel on
  the bottom/right), ``valid`` = none; bias; activation.
* Conv2DTranspose: kernel (kh, kw, out, in); ``out_full[i*s + k] += in[i] * w[k]``; ``same`` crops
  ``max(k - s, 0)`` (``// 2`` before, rest after) so that the output is ``in * s``.
* MaxPooling2D (valid, floor), UpSampling2D (nearest | bilinear with half-pixel centres), Concatenate(axis=-1),
  BatchNormalization (inference; weight order gamma, beta, moving_mean, moving_variance, honouring center/scale),
  Dropout-like layers = identity, Activation / ReLU / LeakyReLU / Softmax, Add, ZeroPadding2D, Cropping2D,
  Rescaling; for the interSeg classifiers (src/interseg.py:96-98,155,168): strided Conv2D, AveragePooling2D,
  GlobalAveragePooling2D / GlobalMaxPooling2D, Flatten (NHWC order), Reshape, Dense (last axis).
* Round 5 - whatever else a Keras file given to ``load_model`` (src/utils.py:27-33, src/interseg.py:96-98) may hold:
  Conv2D ``dilation_rate`` (taps ``d`` pixels apart; 'same' pads for the dilated extent ``(k - 1) d + 1``) and ``groups``
  (output channel ``o`` reads the input channels of group ``o // (filters / groups)``), DepthwiseConv2D (kernel
  (kh, kw, cin, m); output channel ``ci * m + j``), SeparableConv2D (depthwise, then 1x1 pointwise + bias + activation),
  Multiply / Subtract / Maximum / Minimum / Average (numpy broadcasting), PReLU (``shared_axes``), LayerNormalization (last
  axis; moments then ``(x - mean) / sqrt(var + eps) * gamma + beta``), Normalization (``(x - mean) / max(sqrt(var), 1e-7)``),
  ReLU(max_value), ELU(alpha), the activation names selu / softplus / softsign / swish / gelu / hard_sigmoid / exponential /
  relu6, 'same' pooling (the average runs over the pixels inside the input), nested Functional / Sequential sub-models
  (evaluated recursively; their weights arrive as ``{inner layer: [arrays]}`` or as a list with ``.names``) and models with
  several outputs (``output=`` selects one).
The uint8 patch batch is cast to float32 without scaling (Keras casts inputs to the InputLayer dtype).
"""
import json

import numpy as np
import torch
import torch.nn.functional as F


def _same_pad(k, s, n):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def _act(name, x, cfg=None):
    if name in (None, 'linear'):
        return x
    if name == 'relu':
        return F.relu(x)
    if name == 'sigmoid':
        return torch.sigmoid(x)
    if name == 'softmax':
        return F.softmax(x, dim=1)
    if name == 'tanh':
        return torch.tanh(x)
    if name == 'elu':
        return F.elu(x)
    if name == 'relu6':
        return torch.clamp(x, 0.0, 6.0)
    if name == 'selu':
        return F.selu(x)
    if name == 'softplus':
        return F.softplus(x)
    if name == 'softsign':
        return F.softsign(x)
    if name in ('swish', 'silu'):
        return x * torch.sigmoid(x)
    if name == 'gelu':
        return F.gelu(x)                                     # exact erf form = Keras' default (approximate=False)
    if name == 'hard_sigmoid':
        return torch.clamp(0.2 * x + 0.5, 0.0, 1.0)         # Keras 2.x definition
    if name == 'exponential':
        return torch.exp(x)
    raise NotImplementedError('activation %r' % name)


def _dil(cfg):
    d = cfg.get('dilation_rate', [1, 1])
    return (d, d) if isinstance(d, int) else tuple(d)


def _conv2d(x, cfg, w):
    kernel = torch.from_numpy(np.ascontiguousarray(w[0])).permute(3, 2, 0, 1).contiguous()
    bias = torch.from_numpy(np.ascontiguousarray(w[1])) if cfg.get('use_bias', True) else None
    kh, kw = kernel.shape[2:]
    sh, sw = cfg.get('strides', [1, 1])
    dh, dw = _dil(cfg)
    if cfg['padding'] == 'same':
        pt, pb = _same_pad((kh - 1) * dh + 1, sh, x.shape[2])
        pl, pr = _same_pad((kw - 1) * dw + 1, sw, x.shape[3])
        x = F.pad(x, (pl, pr, pt, pb))
    # Keras' grouped kernel (kh, kw, cin / groups, filters) permuted to (filters, cin / groups, kh, kw) is torch's layout as it is
    y = F.conv2d(x, kernel, bias, stride=(sh, sw), dilation=(dh, dw), groups=int(cfg.get('groups', 1) or 1))
    return _act(cfg.get('activation'), y)


def _depthwise(x, cfg, kernel_hwcm, bias, act):
    """kernel (kh, kw, cin, m): output channel ci * m + j = sum over taps of in[ci] * kernel[:, :, ci, j]."""
    kh, kw, ci, m = kernel_hwcm.shape
    wt = torch.from_numpy(np.ascontiguousarray(kernel_hwcm)).permute(2, 3, 0, 1).reshape(ci * m, 1, kh, kw).contiguous()
    sh, sw = cfg.get('strides', [1, 1])
    dh, dw = _dil(cfg)
    if cfg['padding'] == 'same':
        pt, pb = _same_pad((kh - 1) * dh + 1, sh, x.shape[2])
        pl, pr = _same_pad((kw - 1) * dw + 1, sw, x.shape[3])
        x = F.pad(x, (pl, pr, pt, pb))
    y = F.conv2d(x, wt, None if bias is None else torch.from_numpy(np.ascontiguousarray(bias)), stride=(sh, sw), dilation=(dh, dw), groups=ci)
    return _act(act, y)


def _pool_same(a, lc, avg):
    """Pooling with padding='same': TensorFlow pads (smaller half in front) with -inf for the maximum and leaves the padding
    out of the average's divisor."""
    kh, kw = lc['pool_size']
    sh, sw = lc.get('strides') or lc['pool_size']
    pt, pb = _same_pad(kh, sh, a.shape[2])
    pl, pr = _same_pad(kw, sw, a.shape[3])
    if not avg:
        return F.max_pool2d(F.pad(a, (pl, pr, pt, pb), value=float('-inf')), (kh, kw), (sh, sw))
    ones = F.pad(torch.ones_like(a[:1, :1]), (pl, pr, pt, pb))
    s = F.avg_pool2d(F.pad(a, (pl, pr, pt, pb)), (kh, kw), (sh, sw)) * (kh * kw)
    cnt = F.avg_pool2d(ones, (kh, kw), (sh, sw)) * (kh * kw)
    return s / cnt


def _nested_weights(ws):
    if ws is None or isinstance(ws, dict):
        return ws or {}
    names = getattr(ws, 'names', None)
    assert names is not None and len(names) == len(ws), 'weights of a nested model need names'
    d = {}
    for nm, a in zip(names, ws):
        parts = str(nm).split(':')[0].split('/')
        d.setdefault(parts[-2] if len(parts) >= 2 else parts[0], []).append(a)
    return d


def _conv2d_transpose(x, cfg, w):
    kernel = torch.from_numpy(np.ascontiguousarray(w[0])).permute(3, 2, 0, 1).contiguous()   # (in, out, kh, kw)
    bias = torch.from_numpy(np.ascontiguousarray(w[1])) if cfg.get('use_bias', True) else None
    kh, kw = kernel.shape[2:]
    sh, sw = cfg['strides']
    y = F.conv_transpose2d(x, kernel, None, stride=(sh, sw))
    if cfg['padding'] == 'same':
        H, W = x.shape[2] * sh, x.shape[3] * sw
        ct, cl = max(kh - sh, 0) // 2, max(kw - sw, 0) // 2
        y = y[:, :, ct:ct + H, cl:cl + W]
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return _act(cfg.get('activation'), y)


def _batchnorm(x, cfg, w):
    w = list(w)
    C = x.shape[1]
    gamma = torch.from_numpy(w.pop(0)) if cfg.get('scale', True) else torch.ones(C)
    beta = torch.from_numpy(w.pop(0)) if cfg.get('center', True) else torch.zeros(C)
    mean, var = torch.from_numpy(w[0]), torch.from_numpy(w[1])
    inv = gamma / torch.sqrt(var + cfg.get('epsilon', 1e-3))
    return x * inv.view(1, -1, 1, 1) + (beta - mean * inv).view(1, -1, 1, 1)


def _tfop(L, lc, a):
    """TFOpLambda arithmetic with one tensor and one Python constant."""
    fn = lc.get('function', '')
    node = L['inbound_nodes'][0]
    kw = node[3] if (node and isinstance(node[0], str)) else node[0][3]
    k = kw.get('y', kw.get('x'))
    if fn in ('cast', 'identity', 'stop_gradient'):
        return a
    if fn in ('math.truediv', 'math.divide', '__operators__.truediv'):
        return a / float(k)
    if fn in ('math.multiply', '__operators__.mul'):
        return a * float(k)
    if fn in ('math.add', '__operators__.add'):
        return a + float(k)
    if fn in ('math.subtract', '__operators__.sub'):
        return a - float(k)
    raise NotImplementedError('TFOpLambda %s' % fn)


def _calls_in_order(layers):
    """Layers that Keras calls more than once (shared weights: several entries in ``inbound_nodes``) as one evaluation step per CALL,
    ordered so that every call comes after the calls it consumes.  -> list of (layer dict with ONE inbound node whose references name
    calls as ``layer@k``, key ``layer@k`` under which the call's value is stored, name of the layer whose weights it uses).
    A Functional model evaluates node k of a layer on the tensors [layer, node index, tensor index] of its inbound node k
    (tf.keras functional API; the reference loads such files like any other: /root/reference/src/utils.py:27-33)."""
    def refs_of(node):
        return [node] if node and isinstance(node[0], str) else list(node)

    pending = []
    for L in layers:
        name = L['config']['name']
        nodes = L.get('inbound_nodes', [])
        if not nodes:
            pending.append((L, '%s@0' % name, name, []))
        for k, node in enumerate(nodes):
            refs = [[('%s@%d' % (r[0], r[1] if len(r) > 1 and isinstance(r[1], int) else 0))] + [0] + list(r[2:]) for r in refs_of(node)]
            pending.append((dict(L, inbound_nodes=[refs]), '%s@%d' % (name, k), name, [r[0] for r in refs]))
    ready, ordered = set(), []
    while pending:
        rest = []
        for item in pending:
            if all(d in ready for d in item[3]):
                ordered.append(item[:3])
                ready.add(item[1])
            else:
                rest.append(item)
        assert len(rest) < len(pending), 'the layer graph has a cycle or a dangling reference'
        pending = rest
    return ordered


def forward(model_config, weights, x_nhwc, lambda_fns=None, dtype=np.float32, output=0):
    """``model_config``: dict (or JSON text) of a Keras Functional/Sequential model; ``weights``: {layer name:
    [arrays in Keras order]}; ``x_nhwc``: (N, H, W, C) any dtype.  Returns float32 NHWC output of the model.

    ``dtype=np.float64`` evaluates the same graph with the same float32 weights in double precision and returns float64:
    the adjudicator between two float32 evaluations that disagree (tools/label_mismatch.py, tests/test_gpu_configs.py) -
    its rounding error is 2^-29 of a float32 evaluation's, so ``float32(forward(..., dtype=float64))`` is what an exactly
    rounded float32 network would return."""
    if isinstance(model_config, (str, bytes)):
        model_config = json.loads(model_config)
    cfg = model_config['config']
    layers = cfg['layers'] if isinstance(cfg, dict) else cfg
    seq = model_config['class_name'] == 'Sequential'
    # channels_first models (round 6): Keras keeps (N, C, H, W) tensors and (kh, kw, in, out) kernels; the channel axis of every
    # axis-taking layer is then 1 instead of 3 / -1.  Every value below is NCHW either way: only the boundary and the axis checks differ.
    cf = any(L['config'].get('data_format') == 'channels_first' for L in layers)
    chan = (lambda ax: ax in (1, -3, [1], [-3])) if cf else (lambda ax: ax in (-1, 3, [-1], [3]))
    vals = {}
    if isinstance(x_nhwc, (list, tuple)) and x_nhwc and torch.is_tensor(x_nhwc[0]):
        xs = list(x_nhwc)                     # a nested model called on the (NCHW / (N, K)) tensors of its parent
    else:
        x_nhwc = np.ascontiguousarray(x_nhwc)
        if x_nhwc.ndim == 3:                  # (N, H, W): a model whose InputLayer has no channel axis
            x_nhwc = x_nhwc[..., None]
        if cf:                                # a channels_first model takes (N, C, H, W): torch's own layout, nothing to move
            xs = [torch.from_numpy(x_nhwc.astype(dtype)).contiguous()]
        else:
            xs = [torch.from_numpy(x_nhwc.astype(dtype)).permute(0, 3, 1, 2).contiguous()]
    x = xs[0]
    n_in = 0
    prev = None
    # one evaluation step per CALL of a layer: (layer with one inbound node, key of the call's value, layer whose weights it uses)
    shared = not seq and any(len(L.get('inbound_nodes', [])) > 1 for L in layers)
    steps = _calls_in_order(layers) if shared else [(L, L['config']['name'], L['config']['name']) for L in layers]
    okey = (lambda r: '%s@%d' % (r[0], r[1] if len(r) > 1 and isinstance(r[1], int) else 0)) if shared else (lambda r: r[0])
    with torch.no_grad():
        for L, vkey, name in steps:
            cls, lc = L['class_name'], L['config']
            if cls == 'InputLayer':
                order = [r[0] for r in cfg['input_layers']] if isinstance(cfg, dict) and cfg.get('input_layers') else None
                vals[vkey] = xs[order.index(name) if order and name in order else n_in]
                n_in += 1
                prev = vkey
                continue
            if seq:
                if prev is None:
                    vals['__in__'] = x
                    prev = '__in__'
                ins = [vals[prev]]
            else:
                nodes = L['inbound_nodes']
                assert len(nodes) == 1, 'one step per call (_calls_in_order)'
                node = nodes[0]
                if node and isinstance(node[0], str):
                    node = [node]
                ins = [vals[n[0]][n[2]] if isinstance(vals[n[0]], list) else vals[n[0]] for n in node]
            a = ins[0]
            if cls in ('Functional', 'Model', 'Sequential'):    # a nested model: evaluate it on this layer's inputs
                outs = forward({'class_name': cls, 'config': lc}, _nested_weights(weights.get(name)), ins, lambda_fns=lambda_fns,
                               dtype=dtype, output=None)
                vals[vkey] = outs if len(outs) > 1 else outs[0]
                prev = vkey
                continue
            w = [np.asarray(v, dtype) for v in weights.get(name, [])]
            if cls == 'Conv2D':
                y = _conv2d(a, lc, w)
            elif cls == 'DepthwiseConv2D':
                y = _depthwise(a, lc, w[0], w[1] if lc.get('use_bias', True) else None, lc.get('activation'))
            elif cls == 'SeparableConv2D':
                y = _depthwise(a, lc, w[0], None, None)
                pk = torch.from_numpy(np.ascontiguousarray(w[1])).permute(3, 2, 0, 1).contiguous()
                y = F.conv2d(y, pk, torch.from_numpy(np.ascontiguousarray(w[2])) if lc.get('use_bias', True) else None)
                y = _act(lc.get('activation'), y)
            elif cls == 'Conv2DTranspose':
                y = _conv2d_transpose(a, lc, w)
            elif cls == 'MaxPooling2D':
                if lc.get('padding', 'valid') == 'same':
                    y = _pool_same(a, lc, False)
                else:
                    y = F.max_pool2d(a, tuple(lc['pool_size']), tuple(lc.get('strides') or lc['pool_size']))
            elif cls == 'AveragePooling2D':
                if lc.get('padding', 'valid') == 'same':
                    y = _pool_same(a, lc, True)
                else:
                    y = F.avg_pool2d(a, tuple(lc['pool_size']), tuple(lc.get('strides') or lc['pool_size']))
            elif cls == 'GlobalAveragePooling2D':
                y = a.mean(dim=(2, 3))
                if lc.get('keepdims'):
                    y = y[:, :, None, None]
            elif cls == 'GlobalMaxPooling2D':
                y = a.amax(dim=(2, 3))
                if lc.get('keepdims'):
                    y = y[:, :, None, None]
            elif cls == 'Flatten':
                # (a channels_first tensor is flattened in (C, H, W) order unless the layer itself says data_format = channels_first,
                # in which case Keras moves the channels last first)
                nhwc_order = a.dim() == 4 and (not cf or lc.get('data_format') == 'channels_first')
                y = a.permute(0, 2, 3, 1).reshape(a.shape[0], -1) if nhwc_order else a.reshape(a.shape[0], -1)
            elif cls == 'Reshape':
                ts = [int(v) for v in lc['target_shape']]
                flat = a.permute(0, 2, 3, 1).reshape(a.shape[0], -1) if a.dim() == 4 else a.reshape(a.shape[0], -1)
                if len(ts) == 1:
                    y = flat
                else:
                    if len(ts) == 2:
                        ts = ts + [1]
                    y = flat.reshape(a.shape[0], ts[0], ts[1], ts[2]).permute(0, 3, 1, 2).contiguous()
            elif cls == 'Dense':
                kernel = torch.from_numpy(np.ascontiguousarray(w[0]))            # (features, units), acts on the last axis
                bias = torch.from_numpy(np.ascontiguousarray(w[1])) if lc.get('use_bias', True) else None
                if a.dim() == 4:
                    y = torch.einsum('nchw,cu->nuhw', a, kernel)
                    if bias is not None:
                        y = y + bias.view(1, -1, 1, 1)
                    y = _act(lc.get('activation'), y)
                else:
                    y = a @ kernel
                    if bias is not None:
                        y = y + bias
                    y = _act(lc.get('activation'), y)
            elif cls == 'UpSampling2D':
                sz = tuple(lc['size'])
                if lc.get('interpolation', 'nearest') == 'nearest':
                    y = a.repeat_interleave(sz[0], dim=2).repeat_interleave(sz[1], dim=3)
                else:
                    y = F.interpolate(a, scale_factor=sz, mode='bilinear', align_corners=False)
            elif cls == 'Concatenate':
                assert chan(lc.get('axis', -1)), 'Concatenate over the channel axis only'
                y = torch.cat(ins, dim=1)
            elif cls in ('Add', 'Multiply', 'Subtract', 'Maximum', 'Minimum', 'Average'):
                # NCHW tensors broadcast exactly as their NHWC originals do (extents of 1 stretch)
                y = ins[0]
                for t in ins[1:]:
                    y = (y + t if cls in ('Add', 'Average') else y * t if cls == 'Multiply' else y - t if cls == 'Subtract'
                         else torch.maximum(y, t) if cls == 'Maximum' else torch.minimum(y, t))
                if cls == 'Average':
                    y = y / float(len(ins))
            elif cls == 'PReLU':
                al = torch.from_numpy(np.ascontiguousarray(w[0]))
                al = al.permute(2, 0, 1) if a.dim() == 4 and not cf else al          # (h, w, c) with 1s on the shared axes -> (c, h, w); channels_first: (c, h, w) already
                y = torch.where(a > 0, a, al * a)
            elif cls == 'LayerNormalization':
                ax = lc.get('axis', -1)
                ax = list(ax) if isinstance(ax, (list, tuple)) else [ax]
                assert ax in ([-1], [a.dim() - 1]), 'LayerNormalization over the last axis only'
                wl = list(w)
                gamma = torch.from_numpy(wl.pop(0)) if lc.get('scale', True) else None
                beta = torch.from_numpy(wl.pop(0)) if lc.get('center', True) else None
                mean = a.mean(dim=1, keepdim=True)
                var = ((a - mean) ** 2).mean(dim=1, keepdim=True)
                y = (a - mean) / torch.sqrt(var + lc.get('epsilon', 1e-3))
                shp = (1, -1, 1, 1) if a.dim() == 4 else (1, -1)
                if gamma is not None:
                    y = y * gamma.view(shp)
                if beta is not None:
                    y = y + beta.view(shp)
            elif cls == 'Normalization':
                if lc.get('mean') is not None:
                    mean, var = np.asarray(lc['mean'], dtype), np.asarray(lc['variance'], dtype)
                else:
                    mean, var = w[0], w[1]
                shp = (1, -1, 1, 1) if a.dim() == 4 else (1, -1)
                mean = torch.from_numpy(np.ascontiguousarray(mean).reshape(-1)).view(shp)
                std = torch.clamp(torch.sqrt(torch.from_numpy(np.ascontiguousarray(var).reshape(-1))), min=1e-7).view(shp)
                y = (a - mean) / std
            elif cls == 'BatchNormalization':
                y = _batchnorm(a, lc, w)
            elif cls in ('Dropout', 'SpatialDropout2D', 'GaussianNoise', 'GaussianDropout', 'AlphaDropout'):
                y = a
            elif cls == 'Activation':
                y = _act(lc['activation'], a)
            elif cls == 'ReLU':
                assert not lc.get('threshold')
                if lc.get('max_value') is not None:
                    assert not lc.get('negative_slope')
                    y = torch.clamp(a, 0.0, float(lc['max_value']))
                elif lc.get('negative_slope'):
                    y = F.leaky_relu(a, float(lc['negative_slope']))
                else:
                    y = F.relu(a)
            elif cls == 'ELU':
                y = F.elu(a, float(lc.get('alpha', 1.0)))
            elif cls == 'LeakyReLU':
                y = F.leaky_relu(a, float(lc.get('alpha', 0.3)))
            elif cls == 'Softmax':
                assert chan(lc.get('axis', -1)) or a.dim() == 2, 'Softmax over the channel axis only'
                y = F.softmax(a, dim=1)
            elif cls == 'ZeroPadding2D':
                (t, b), (l, r) = lc['padding']
                y = F.pad(a, (l, r, t, b))
            elif cls == 'Cropping2D':
                (t, b), (l, r) = lc['cropping']
                y = a[:, :, t:a.shape[2] - b, l:a.shape[3] - r]
            elif cls == 'TFOpLambda':
                y = _tfop(L, lc, a)
            elif cls == 'Lambda':
                y = (lambda_fns or {})[name](a)
            elif cls == 'Rescaling':
                y = a * float(lc['scale']) + float(lc.get('offset', 0.0))
            else:
                raise NotImplementedError('Keras layer %s' % cls)
            vals[vkey] = y
            prev = vkey
        if seq:
            outs = [vals[prev]]
        else:
            outs = [vals[okey(r)][r[2]] if isinstance(vals[okey(r)], list) else vals[okey(r)] for r in cfg['output_layers']]
        if output is None:                    # (nested call: torch tensors of every output)
            return outs
        out = outs[[r[0] for r in cfg['output_layers']].index(output)] if isinstance(output, str) else outs[int(output)]
    if out.dim() == 2 or cf:
        return out.contiguous().numpy()
    return out.permute(0, 2, 3, 1).contiguous().numpy()


def conv_numpy(x_nhwc, kernel_hwio, bias, padding='same', stride=1):
    """Independent numpy float32 Conv2D used to cross-check the torch path on small inputs.  'same' at stride s pads a
    total of max((ceil(n / s) - 1) s + k - n, 0), the smaller half in front (TensorFlow's rule)."""
    x = np.asarray(x_nhwc, np.float32)
    kh, kw, ci, co = kernel_hwio.shape
    s = int(stride)
    if padding == 'same':
        pt, pb = _same_pad(kh, s, x.shape[1])
        pl, pr = _same_pad(kw, s, x.shape[2])
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    N, H, W, _ = x.shape
    Ho, Wo = (H - kh) // s + 1, (W - kw) // s + 1
    y = np.zeros((N, Ho, Wo, co), np.float32)
    for i in range(kh):
        for j in range(kw):
            y += np.einsum('nhwc,co->nhwo', x[:, i:i + (Ho - 1) * s + 1:s, j:j + (Wo - 1) * s + 1:s, :], kernel_hwio[i, j]).astype(np.float32)
    return y + np.asarray(bias, np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# Independent numpy restatements of the other Keras layer types (no torch), written from the layers' published
# definitions; tests/test_oracle_layers.py checks them against hand-computed known answers AND against the torch path
# above, so that every layer type of ``forward`` rests on two independent implementations.
# ---------------------------------------------------------------------------------------------------------------------
def conv_transpose_numpy(x_nhwc, kernel_hwoi, bias, stride, padding='same'):
    """Conv2DTranspose: every input pixel stamps ``in[c] * w[kh, kw, o, c]`` at output rows ``i * s + kh``; 'same' crops
    the full ((n - 1) s + k) result to n * s, dropping ``max(k - s, 0) // 2`` rows/columns in front (the padding TF's
    forward 'same' convolution of that stride would have put in front)."""
    x = np.asarray(x_nhwc, np.float32)
    kh, kw, co, ci = kernel_hwoi.shape
    N, Hi, Wi, _ = x.shape
    s = int(stride)
    full = np.zeros((N, (Hi - 1) * s + max(kh, s), (Wi - 1) * s + max(kw, s), co), np.float32)
    for a in range(kh):
        for b in range(kw):
            contrib = np.einsum('nhwc,oc->nhwo', x, kernel_hwoi[a, b]).astype(np.float32)
            full[:, a:a + (Hi - 1) * s + 1:s, b:b + (Wi - 1) * s + 1:s, :] += contrib
    if padding == 'same':
        ct, cl = max(kh - s, 0) // 2, max(kw - s, 0) // 2
        full = full[:, ct:ct + Hi * s, cl:cl + Wi * s, :]
    if bias is not None:
        full = full + np.asarray(bias, np.float32)
    return full


def upsample_numpy(x_nhwc, factor, interpolation='nearest'):
    """UpSampling2D.  nearest: repeat.  bilinear: tf.image.resize semantics with half-pixel centres, no antialiasing:
    source coordinate ``(dst + 0.5) / f - 0.5``, neighbours clamped to the image, weights from the fractional part."""
    x = np.asarray(x_nhwc, np.float32)
    f = int(factor)
    if interpolation == 'nearest':
        return x.repeat(f, axis=1).repeat(f, axis=2)

    def axis_weights(n):
        src = (np.arange(n * f) + 0.5) / f - 0.5
        lo = np.floor(src)
        frac = (src - lo).astype(np.float32)
        i0 = np.clip(lo, 0, n - 1).astype(int)
        i1 = np.clip(lo + 1, 0, n - 1).astype(int)
        return i0, i1, frac

    N, Hh, Ww, C = x.shape
    r0, r1, rf = axis_weights(Hh)
    c0, c1, cf = axis_weights(Ww)
    rows = x[:, r0] * (1 - rf)[None, :, None, None] + x[:, r1] * rf[None, :, None, None]
    return (rows[:, :, c0] * (1 - cf)[None, None, :, None] + rows[:, :, c1] * cf[None, None, :, None]).astype(np.float32)


def batchnorm_numpy(x_nhwc, gamma, beta, mean, var, eps=1e-3):
    """BatchNormalization at inference: ``gamma * (x - mean) / sqrt(var + eps) + beta`` per channel."""
    x = np.asarray(x_nhwc, np.float64)
    g = np.ones(x.shape[-1]) if gamma is None else np.asarray(gamma, np.float64)
    b = np.zeros(x.shape[-1]) if beta is None else np.asarray(beta, np.float64)
    return (g * (x - np.asarray(mean, np.float64)) / np.sqrt(np.asarray(var, np.float64) + eps) + b).astype(np.float32)


def maxpool_numpy(x_nhwc, k=2, s=2):
    """MaxPooling2D, 'valid' (floor)."""
    x = np.asarray(x_nhwc, np.float32)
    N, Hh, Ww, C = x.shape
    Ho, Wo = (Hh - k) // s + 1, (Ww - k) // s + 1
    out = np.full((N, Ho, Wo, C), -np.inf, np.float32)
    for a in range(k):
        for b in range(k):
            out = np.maximum(out, x[:, a:a + (Ho - 1) * s + 1:s, b:b + (Wo - 1) * s + 1:s, :])
    return out


def avgpool_numpy(x_nhwc, k=2, s=2):
    """AveragePooling2D, 'valid' (floor)."""
    x = np.asarray(x_nhwc, np.float32)
    N, Hh, Ww, C = x.shape
    Ho, Wo = (Hh - k) // s + 1, (Ww - k) // s + 1
    out = np.zeros((N, Ho, Wo, C), np.float64)
    for a in range(k):
        for b in range(k):
            out += x[:, a:a + (Ho - 1) * s + 1:s, b:b + (Wo - 1) * s + 1:s, :]
    return (out / (k * k)).astype(np.float32)


def dense_numpy(x, kernel, bias):
    """Dense: acts on the last axis."""
    y = np.asarray(x, np.float64) @ np.asarray(kernel, np.float64)
    return (y + (0 if bias is None else np.asarray(bias, np.float64))).astype(np.float32)


def softmax_numpy(x_nhwc):
    x = np.asarray(x_nhwc, np.float64)
    e = np.exp(x - x.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


# ---- round 5: numpy restatements of the wider layer vocabulary (no torch) ----------------------------------------
def conv_general_numpy(x_nhwc, kernel, bias, padding='same', stride=1, dilation=1, groups=1):
    """Conv2D with ``dilation_rate`` and ``groups``: tap (i, j) reads the input ``i * d`` rows / ``j * d`` columns from the
    window's origin; 'same' pads for the dilated kernel extent (k - 1) d + 1; kernel (kh, kw, cin / groups, filters) and
    output channel o reads the input channels of group o // (filters / groups)."""
    x = np.asarray(x_nhwc, np.float32)
    kh, kw, cg, co = kernel.shape
    # ``stride`` / ``dilation``: one number, or (vertical, horizontal) as Keras' strides / dilation_rate lists
    (s, sx), (d, dx) = [(int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[1])) for v in (stride, dilation)]
    g = int(groups)
    ekh, ekw = (kh - 1) * d + 1, (kw - 1) * dx + 1
    if padding == 'same':
        pt, pb = _same_pad(ekh, s, x.shape[1])
        pl, pr = _same_pad(ekw, sx, x.shape[2])
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    N, H, W, C = x.shape
    assert C == cg * g and co % g == 0
    Ho, Wo = (H - ekh) // s + 1, (W - ekw) // sx + 1
    fg = co // g
    y = np.zeros((N, Ho, Wo, co), np.float64)
    for i in range(kh):
        for j in range(kw):
            win = x[:, i * d:i * d + (Ho - 1) * s + 1:s, j * dx:j * dx + (Wo - 1) * sx + 1:sx, :].astype(np.float64)
            for q in range(g):
                y[..., q * fg:(q + 1) * fg] += np.einsum('nhwc,co->nhwo', win[..., q * cg:(q + 1) * cg], kernel[i, j, :, q * fg:(q + 1) * fg].astype(np.float64))
    return (y + (0 if bias is None else np.asarray(bias, np.float64))).astype(np.float32)


def depthwise_numpy(x_nhwc, kernel_hwcm, bias, padding='same', stride=1, dilation=1):
    """DepthwiseConv2D: output channel ``ci * m + j`` is the 2-D correlation of input channel ``ci`` with
    ``kernel[:, :, ci, j]`` - no sum over channels."""
    x = np.asarray(x_nhwc, np.float32)
    kh, kw, ci, m = kernel_hwcm.shape
    s, d = int(stride), int(dilation)
    ekh, ekw = (kh - 1) * d + 1, (kw - 1) * d + 1
    if padding == 'same':
        pt, pb = _same_pad(ekh, s, x.shape[1])
        pl, pr = _same_pad(ekw, s, x.shape[2])
        x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    N, H, W, C = x.shape
    Ho, Wo = (H - ekh) // s + 1, (W - ekw) // s + 1
    y = np.zeros((N, Ho, Wo, ci, m), np.float64)
    for i in range(kh):
        for j in range(kw):
            win = x[:, i * d:i * d + (Ho - 1) * s + 1:s, j * d:j * d + (Wo - 1) * s + 1:s, :].astype(np.float64)
            y += win[..., None] * kernel_hwcm[i, j].astype(np.float64)
    y = y.reshape(N, Ho, Wo, ci * m)
    return (y + (0 if bias is None else np.asarray(bias, np.float64))).astype(np.float32)


def pool_same_numpy(x_nhwc, k, s, avg):
    """Pooling with padding='same': windows start ``pad_front`` before the image; only pixels inside it take part."""
    x = np.asarray(x_nhwc, np.float32)
    N, H, W, C = x.shape
    Ho, Wo = -(-H // s), -(-W // s)
    pt, pl = _same_pad(k, s, H)[0], _same_pad(k, s, W)[0]
    out = np.zeros((N, Ho, Wo, C), np.float32)
    for oy in range(Ho):
        for ox in range(Wo):
            y0, x0 = oy * s - pt, ox * s - pl
            win = x[:, max(y0, 0):min(y0 + k, H), max(x0, 0):min(x0 + k, W), :]
            out[:, oy, ox] = win.mean(axis=(1, 2)) if avg else win.max(axis=(1, 2))
    return out


def layernorm_numpy(x, gamma, beta, eps=1e-3):
    """LayerNormalization over the last axis."""
    x = np.asarray(x, np.float64)
    mean = x.mean(-1, keepdims=True)
    var = ((x - mean) ** 2).mean(-1, keepdims=True)
    y = (x - mean) / np.sqrt(var + eps)
    if gamma is not None:
        y = y * np.asarray(gamma, np.float64)
    if beta is not None:
        y = y + np.asarray(beta, np.float64)
    return y.astype(np.float32)


def prelu_numpy(x_nhwc, alpha):
    """PReLU: ``alpha`` has the input's (h, w, c) shape with 1 on the shared axes."""
    x = np.asarray(x_nhwc, np.float32)
    return np.where(x > 0, x, np.asarray(alpha, np.float32) * x).astype(np.float32)


def activation_numpy(name, x, alpha=None):
    """The activation functions by their published formulas (float64 inside)."""
    from math import erf
    x = np.asarray(x, np.float64)
    if name == 'relu6':
        y = np.clip(x, 0, 6)
    elif name == 'selu':
        y = 1.0507009873554805 * np.where(x > 0, x, 1.6732632423543772 * (np.exp(x) - 1))
    elif name == 'softplus':
        y = np.log1p(np.exp(x))
    elif name == 'softsign':
        y = x / (1 + np.abs(x))
    elif name == 'swish':
        y = x / (1 + np.exp(-x))
    elif name == 'gelu':
        y = 0.5 * x * (1 + np.vectorize(erf)(x / np.sqrt(2.0)))
    elif name == 'hard_sigmoid':
        y = np.clip(0.2 * x + 0.5, 0, 1)
    elif name == 'exponential':
        y = np.exp(x)
    elif name == 'elu':
        y = np.where(x > 0, x, (1.0 if alpha is None else alpha) * (np.exp(x) - 1))
    else:
        raise NotImplementedError(name)
    return y.astype(np.float32)

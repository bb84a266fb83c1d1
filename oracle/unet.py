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
    raise NotImplementedError('activation %r' % name)


def _conv2d(x, cfg, w):
    kernel = torch.from_numpy(np.ascontiguousarray(w[0])).permute(3, 2, 0, 1).contiguous()
    bias = torch.from_numpy(np.ascontiguousarray(w[1])) if cfg.get('use_bias', True) else None
    kh, kw = kernel.shape[2:]
    sh, sw = cfg.get('strides', [1, 1])
    assert list(cfg.get('dilation_rate', [1, 1])) == [1, 1] and cfg.get('groups', 1) == 1
    if cfg['padding'] == 'same':
        pt, pb = _same_pad(kh, sh, x.shape[2])
        pl, pr = _same_pad(kw, sw, x.shape[3])
        x = F.pad(x, (pl, pr, pt, pb))
    y = F.conv2d(x, kernel, bias, stride=(sh, sw))
    return _act(cfg.get('activation'), y)


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


def forward(model_config, weights, x_nhwc, lambda_fns=None, dtype=np.float32):
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
    vals = {}
    x_nhwc = np.ascontiguousarray(x_nhwc)
    if x_nhwc.ndim == 3:                      # (N, H, W): a model whose InputLayer has no channel axis
        x_nhwc = x_nhwc[..., None]
    x = torch.from_numpy(x_nhwc.astype(dtype)).permute(0, 3, 1, 2).contiguous()
    prev = None
    with torch.no_grad():
        for L in layers:
            cls, lc = L['class_name'], L['config']
            name = lc['name']
            if cls == 'InputLayer':
                vals[name] = x
                prev = name
                continue
            if seq:
                if prev is None:
                    vals['__in__'] = x
                    prev = '__in__'
                ins = [vals[prev]]
            else:
                nodes = L['inbound_nodes']
                assert len(nodes) == 1, 'shared layers are not supported'
                node = nodes[0]
                if node and isinstance(node[0], str):
                    node = [node]
                ins = [vals[n[0]] for n in node]
            w = [np.asarray(v, dtype) for v in weights.get(name, [])]
            a = ins[0]
            if cls == 'Conv2D':
                y = _conv2d(a, lc, w)
            elif cls == 'Conv2DTranspose':
                y = _conv2d_transpose(a, lc, w)
            elif cls == 'MaxPooling2D':
                assert lc.get('padding', 'valid') == 'valid'
                y = F.max_pool2d(a, tuple(lc['pool_size']), tuple(lc.get('strides') or lc['pool_size']))
            elif cls == 'AveragePooling2D':
                assert lc.get('padding', 'valid') == 'valid'
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
                y = a.permute(0, 2, 3, 1).reshape(a.shape[0], -1) if a.dim() == 4 else a.reshape(a.shape[0], -1)
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
                assert lc.get('axis', -1) in (-1, 3)
                y = torch.cat(ins, dim=1)
            elif cls == 'Add':
                y = ins[0]
                for t in ins[1:]:
                    y = y + t
            elif cls == 'BatchNormalization':
                y = _batchnorm(a, lc, w)
            elif cls in ('Dropout', 'SpatialDropout2D', 'GaussianNoise', 'GaussianDropout', 'AlphaDropout'):
                y = a
            elif cls == 'Activation':
                y = _act(lc['activation'], a)
            elif cls == 'ReLU':
                assert lc.get('max_value') is None and not lc.get('threshold') and not lc.get('negative_slope')
                y = F.relu(a)
            elif cls == 'LeakyReLU':
                y = F.leaky_relu(a, float(lc.get('alpha', 0.3)))
            elif cls == 'Softmax':
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
            vals[name] = y
            prev = name
        if seq:
            out = vals[prev]
        else:
            out = vals[cfg['output_layers'][0][0]]
    if out.dim() == 2:
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

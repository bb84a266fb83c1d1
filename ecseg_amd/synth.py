"""Synthetic workloads (no network, no datasets): the canonical U-Net written as a Keras ``model_config`` with
seeded weights, and seeded DAPI / FISH / label images of the shapes BASELINE.json names."""
import numpy as np


def unet_config(base=64, depth=4, in_ch=1, n_classes=4, up='transpose', batchnorm=False, name='metaseg_synth'):
    """Keras-2 functional ``model_config`` of the classic U-Net: (conv3-relu x2, pool) x depth, bottleneck,
    (2x2 up-conv, concat skip, conv3-relu x2) x depth, 1x1 softmax head.  ``up='transpose'`` uses Conv2DTranspose
    2x2/s2 (``'transpose3'`` / ``'transpose4'``: 3x3 / 4x4 kernels at stride 2), ``up='upsample'`` uses UpSampling2D +
    Conv2D 2x2 'same' (all occur in public Keras U-Nets)."""
    layers = []
    counter = {}

    def nm(kind):
        k = counter.get(kind, 0)
        counter[kind] = k + 1
        return kind if k == 0 else '%s_%d' % (kind, k)

    def L(cls, kind, inbound, **cfg):
        name_ = nm(kind)
        cfg = dict(cfg, name=name_)
        layers.append({'class_name': cls, 'name': name_, 'config': cfg,
                       'inbound_nodes': [[[i, 0, 0, {}] for i in inbound]] if inbound else []})
        return name_

    def conv(x, filters, k=3, act='relu'):
        y = L('Conv2D', 'conv2d', [x], filters=filters, kernel_size=[k, k], strides=[1, 1], padding='same',
              data_format='channels_last', dilation_rate=[1, 1], groups=1,
              activation='linear' if batchnorm and act == 'relu' else act, use_bias=True, trainable=True, dtype='float32')
        if batchnorm and act == 'relu':
            y = L('BatchNormalization', 'batch_normalization', [y], axis=[3], momentum=0.99, epsilon=1e-3, center=True,
                  scale=True)
            y = L('Activation', 'activation', [y], activation='relu')
        return y

    x = L('InputLayer', 'input', [], batch_input_shape=[None, 256, 256, in_ch], dtype='float32', sparse=False, ragged=False)
    skips = []
    f = base
    for _ in range(depth):
        x = conv(conv(x, f), f)
        skips.append(x)
        x = L('MaxPooling2D', 'max_pooling2d', [x], pool_size=[2, 2], padding='valid', strides=[2, 2],
              data_format='channels_last')
        f *= 2
    x = conv(conv(x, f), f)
    for d in range(depth):
        f //= 2
        if up.startswith('transpose'):                       # 'transpose' 2x2, 'transpose3' 3x3 (NuSeT: src/model_layers/models.py:78-80), 'transpose4' 4x4
            kt = int(up[len('transpose'):] or 2)
            u = L('Conv2DTranspose', 'conv2d_transpose', [x], filters=f, kernel_size=[kt, kt], strides=[2, 2], padding='same',
                  data_format='channels_last', dilation_rate=[1, 1], activation='linear', use_bias=True, output_padding=None)
        else:
            u = L('UpSampling2D', 'up_sampling2d', [x], size=[2, 2], data_format='channels_last', interpolation='nearest')
            u = conv(u, f, k=2)
        x = L('Concatenate', 'concatenate', [skips[depth - 1 - d], u], axis=3)
        x = conv(conv(x, f), f)
    out = conv(x, n_classes, k=1, act='softmax')
    return {'class_name': 'Functional',
            'config': {'name': name, 'layers': layers, 'input_layers': [[layers[0]['name'], 0, 0]],
                       'output_layers': [[out, 0, 0]]}}


def nuset_unet_config(base=64, n_classes=2, in_ch=1, hw=256):
    """The shape of NuSeT's U-Net (reference src/model_layers/models.py:5-136, a TF1-layers graph; SURVEY 8(f)4: "the same conv
    kernels can serve ... NuSeT's U-Net") written as a Keras functional ``model_config``: four (conv3-relu x2, 2x2 pool) levels of
    base .. 8 x base channels, a 16 x base bottleneck, four 3x3 / stride-2 ``Conv2DTranspose`` + ReLU up-samplers of which the
    FIRST is not concatenated with its skip (models.py:78-94) and the other three are, and a final 3x3 convolution to
    ``n_classes`` feature maps without bias or activation (models.py:134).  Only the architecture is taken from the reference -
    weights are seeded like every other synthetic model here."""
    layers = []

    def L(cls, name, inbound, **cfg):
        layers.append({'class_name': cls, 'name': name, 'config': dict(cfg, name=name),
                       'inbound_nodes': [[[i, 0, 0, {}] for i in inbound]] if inbound else []})
        return name

    def conv(name, x, filters, act='relu', bias=True):
        return L('Conv2D', name, [x], filters=filters, kernel_size=[3, 3], strides=[1, 1], padding='same', activation=act,
                 use_bias=bias, dilation_rate=[1, 1], groups=1)

    x = L('InputLayer', 'input', [], batch_input_shape=[None, hw, hw, in_ch], dtype='float32')
    skips = []
    for lv in range(4):
        f = base << lv
        x = conv('conv%d-2' % (lv + 1), conv('conv%d-1' % (lv + 1), x, f), f)
        skips.append(x)
        x = L('MaxPooling2D', 'pool%d' % (lv + 1), [x], pool_size=[2, 2], strides=[2, 2], padding='valid')
    x = conv('conv5-2', conv('conv5-1', x, base << 4), base << 4)
    for lv in (3, 2, 1, 0):
        f = base << lv
        x = L('Conv2DTranspose', 'up%d' % (lv + 1), [x], filters=f, kernel_size=[3, 3], strides=[2, 2], padding='same',
              activation='relu', use_bias=True, dilation_rate=[1, 1], output_padding=None)
        if lv != 3:
            x = L('Concatenate', 'cat%d' % (lv + 1), [skips[lv], x], axis=3)
        x = conv('conv%d-4' % (lv + 1), conv('conv%d-3' % (lv + 1), x, f), f)
    out = conv('final', x, n_classes, act='linear', bias=False)
    return {'class_name': 'Functional', 'config': {'name': 'nuset_unet_synth', 'layers': layers,
                                                   'input_layers': [['input', 0, 0]], 'output_layers': [[out, 0, 0]]}}


def conv_stack_config(n_convs=3, ch=64, n_classes=4, in_ch=1):
    """Plain stack: input -> (conv3-relu) x n_convs -> 1x1 softmax head (no pooling, no skips)."""
    def L(cls, name, inb, **c):
        return {'class_name': cls, 'name': name, 'config': dict(c, name=name),
                'inbound_nodes': [[[i, 0, 0, {}] for i in inb]] if inb else []}
    layers = [L('InputLayer', 'in', [], batch_input_shape=[None, 256, 256, in_ch])]
    prev = 'in'
    for k in range(n_convs):
        layers.append(L('Conv2D', 'c%d' % k, [prev], filters=ch, kernel_size=[3, 3], strides=[1, 1], padding='same',
                        activation='relu', use_bias=True))
        prev = 'c%d' % k
    layers.append(L('Conv2D', 'head', [prev], filters=n_classes, kernel_size=[1, 1], strides=[1, 1], padding='same',
                    activation='softmax', use_bias=True))
    return {'class_name': 'Functional', 'config': {'name': 'stack', 'layers': layers, 'input_layers': [['in', 0, 0]],
                                                   'output_layers': [['head', 0, 0]]}}


def classifier_config(kind='interseg'):
    """Keras-2 functional ``model_config`` of a small per-nucleus classifier in the two call shapes of the reference's
    interSeg step: ``'interseg'``: input (256, 256) uint8 without a channel axis -> 3-way softmax (src/interseg.py:155,
    ``ecseg_i_model.predict(p[..., 0])``); ``'ecseg_c'``: input (256, 256, 3) float32 in [0, 1] (preprocess_ecseg_c,
    src/utils.py:166-173) -> one sigmoid unit (src/interseg.py:168-170).  The real architectures live only in
    interseg_models/*.h5 (not distributable); these cover the layer vocabulary such Keras classifiers are written in."""
    layers = []

    def L(cls, name, inb, **c):
        layers.append({'class_name': cls, 'name': name, 'config': dict(c, name=name),
                       'inbound_nodes': [[[i, 0, 0, {}] for i in inb]] if inb else []})
        return name

    def conv(name, x, f, k, s=1, pad='same', act='relu'):
        return L('Conv2D', name, [x], filters=f, kernel_size=[k, k], strides=[s, s], padding=pad, data_format='channels_last',
                 dilation_rate=[1, 1], groups=1, activation=act, use_bias=True)

    if kind == 'interseg':
        x = L('InputLayer', 'input_1', [], batch_input_shape=[None, 256, 256], dtype='float32')
        x = L('Reshape', 'reshape', [x], target_shape=[256, 256, 1])
        x = L('Rescaling', 'rescaling', [x], scale=1.0 / 255.0, offset=0.0)
        x = conv('conv_a', x, 16, 3, s=2)
        x = conv('conv_b', x, 32, 3)
        x = L('MaxPooling2D', 'pool_a', [x], pool_size=[2, 2], strides=[2, 2], padding='valid')
        x = conv('conv_c', x, 64, 3, act='linear')
        x = L('BatchNormalization', 'bn_c', [x], axis=[3], momentum=0.99, epsilon=1e-3, center=True, scale=True)
        x = L('Activation', 'act_c', [x], activation='relu')
        x = L('MaxPooling2D', 'pool_b', [x], pool_size=[2, 2], strides=[2, 2], padding='valid')
        x = conv('conv_d', x, 128, 3)
        x = L('GlobalAveragePooling2D', 'gap', [x], data_format='channels_last', keepdims=False)
        x = L('Dense', 'dense_a', [x], units=64, activation='relu', use_bias=True)
        x = L('Dropout', 'dropout', [x], rate=0.3)
        out = L('Dense', 'dense_out', [x], units=3, activation='softmax', use_bias=True)
    elif kind == 'ecseg_c':
        x = L('InputLayer', 'input_1', [], batch_input_shape=[None, 256, 256, 3], dtype='float32')
        x = conv('conv_a', x, 16, 5, s=2)
        x = L('AveragePooling2D', 'avg_a', [x], pool_size=[2, 2], strides=[2, 2], padding='valid')
        x = conv('conv_b', x, 32, 3, pad='valid')
        x = L('MaxPooling2D', 'pool_b', [x], pool_size=[2, 2], strides=[2, 2], padding='valid')
        x = conv('conv_c', x, 64, 3)
        x = L('MaxPooling2D', 'pool_c', [x], pool_size=[4, 4], strides=[4, 4], padding='valid')
        x = L('Flatten', 'flatten', [x], data_format='channels_last')
        x = L('Dense', 'dense_a', [x], units=32, activation='relu', use_bias=True)
        out = L('Dense', 'dense_out', [x], units=1, activation='sigmoid', use_bias=True)
    else:
        raise ValueError(kind)
    return {'class_name': 'Functional', 'config': {'name': kind + '_synth', 'layers': layers,
                                                   'input_layers': [['input_1', 0, 0]], 'output_layers': [[out, 0, 0]]}}


def classifier_weights(config, seed=0):
    """Seeded He-normal weights for ``classifier_config`` (Conv2D / Dense / BatchNormalization)."""
    rng = np.random.default_rng(seed)
    weights = {}
    shape = {}
    for Ld in config['config']['layers']:
        cls, lc, name = Ld['class_name'], Ld['config'], Ld['config']['name']
        if cls == 'InputLayer':
            bis = lc['batch_input_shape']
            shape[name] = (bis[1], bis[2], bis[3] if len(bis) > 3 else 1)
            continue
        h, w, c = shape[Ld['inbound_nodes'][0][0][0]]
        if cls == 'Conv2D':
            k, st, f = lc['kernel_size'][0], lc['strides'][0], lc['filters']
            weights[name] = [(rng.normal(size=(k, k, c, f)) * np.sqrt(2.0 / (k * k * c))).astype(np.float32),
                             (rng.normal(size=f) * 0.05).astype(np.float32)]
            shape[name] = (-(-h // st), -(-w // st), f) if lc['padding'] == 'same' else ((h - k) // st + 1, (w - k) // st + 1, f)
        elif cls in ('MaxPooling2D', 'AveragePooling2D'):
            k, st = lc['pool_size'][0], lc['strides'][0]
            shape[name] = ((h - k) // st + 1, (w - k) // st + 1, c)
        elif cls == 'BatchNormalization':
            weights[name] = [rng.uniform(0.8, 1.2, c).astype(np.float32), (rng.normal(size=c) * 0.05).astype(np.float32),
                             (rng.normal(size=c) * 0.05).astype(np.float32), rng.uniform(0.8, 1.2, c).astype(np.float32)]
            shape[name] = (h, w, c)
        elif cls in ('GlobalAveragePooling2D', 'GlobalMaxPooling2D'):
            shape[name] = (1, 1, c)
        elif cls == 'Flatten':
            shape[name] = (1, 1, h * w * c)
        elif cls == 'Reshape':
            ts = lc['target_shape']
            shape[name] = tuple(ts) if len(ts) == 3 else (1, 1, ts[0])
        elif cls == 'Dense':
            u = lc['units']
            weights[name] = [(rng.normal(size=(c, u)) * np.sqrt(2.0 / c)).astype(np.float32), (rng.normal(size=u) * 0.05).astype(np.float32)]
            shape[name] = (h, w, u)
        else:
            shape[name] = (h, w, c)
    return weights


def unet_weights(config, seed=0, input_scale=1.0 / 255.0, head_gain=6.0, smooth=False):
    """Seeded He-normal kernels.  The first convolution is scaled by ``input_scale`` because the reference feeds raw
    0..255 pixel values (no normalisation anywhere in src/utils.py:109-120); the head is scaled up so that the
    softmax is decisive rather than uniform.

    ``smooth=True``: a SMOOTH-OUTPUT model from the same seed without any training - every 3x3 kernel is a seeded
    He-normal channel-mixing matrix times the binomial low-pass stencil [1 2 1] x [1 2 1] / 16 (plus 3 % of the ordinary
    random kernel), every 2x2 up-convolution the same mixing matrix on all four taps: the network is a cascade of blurs
    and channel mixes, its label maps are blobs with smooth boundaries like a trained model's instead of the speckle
    of ``smooth=False``, and it still executes every multiply of the architecture with generic float32 values.  Used
    where a realistic tie density matters (label-mismatch measurements, tests) and 124 MB of fitted weights cannot travel."""
    rng = np.random.default_rng(seed)
    stencil = np.outer([1.0, 2.0, 1.0], [1.0, 2.0, 1.0]) / 16.0
    weights = {}
    first = True
    layers = config['config']['layers']
    cin = {}
    for Ld in layers:
        cls, lc, name = Ld['class_name'], Ld['config'], Ld['config']['name']
        if cls == 'InputLayer':
            cin[name] = lc['batch_input_shape'][3]
            continue
        srcs = [r[0] for r in Ld['inbound_nodes'][0]]
        c_in = sum(cin[s] for s in srcs) if cls == 'Concatenate' else cin[srcs[0]]
        if cls == 'Conv2D':
            kh, kw = lc['kernel_size']
            co = lc['filters']
            k = rng.normal(size=(kh, kw, c_in, co)) * np.sqrt(2.0 / (kh * kw * c_in))
            if smooth and (kh, kw) == (3, 3):
                k = 0.03 * k + (rng.normal(size=(c_in, co)) * np.sqrt(2.0 / c_in))[None, None] * stencil[:, :, None, None]
            if first:
                k *= input_scale
                first = False
            if lc['activation'] == 'softmax':
                k *= head_gain
            weights[name] = [k.astype(np.float32), (rng.normal(size=co) * 0.05).astype(np.float32)]
            cin[name] = co
        elif cls == 'Conv2DTranspose':
            kh, kw = lc['kernel_size']
            co = lc['filters']
            k = rng.normal(size=(kh, kw, co, c_in)) * np.sqrt(1.0 / c_in)
            if smooth:
                k = 0.03 * k + (rng.normal(size=(co, c_in)) * np.sqrt(1.0 / c_in))[None, None]
            weights[name] = [k.astype(np.float32), (rng.normal(size=co) * 0.05).astype(np.float32)]
            cin[name] = co
        elif cls == 'BatchNormalization':
            weights[name] = [rng.uniform(0.8, 1.2, c_in).astype(np.float32), (rng.normal(size=c_in) * 0.05).astype(np.float32),
                             (rng.normal(size=c_in) * 0.05).astype(np.float32), rng.uniform(0.8, 1.2, c_in).astype(np.float32)]
            cin[name] = c_in
        else:
            cin[name] = c_in
    return weights


def _blur(img, sigma):
    r = int(3 * sigma + 0.5)
    x = np.arange(-r, r + 1)
    k = np.exp(-0.5 * (x / sigma) ** 2)
    k /= k.sum()
    pad = np.pad(img, ((r, r), (r, r)), mode='edge')
    tmp = sum(k[i] * pad[i:i + img.shape[0], :] for i in range(2 * r + 1))
    return sum(k[i] * tmp[:, i:i + img.shape[1]] for i in range(2 * r + 1))


def dapi_image(idx, H=1040, W=1392, rgb=False, with_labels=False):
    """Seeded synthetic DAPI metaphase image (uint8): dim noisy background, a few nuclei discs, a cluster of
    chromosome ellipses, ecDNA dots; blurred with sigma 1.5.  ``rgb=True`` puts it in channel 2 and adds red /
    green FISH spot fields in channels 0 / 1.  ``with_labels=True`` also returns the scene's class map
    (uint8: 0 background, 1 nucleus, 2 chromosome, 3 ecDNA; later objects cover earlier ones) - the pixel image is the
    same either way."""
    rng = np.random.default_rng(1234 + idx)
    img = np.clip(rng.normal(6, 4, size=(H, W)), 0, 255)
    lab = np.zeros((H, W), np.uint8)
    yy, xx = np.ogrid[:H, :W]
    for _ in range(int(rng.integers(2, 5))):
        r = rng.integers(60, 121)
        cy, cx = rng.integers(0, H), rng.integers(0, W)
        m = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
        img[m] = np.clip(rng.normal(160, 25), 60, 255)
        lab[m] = 1
    cy0, cx0 = rng.integers(H // 4, 3 * H // 4), rng.integers(W // 4, 3 * W // 4)
    for _ in range(int(rng.integers(40, 71))):
        a, b = rng.uniform(8, 25), rng.uniform(4, 9)
        cy, cx = cy0 + rng.integers(-250, 251), cx0 + rng.integers(-250, 251)
        th = rng.uniform(0, np.pi)
        y0, y1 = max(int(cy - 30), 0), min(int(cy + 30), H)
        x0, x1 = max(int(cx - 30), 0), min(int(cx + 30), W)
        if y0 >= y1 or x0 >= x1:
            continue
        sy, sx = np.ogrid[y0:y1, x0:x1]
        u = (sx - cx) * np.cos(th) + (sy - cy) * np.sin(th)
        v = -(sx - cx) * np.sin(th) + (sy - cy) * np.cos(th)
        m = (u / a) ** 2 + (v / b) ** 2 <= 1
        img[y0:y1, x0:x1][m] = np.clip(rng.normal(200, 30), 80, 255)
        lab[y0:y1, x0:x1][m] = 2
    for _ in range(int(rng.integers(50, 201))):
        r = rng.integers(2, 5)
        cy, cx = rng.integers(r, H - r), rng.integers(r, W - r)
        sy, sx = np.ogrid[cy - r:cy + r + 1, cx - r:cx + r + 1]
        m = (sy - cy) ** 2 + (sx - cx) ** 2 <= r * r
        img[cy - r:cy + r + 1, cx - r:cx + r + 1][m] = np.clip(rng.normal(180, 40), 60, 255)
        lab[cy - r:cy + r + 1, cx - r:cx + r + 1][m] = 3
    gray = np.clip(np.rint(_blur(img, 1.5)), 0, 255).astype(np.uint8)
    if not rgb:
        return (gray, lab) if with_labels else gray
    out = np.zeros((H, W, 3), np.uint8)
    out[..., 2] = gray
    for ch in (0, 1):
        f = np.clip(rng.normal(20, 8, size=(H, W)), 0, 255)
        for _ in range(int(rng.integers(50, 151))):
            r = rng.integers(1, 4)
            cy, cx = rng.integers(r, H - r), rng.integers(r, W - r)
            sy, sx = np.ogrid[cy - r:cy + r + 1, cx - r:cx + r + 1]
            m = (sy - cy) ** 2 + (sx - cx) ** 2 <= r * r
            f[cy - r:cy + r + 1, cx - r:cx + r + 1][m] = rng.integers(120, 256)
        out[..., ch] = f.astype(np.uint8)
    return (out, lab) if with_labels else out


def label_map(idx, H=1040, W=1392, salt=0.002):
    """Seeded synthetic label image (uint8 0..3) from the same kind of scene, independent of any network: class by
    blob type plus salt noise.  Used for post-processing / CCL measurements."""
    rng = np.random.default_rng(4321 + idx)
    lab = np.zeros((H, W), np.uint8)
    yy, xx = np.ogrid[:H, :W]
    for _ in range(int(rng.integers(2, 5))):
        r = rng.integers(60, 121)
        cy, cx = rng.integers(0, H), rng.integers(0, W)
        d2 = (yy - cy) ** 2 + (xx - cx) ** 2
        lab[d2 <= r * r] = 1
        if rng.random() < 0.5:
            lab[d2 <= (r // 4) ** 2] = 0
    cy0, cx0 = rng.integers(H // 4, 3 * H // 4), rng.integers(W // 4, 3 * W // 4)
    for _ in range(int(rng.integers(40, 71))):
        a, b = rng.uniform(8, 25), rng.uniform(4, 9)
        cy, cx = cy0 + rng.integers(-250, 251), cx0 + rng.integers(-250, 251)
        th = rng.uniform(0, np.pi)
        y0, y1 = max(int(cy - 30), 0), min(int(cy + 30), H)
        x0, x1 = max(int(cx - 30), 0), min(int(cx + 30), W)
        if y0 >= y1 or x0 >= x1:
            continue
        sy, sx = np.ogrid[y0:y1, x0:x1]
        u = (sx - cx) * np.cos(th) + (sy - cy) * np.sin(th)
        v = -(sx - cx) * np.sin(th) + (sy - cy) * np.cos(th)
        lab[y0:y1, x0:x1][(u / a) ** 2 + (v / b) ** 2 <= 1] = 2
    for _ in range(int(rng.integers(50, 201))):
        r = rng.integers(2, 5)
        cy, cx = rng.integers(r, H - r), rng.integers(r, W - r)
        sy, sx = np.ogrid[cy - r:cy + r + 1, cx - r:cx + r + 1]
        lab[cy - r:cy + r + 1, cx - r:cx + r + 1][(sy - cy) ** 2 + (sx - cx) ** 2 <= r * r] = 3
    if salt > 0:
        m = rng.random((H, W)) < salt
        lab[m] = rng.integers(0, 4, size=int(m.sum()))
    return lab


def mobilenet_classifier(seed=0, hw=96, n_classes=3, width=16):
    """-> (model_config, weights): a MobileNet-style transfer-learning classifier as Keras 2 saves one - the vocabulary an
    ``interseg_models/*`` file may hold beyond plain convolutions (src/interseg.py:96-98 loads whatever the file contains):
    a NESTED Functional backbone used as one layer (strided stem, depthwise-separable blocks with BatchNormalization and
    ReLU(max_value=6), a squeeze-and-excite gate - GlobalAveragePooling2D(keepdims) -> 1x1 convolutions -> Multiply -, a
    residual Add, a dilated 3x3 convolution, a grouped convolution, a SeparableConv2D, PReLU), then a head of global pooling,
    LayerNormalization and Dense.  ``weights``: {outer layer: [arrays]} with the backbone's as {inner layer: [arrays]}."""
    rng = np.random.default_rng(seed)
    inner, wi = [], {}

    def L(dst, cls, name, inb, **c):
        dst.append({'class_name': cls, 'name': name, 'config': dict(c, name=name),
                    'inbound_nodes': [[[i, 0, 0, {}] for i in inb]] if inb else []})
        return name

    def he(*s):
        fan = int(np.prod(s[:-1])) if len(s) > 1 else s[0]
        return (rng.normal(size=s) * np.sqrt(2.0 / fan)).astype(np.float32)

    def bn(name, x, c):
        wi[name] = [rng.uniform(0.8, 1.2, c).astype(np.float32), (rng.normal(size=c) * 0.05).astype(np.float32),
                    (rng.normal(size=c) * 0.05).astype(np.float32), rng.uniform(0.8, 1.2, c).astype(np.float32)]
        return L(inner, 'BatchNormalization', name, [x], axis=[3], momentum=0.99, epsilon=1e-3, center=True, scale=True)

    def relu6(name, x):
        return L(inner, 'ReLU', name, [x], max_value=6.0, negative_slope=0.0, threshold=0.0)

    def conv(name, x, cin, f, k=1, s=1, act='linear', bias=False, dil=1, groups=1, pad='same'):
        wi[name] = [he(k, k, cin // groups, f)] + ([(rng.normal(size=f) * 0.05).astype(np.float32)] if bias else [])
        return L(inner, 'Conv2D', name, [x], filters=f, kernel_size=[k, k], strides=[s, s], padding=pad, data_format='channels_last',
                 dilation_rate=[dil, dil], groups=groups, activation=act, use_bias=bias)

    def dw(name, x, c, k=3, s=1, dil=1):
        wi[name] = [(rng.normal(size=(k, k, c, 1)) * np.sqrt(2.0 / (k * k))).astype(np.float32)]
        return L(inner, 'DepthwiseConv2D', name, [x], kernel_size=[k, k], strides=[s, s], padding='same', data_format='channels_last',
                 dilation_rate=[dil, dil], depth_multiplier=1, activation='linear', use_bias=False)

    w1, w2 = width, 2 * width
    x = L(inner, 'InputLayer', 'backbone_in', [], batch_input_shape=[None, hw, hw, 3], dtype='float32')
    x = relu6('stem_relu', bn('stem_bn', conv('stem', x, 3, w1, k=3, s=2), w1))
    # depthwise-separable block 1 (stride 1, residual)
    y = relu6('b1_dw_relu', bn('b1_dw_bn', dw('b1_dw', x, w1), w1))
    y = bn('b1_pw_bn', conv('b1_pw', y, w1, w1), w1)
    x = L(inner, 'Add', 'b1_add', [x, y])
    # block 2: expand, depthwise stride 2, squeeze-and-excite, project
    y = relu6('b2_exp_relu', bn('b2_exp_bn', conv('b2_exp', x, w1, w2), w2))
    y = relu6('b2_dw_relu', bn('b2_dw_bn', dw('b2_dw', y, w2, k=3, s=2), w2))
    se = L(inner, 'GlobalAveragePooling2D', 'b2_se_gap', [y], data_format='channels_last', keepdims=True)
    se = conv('b2_se_reduce', se, w2, w2 // 4, act='relu', bias=True)
    se = conv('b2_se_expand', se, w2 // 4, w2, act='hard_sigmoid', bias=True)
    y = L(inner, 'Multiply', 'b2_se_mul', [y, se])
    x = bn('b2_pw_bn', conv('b2_pw', y, w2, w2), w2)
    # dilated context, grouped convolution, separable convolution, PReLU
    y = conv('ctx_dil', x, w2, w2, k=3, dil=2, act='relu', bias=True)
    y = conv('ctx_grp', y, w2, w2, k=3, groups=4, act='linear', bias=True)
    wi['ctx_prelu'] = [rng.uniform(0.05, 0.3, (1, 1, w2)).astype(np.float32)]
    y = L(inner, 'PReLU', 'ctx_prelu', [y], shared_axes=[1, 2])
    wi['ctx_sep'] = [(rng.normal(size=(3, 3, w2, 1)) * np.sqrt(2.0 / 9)).astype(np.float32), he(1, 1, w2, w2),
                     (rng.normal(size=w2) * 0.05).astype(np.float32)]
    y = L(inner, 'SeparableConv2D', 'ctx_sep', [y], filters=w2, kernel_size=[3, 3], strides=[1, 1], padding='same',
          data_format='channels_last', dilation_rate=[1, 1], depth_multiplier=1, activation='swish', use_bias=True)
    x = L(inner, 'Add', 'ctx_add', [x, y])
    x = L(inner, 'MaxPooling2D', 'ctx_pool', [x], pool_size=[3, 3], strides=[2, 2], padding='same', data_format='channels_last')
    backbone = {'class_name': 'Functional', 'name': 'backbone',
                'config': {'name': 'backbone', 'layers': inner, 'input_layers': [['backbone_in', 0, 0]], 'output_layers': [[x, 0, 0]]}}
    outer, wo = [], {}
    i = L(outer, 'InputLayer', 'input_1', [], batch_input_shape=[None, hw, hw, 3], dtype='float32')
    r = L(outer, 'Rescaling', 'rescaling', [i], scale=1.0 / 127.5, offset=-1.0)
    outer.append(dict(backbone, inbound_nodes=[[[r, 0, 0, {}]]]))
    g = L(outer, 'GlobalAveragePooling2D', 'gap', ['backbone'], data_format='channels_last', keepdims=False)
    wo['head_ln'] = [rng.uniform(0.8, 1.2, w2).astype(np.float32), (rng.normal(size=w2) * 0.05).astype(np.float32)]
    g = L(outer, 'LayerNormalization', 'head_ln', [g], axis=[1], epsilon=1e-3, center=True, scale=True)
    wo['dense_out'] = [(rng.normal(size=(w2, n_classes)) * np.sqrt(2.0 / w2) * 3).astype(np.float32), (rng.normal(size=n_classes) * 0.05).astype(np.float32)]
    out = L(outer, 'Dense', 'dense_out', [g], units=n_classes, activation='softmax', use_bias=True)
    wo['backbone'] = wi
    cfg = {'class_name': 'Functional', 'config': {'name': 'mobilenet_synth', 'layers': outer, 'input_layers': [['input_1', 0, 0]],
                                                  'output_layers': [[out, 0, 0]]}}
    return cfg, wo

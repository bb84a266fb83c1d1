"""Every Keras layer type the U-Net oracle (oracle/unet.py ``forward``, torch CPU) evaluates is checked here against
(a) known answers written out by hand from the layers' published definitions and (b) an independent numpy restatement
(``*_numpy`` in oracle/unet.py).  TensorFlow itself is not available (PARITY UNPINNED vs TF 2.8); this pins the oracle
on two implementations that share no code."""
import numpy as np
import pytest

from oracle import unet


def _model(layer_cls, shape, **cfg):
    cfg = dict(cfg, name='L')
    return {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None] + list(shape)},
         'inbound_nodes': []},
        {'class_name': layer_cls, 'name': 'L', 'config': cfg, 'inbound_nodes': [[['in', 0, 0, {}]]]}],
        'input_layers': [['in', 0, 0]], 'output_layers': [['L', 0, 0]]}}


def test_conv2d_same_padding_even_kernel_known_answer():
    """TF 'same' with a 2x2 kernel at stride 1 pads ONE row/column, at the bottom/right only."""
    x = np.array([[1, 2], [3, 4]], np.float32).reshape(1, 2, 2, 1)
    k = np.array([[1, 10], [100, 1000]], np.float32).reshape(2, 2, 1, 1)
    cfg = _model('Conv2D', (2, 2, 1), filters=1, kernel_size=[2, 2], strides=[1, 1], padding='same', activation='linear',
                 use_bias=False)
    got = unet.forward(cfg, {'L': [k]}, x)[0, :, :, 0]
    # out[i,j] = x[i,j] + 10 x[i,j+1] + 100 x[i+1,j] + 1000 x[i+1,j+1], zeros beyond the bottom/right edge
    want = np.array([[1 + 20 + 300 + 4000, 2 + 400], [3 + 40, 4]], np.float32)
    assert np.array_equal(got, want)
    assert np.array_equal(unet.conv_numpy(x, k, np.zeros(1))[0, :, :, 0], want)


def test_conv2d_3x3_same_known_answer():
    x = np.arange(1, 10, dtype=np.float32).reshape(1, 3, 3, 1)
    k = np.zeros((3, 3, 1, 1), np.float32)
    k[0, 0] = 1; k[2, 2] = 2; k[1, 1] = 3              # top-left, bottom-right, centre taps (cross-correlation)
    cfg = _model('Conv2D', (3, 3, 1), filters=1, kernel_size=[3, 3], strides=[1, 1], padding='same', activation='linear',
                 use_bias=True)
    got = unet.forward(cfg, {'L': [k, np.array([0.5], np.float32)]}, x)[0, :, :, 0]
    X = np.pad(x[0, :, :, 0], 1)
    want = X[:-2, :-2] * 1 + X[2:, 2:] * 2 + X[1:-1, 1:-1] * 3 + 0.5
    assert np.array_equal(got, want)
    assert got[1, 1] == 1 * 1 + 9 * 2 + 5 * 3 + 0.5


def test_conv2d_transpose_2x2_stride2_known_answer():
    """kernel (kh, kw, out, in): out[2i + a, 2j + b, o] = sum_c in[i, j, c] * w[a, b, o, c] (+ bias): no overlap."""
    x = np.array([[1, 2], [3, 4]], np.float32).reshape(1, 2, 2, 1)
    k = np.array([[1, 10], [100, 1000]], np.float32).reshape(2, 2, 1, 1)
    cfg = _model('Conv2DTranspose', (2, 2, 1), filters=1, kernel_size=[2, 2], strides=[2, 2], padding='same',
                 activation='linear', use_bias=True, output_padding=None)
    got = unet.forward(cfg, {'L': [k, np.array([0.25], np.float32)]}, x)[0, :, :, 0]
    want = np.kron(x[0, :, :, 0], k[:, :, 0, 0]) + 0.25
    assert np.array_equal(got, want)
    assert np.array_equal(unet.conv_transpose_numpy(x, k, [0.25], 2)[0, :, :, 0], want)


def test_conv2d_transpose_3x3_stride2_same_and_valid_known_answer():
    """1-D-like case by hand: in = [1, 2], w = [1, 10, 100], stride 2: full = [1, 10, 100 + 2, 20, 200]; 'valid' keeps all
    5, 'same' (output 4) drops max(k - s, 0) // 2 = 0 in front and the last one."""
    x = np.array([1, 2], np.float32).reshape(1, 1, 2, 1)
    k = np.zeros((3, 3, 1, 1), np.float32)
    k[0, :, 0, 0] = [1, 10, 100]                          # only the first kernel row: output row 0 carries the 1-D case
    for pad, want in (('valid', [1, 10, 102, 20, 200]), ('same', [1, 10, 102, 20])):
        cfg = _model('Conv2DTranspose', (1, 2, 1), filters=1, kernel_size=[3, 3], strides=[2, 2], padding=pad,
                     activation='linear', use_bias=False, output_padding=None)
        got = unet.forward(cfg, {'L': [k]}, x)
        assert got.shape == ((1, 3, 5, 1) if pad == 'valid' else (1, 2, 4, 1))
        assert np.array_equal(got[0, 0, :, 0], np.array(want, np.float32))
        assert np.array_equal(unet.conv_transpose_numpy(x, k, None, 2, pad), got)


def test_conv2d_transpose_4x4_stride2_same_crops_one_in_front():
    x = np.array([1, 2, 3], np.float32).reshape(1, 1, 3, 1)
    k = np.zeros((4, 4, 1, 1), np.float32)
    k[1, :, 0, 0] = [1, 10, 100, 1000]                    # kernel row 1 lands on output row 0 after the 1-row crop
    cfg = _model('Conv2DTranspose', (1, 3, 1), filters=1, kernel_size=[4, 4], strides=[2, 2], padding='same',
                 activation='linear', use_bias=False, output_padding=None)
    got = unet.forward(cfg, {'L': [k]}, x)
    full = np.array([1, 10, 100 + 2, 1000 + 20, 200 + 3, 2000 + 30, 300, 3000], np.float32)
    assert np.array_equal(got[0, 0, :, 0], full[1:7])
    assert np.array_equal(unet.conv_transpose_numpy(x, k, None, 2, 'same'), got)


def test_upsampling_known_answers():
    x = np.array([1, 2], np.float32).reshape(1, 1, 2, 1)
    near = unet.forward(_model('UpSampling2D', (1, 2, 1), size=[2, 2], interpolation='nearest'), {}, x)
    assert np.array_equal(near[0, :, :, 0], [[1, 1, 2, 2], [1, 1, 2, 2]])
    bil = unet.forward(_model('UpSampling2D', (1, 2, 1), size=[2, 2], interpolation='bilinear'), {}, x)
    # half-pixel centres: sources -0.25, 0.25, 0.75, 1.25 -> clamp, 0.75/0.25, 0.25/0.75, clamp
    assert np.allclose(bil[0, 0, :, 0], [1, 1.25, 1.75, 2], atol=1e-7)
    x3 = np.array([0, 4, 8], np.float32).reshape(1, 3, 1, 1)
    bil3 = unet.forward(_model('UpSampling2D', (3, 1, 1), size=[2, 2], interpolation='bilinear'), {}, x3)
    assert np.allclose(bil3[0, :, 0, 0], [0, 1, 3, 5, 7, 8], atol=1e-6)


def test_batchnorm_maxpool_softmax_known_answers():
    x = np.array([[2.0, -1.0]], np.float32).reshape(1, 1, 1, 2)
    w = [np.array([2, 3], np.float32), np.array([1, -1], np.float32), np.array([1, 1], np.float32), np.array([3, 8], np.float32)]
    cfg = _model('BatchNormalization', (1, 1, 2), axis=[3], epsilon=1.0, center=True, scale=True)
    got = unet.forward(cfg, {'L': w}, x)[0, 0, 0]
    # gamma (x - mean) / sqrt(var + eps) + beta = 2 * 1 / 2 + 1, 3 * (-2) / 3 - 1
    assert np.allclose(got, [2.0, -3.0], atol=1e-6)
    nocenter = unet.forward(_model('BatchNormalization', (1, 1, 2), axis=[3], epsilon=1.0, center=False, scale=True),
                            {'L': [w[0], w[2], w[3]]}, x)[0, 0, 0]
    assert np.allclose(nocenter, [1.0, -2.0], atol=1e-6)
    p = np.arange(25, dtype=np.float32).reshape(1, 5, 5, 1)
    mp = unet.forward(_model('MaxPooling2D', (5, 5, 1), pool_size=[2, 2], strides=[2, 2], padding='valid'), {}, p)
    assert np.array_equal(mp[0, :, :, 0], [[6, 8], [16, 18]])           # floor: the fifth row / column is dropped
    z = np.log(np.array([1, 2, 3, 4], np.float32)).reshape(1, 1, 1, 4)
    sm = unet.forward(_model('Softmax', (1, 1, 4), axis=-1), {}, z)[0, 0, 0]
    assert np.allclose(sm, [0.1, 0.2, 0.3, 0.4], atol=1e-6)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_numpy_restatements_match_torch_path_on_random_tensors(seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(2, 7, 9, 5)).astype(np.float32)
    for k, s, pad in ((2, 2, 'same'), (3, 2, 'same'), (4, 2, 'same'), (3, 2, 'valid'), (2, 2, 'valid'), (3, 3, 'same')):
        w = rng.normal(size=(k, k, 6, 5)).astype(np.float32)
        b = rng.normal(size=6).astype(np.float32)
        cfg = _model('Conv2DTranspose', (7, 9, 5), filters=6, kernel_size=[k, k], strides=[s, s], padding=pad,
                     activation='linear', use_bias=True, output_padding=None)
        got = unet.forward(cfg, {'L': [w, b]}, x)
        want = unet.conv_transpose_numpy(x, w, b, s, pad)
        assert got.shape == want.shape, (k, s, pad)
        assert np.abs(got - want).max() < 1e-4, (k, s, pad)
    for k in (1, 2, 3, 4, 5):
        w = rng.normal(size=(k, k, 5, 4)).astype(np.float32)
        b = rng.normal(size=4).astype(np.float32)
        for pad in ('same', 'valid'):
            cfg = _model('Conv2D', (7, 9, 5), filters=4, kernel_size=[k, k], strides=[1, 1], padding=pad,
                         activation='linear', use_bias=True)
            assert np.abs(unet.forward(cfg, {'L': [w, b]}, x) - unet.conv_numpy(x, w, b, pad)).max() < 1e-4, (k, pad)
    for interp in ('nearest', 'bilinear'):
        for f in (2, 3):
            got = unet.forward(_model('UpSampling2D', (7, 9, 5), size=[f, f], interpolation=interp), {}, x)
            assert np.abs(got - unet.upsample_numpy(x, f, interp)).max() < 1e-5, (interp, f)
    bn = [rng.uniform(0.5, 2, 5).astype(np.float32), rng.normal(size=5).astype(np.float32),
          rng.normal(size=5).astype(np.float32), rng.uniform(0.5, 2, 5).astype(np.float32)]
    got = unet.forward(_model('BatchNormalization', (7, 9, 5), axis=[3], epsilon=1e-3, center=True, scale=True), {'L': bn}, x)
    assert np.abs(got - unet.batchnorm_numpy(x, *bn, eps=1e-3)).max() < 1e-5
    got = unet.forward(_model('MaxPooling2D', (7, 9, 5), pool_size=[2, 2], strides=[2, 2], padding='valid'), {}, x)
    assert np.array_equal(got, unet.maxpool_numpy(x))
    got = unet.forward(_model('Softmax', (7, 9, 5), axis=-1), {}, x)
    assert np.abs(got - unet.softmax_numpy(x)).max() < 1e-6


def test_whole_unet_numpy_chain_matches_torch_path():
    """The canonical U-Net (depth 2, base 8, both up-sampling styles, with and without BatchNormalization) evaluated layer
    by layer with the numpy restatements only == the torch path."""
    from ecseg_amd import synth
    rng = np.random.default_rng(4)
    x = rng.integers(0, 256, size=(1, 32, 32, 1), dtype=np.uint8)
    for up, bn in (('transpose', False), ('upsample', True)):
        cfg = synth.unet_config(base=8, depth=2, up=up, batchnorm=bn)
        cfg['config']['layers'][0]['config']['batch_input_shape'] = [None, 32, 32, 1]
        w = synth.unet_weights(cfg, seed=2)
        want = unet.forward(cfg, w, x)
        vals = {}
        for L in cfg['config']['layers']:
            cls, lc, name = L['class_name'], L['config'], L['config']['name']
            if cls == 'InputLayer':
                vals[name] = x.astype(np.float32)
                continue
            ins = [vals[r[0]] for r in L['inbound_nodes'][0]]
            a = ins[0]
            if cls == 'Conv2D':
                y = unet.conv_numpy(a, w[name][0], w[name][1], lc['padding'])
                y = np.maximum(y, 0) if lc['activation'] == 'relu' else unet.softmax_numpy(y) if lc['activation'] == 'softmax' else y
            elif cls == 'Conv2DTranspose':
                y = unet.conv_transpose_numpy(a, w[name][0], w[name][1], lc['strides'][0], lc['padding'])
            elif cls == 'MaxPooling2D':
                y = unet.maxpool_numpy(a)
            elif cls == 'UpSampling2D':
                y = unet.upsample_numpy(a, lc['size'][0], lc['interpolation'])
            elif cls == 'Concatenate':
                y = np.concatenate(ins, -1)
            elif cls == 'BatchNormalization':
                y = unet.batchnorm_numpy(a, *w[name], eps=lc['epsilon'])
            elif cls == 'Activation':
                y = np.maximum(a, 0)
            else:
                raise AssertionError(cls)
            vals[name] = y
        got = vals[cfg['config']['output_layers'][0][0]]
        assert np.abs(got - want).max() < 1e-5, (up, bn)


def test_classifier_layers_known_answers_and_numpy_restatements():
    """Strided Conv2D ('same' pads the smaller half in front), AveragePooling2D, global pooling, Flatten (NHWC order),
    Reshape, Dense - the layer types of the interSeg classifiers (src/interseg.py:96-98)."""
    # stride 2, 'same', 3x3 on 4 pixels: total pad = (2 - 1) * 2 + 3 - 4 = 1 -> 0 in front, 1 behind
    x = np.array([1, 2, 3, 4], np.float32).reshape(1, 1, 4, 1)
    k = np.zeros((3, 3, 1, 1), np.float32)
    k[0, :, 0, 0] = [1, 10, 100]                      # with 1 row: pad_top = 0 (total (1-1)*2+3-1 = 2 -> 1 front?)
    cfg = _model('Conv2D', (1, 4, 1), filters=1, kernel_size=[3, 3], strides=[2, 2], padding='same', activation='linear', use_bias=False)
    got = unet.forward(cfg, {'L': [k]}, x)
    assert got.shape == (1, 1, 2, 1)
    # rows: H = 1, total pad 2 -> 1 in front: kernel row 1 sits on the data row, row 0 on padding -> all zero output
    assert np.array_equal(got[0, 0, :, 0], [0, 0])
    k2 = np.zeros((3, 3, 1, 1), np.float32)
    k2[1, :, 0, 0] = [1, 10, 100]
    got = unet.forward(cfg, {'L': [k2]}, x)
    assert np.array_equal(got[0, 0, :, 0], [1 + 20 + 300, 3 + 40 + 0])       # windows [1,2,3] and [3,4,pad]
    assert np.array_equal(unet.conv_numpy(x, k2, np.zeros(1), 'same', 2), got)
    rng = np.random.default_rng(5)
    xr = rng.normal(size=(2, 9, 11, 3)).astype(np.float32)
    for kk, st, pad in ((3, 2, 'same'), (5, 2, 'valid'), (2, 3, 'same'), (4, 2, 'same')):
        w = rng.normal(size=(kk, kk, 3, 4)).astype(np.float32)
        b = rng.normal(size=4).astype(np.float32)
        c = _model('Conv2D', (9, 11, 3), filters=4, kernel_size=[kk, kk], strides=[st, st], padding=pad, activation='linear', use_bias=True)
        assert np.abs(unet.forward(c, {'L': [w, b]}, xr) - unet.conv_numpy(xr, w, b, pad, st)).max() < 1e-4, (kk, st, pad)
    p = np.arange(16, dtype=np.float32).reshape(1, 4, 4, 1)
    ap = unet.forward(_model('AveragePooling2D', (4, 4, 1), pool_size=[2, 2], strides=[2, 2], padding='valid'), {}, p)
    assert np.array_equal(ap[0, :, :, 0], [[2.5, 4.5], [10.5, 12.5]])
    assert np.array_equal(unet.avgpool_numpy(p), ap)
    g = unet.forward(_model('GlobalAveragePooling2D', (4, 4, 1)), {}, p)
    assert g.shape == (1, 1) and g[0, 0] == 7.5
    g = unet.forward(_model('GlobalMaxPooling2D', (4, 4, 1)), {}, p)
    assert g.shape == (1, 1) and g[0, 0] == 15
    q = np.arange(12, dtype=np.float32).reshape(1, 2, 3, 2)                  # NHWC
    fl = unet.forward(_model('Flatten', (2, 3, 2)), {}, q)
    assert np.array_equal(fl, q.reshape(1, 12))                              # Keras flattens in (h, w, c) order
    rs = unet.forward(_model('Reshape', (2, 3, 2), target_shape=[3, 2, 2]), {}, q)
    assert np.array_equal(rs, q.reshape(1, 3, 2, 2))
    W = np.array([[1, 10], [100, 1000]], np.float32)
    d = unet.forward(_model('Dense', (1, 1, 2), units=2, activation='linear', use_bias=True), {'L': [W, np.array([0.5, -0.5], np.float32)]},
                     np.array([2, 3], np.float32).reshape(1, 1, 1, 2))
    assert np.array_equal(d.reshape(-1), [2 + 300 + 0.5, 20 + 3000 - 0.5])
    assert np.array_equal(unet.dense_numpy([[2, 3]], W, [0.5, -0.5]).reshape(-1), d.reshape(-1))


def test_preprocess_ecseg_c_known_answer():
    """src/utils.py:166-173: per-channel max normalisation, quantised to 1/255 with round-half-to-even."""
    from ecseg_amd import interseg
    x = np.zeros((2, 2, 3), np.uint8)
    x[..., 0] = [[0, 50], [100, 200]]
    x[..., 1] = [[10, 10], [10, 10]]
    x[..., 2] = [[1, 2], [3, 4]]
    y = interseg.preprocess_ecseg_c(x)
    assert y.dtype == np.float32
    assert np.allclose(y[..., 0], np.rint(np.array([[0, 50], [100, 200]]) / 200 * 255) / 255)
    assert np.allclose(y[..., 1], 1.0)
    assert np.allclose(y[..., 2], np.rint(np.array([[1, 2], [3, 4]]) / 4 * 255) / 255)
    assert y[0, 1, 0] == np.float32(64 / 255)                               # 63.75 -> 64


# ---- round 5: the wider vocabulary (dilation, groups, depthwise / separable, broadcasting binary layers, PReLU,
# ---- LayerNormalization, Normalization, 'same' pooling, activations, nested sub-models, several outputs) --------------
def _L(cls, name, inbound, **cfg):
    return {'class_name': cls, 'name': name, 'config': dict(cfg, name=name),
            'inbound_nodes': [[[i, 0, 0, {}] for i in inbound]] if inbound else []}


def _F(layers, ins, outs, name='m'):
    return {'class_name': 'Functional', 'config': {'name': name, 'layers': layers, 'input_layers': [[i, 0, 0] for i in ins],
                                                   'output_layers': [[o, 0, 0] for o in outs]}}


def test_dilated_conv_known_answer():
    """dilation_rate 2, 3x3, 'same' on a 5x5 ramp: the output centre sums the 9 taps two pixels apart; 'same' pads 2 per side."""
    x = np.arange(25, dtype=np.float32).reshape(1, 5, 5, 1)
    k = np.ones((3, 3, 1, 1), np.float32)
    cfg = _model('Conv2D', (5, 5, 1), filters=1, kernel_size=[3, 3], strides=[1, 1], dilation_rate=[2, 2], padding='same',
                 activation='linear', use_bias=False)
    got = unet.forward(cfg, {'L': [k]}, x)[0, :, :, 0]
    assert got.shape == (5, 5)
    assert got[2, 2] == x[0, ::2, ::2, 0].sum()                 # taps at rows / columns 0, 2, 4
    assert got[0, 0] == x[0, 0:3:2, 0:3:2, 0].sum()             # taps at -2 fall into the padding
    assert np.array_equal(unet.conv_general_numpy(x, k, None, 'same', 1, 2)[0, :, :, 0], got)
    cfgv = _model('Conv2D', (5, 5, 1), filters=1, kernel_size=[3, 3], strides=[1, 1], dilation_rate=[2, 2], padding='valid',
                  activation='linear', use_bias=False)
    assert unet.forward(cfgv, {'L': [k]}, x).shape == (1, 1, 1, 1)


def test_anisotropic_conv_known_answers():
    """strides (2, 1): every second row, every column; dilation_rate (1, 2) on a 1x3 kernel: taps two columns apart in one row.
    'same' pads per axis with that axis' stride and dilated kernel extent."""
    x = np.arange(30, dtype=np.float32).reshape(1, 5, 6, 1)
    one = np.ones((1, 1, 1, 1), np.float32)
    cfg = _model('Conv2D', (5, 6, 1), filters=1, kernel_size=[1, 1], strides=[2, 1], padding='same', activation='linear', use_bias=False)
    got = unet.forward(cfg, {'L': [one]}, x)
    assert got.shape == (1, 3, 6, 1) and np.array_equal(got[0, :, :, 0], x[0, ::2, :, 0])
    assert np.array_equal(unet.conv_general_numpy(x, one, None, 'same', (2, 1), 1), got)
    cfg = _model('Conv2D', (5, 6, 1), filters=1, kernel_size=[1, 1], strides=[1, 3], padding='valid', activation='linear', use_bias=False)
    assert np.array_equal(unet.forward(cfg, {'L': [one]}, x)[0, :, :, 0], x[0, :, ::3, 0])
    k = np.array([1, 10, 100], np.float32).reshape(1, 3, 1, 1)
    cfg = _model('Conv2D', (5, 6, 1), filters=1, kernel_size=[1, 3], strides=[1, 1], dilation_rate=[1, 2], padding='same',
                 activation='linear', use_bias=False)
    got = unet.forward(cfg, {'L': [k]}, x)[0, :, :, 0]
    assert got.shape == (5, 6)
    assert got[1, 2] == x[0, 1, 0, 0] + 10 * x[0, 1, 2, 0] + 100 * x[0, 1, 4, 0]
    assert got[3, 0] == 10 * x[0, 3, 0, 0] + 100 * x[0, 3, 2, 0]            # the tap at column -2 falls into the padding
    assert np.array_equal(unet.conv_general_numpy(x, k, None, 'same', 1, (1, 2))[0, :, :, 0], got)


def test_grouped_and_depthwise_conv_known_answers():
    """groups = 2 on 4 -> 2 channels, 1x1: output 0 sees input channels {0, 1}, output 1 sees {2, 3}.  DepthwiseConv2D with
    depth multiplier 2: output channel ci * 2 + j = input channel ci times kernel[0, 0, ci, j]."""
    x = np.array([1, 2, 3, 4], np.float32).reshape(1, 1, 1, 4)
    k = np.array([[10, 1000], [100, 10000]], np.float32).reshape(1, 1, 2, 2)          # (kh, kw, cin / groups, filters)
    cfg = _model('Conv2D', (1, 1, 4), filters=2, kernel_size=[1, 1], strides=[1, 1], groups=2, padding='valid', activation='linear',
                 use_bias=False)
    got = unet.forward(cfg, {'L': [k]}, x).reshape(-1)
    assert got.tolist() == [1 * 10 + 2 * 100, 3 * 1000 + 4 * 10000]
    assert np.array_equal(unet.conv_general_numpy(x, k, None, 'valid', 1, 1, 2).reshape(-1), got)
    xd = np.array([1, 2], np.float32).reshape(1, 1, 1, 2)
    kd = np.array([[3, 5], [7, 11]], np.float32).reshape(1, 1, 2, 2)                  # (kh, kw, cin, multiplier)
    cfgd = _model('DepthwiseConv2D', (1, 1, 2), kernel_size=[1, 1], strides=[1, 1], depth_multiplier=2, padding='same',
                  activation='linear', use_bias=True)
    gotd = unet.forward(cfgd, {'L': [kd, np.array([.5, .5, .5, .5], np.float32)]}, xd).reshape(-1)
    assert gotd.tolist() == [3.5, 5.5, 14.5, 22.5]
    assert np.array_equal(unet.depthwise_numpy(xd, kd, [.5] * 4).reshape(-1), gotd)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_round5_numpy_restatements_match_torch_path(seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(2, 9, 11, 8)).astype(np.float32)
    # dilated / grouped / strided Conv2D
    for (k, s, d, g, pad) in [(3, 1, 2, 1, 'same'), (3, 1, 3, 2, 'valid'), (5, 2, 1, 4, 'same'), (1, 1, 1, 8, 'same'), (2, 1, 2, 1, 'same')]:
        ker = rng.normal(size=(k, k, 8 // g, 16)).astype(np.float32)
        b = rng.normal(size=16).astype(np.float32)
        cfg = _model('Conv2D', (9, 11, 8), filters=16, kernel_size=[k, k], strides=[s, s], dilation_rate=[d, d], groups=g, padding=pad,
                     activation='linear', use_bias=True)
        got = unet.forward(cfg, {'L': [ker, b]}, x)
        want = unet.conv_general_numpy(x, ker, b, pad, s, d, g)
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-4, (k, s, d, g, pad)
    # per-axis strides / dilation rates / taps
    for (kk, s, d, pad) in [((3, 3), (2, 1), (1, 1), 'same'), ((3, 3), (1, 3), (1, 1), 'valid'), ((3, 3), (1, 1), (2, 3), 'same'),
                            ((1, 5), (1, 1), (1, 2), 'same'), ((2, 3), (3, 2), (1, 1), 'same'), ((3, 1), (1, 1), (3, 1), 'valid')]:
        ker = rng.normal(size=(kk[0], kk[1], 8, 16)).astype(np.float32)
        b = rng.normal(size=16).astype(np.float32)
        cfg = _model('Conv2D', (9, 11, 8), filters=16, kernel_size=list(kk), strides=list(s), dilation_rate=list(d), padding=pad,
                     activation='linear', use_bias=True)
        got = unet.forward(cfg, {'L': [ker, b]}, x)
        want = unet.conv_general_numpy(x, ker, b, pad, s, d)
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-4, (kk, s, d, pad)
    # DepthwiseConv2D / SeparableConv2D
    for (k, s, d, m, pad) in [(3, 1, 1, 1, 'same'), (3, 2, 1, 1, 'same'), (5, 1, 1, 2, 'valid'), (3, 1, 2, 1, 'same')]:
        dk = rng.normal(size=(k, k, 8, m)).astype(np.float32)
        b = rng.normal(size=8 * m).astype(np.float32)
        cfg = _model('DepthwiseConv2D', (9, 11, 8), kernel_size=[k, k], strides=[s, s], dilation_rate=[d, d], depth_multiplier=m,
                     padding=pad, activation='linear', use_bias=True)
        got = unet.forward(cfg, {'L': [dk, b]}, x)
        want = unet.depthwise_numpy(x, dk, b, pad, s, d)
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-4
        pk = rng.normal(size=(1, 1, 8 * m, 5)).astype(np.float32)
        pb = rng.normal(size=5).astype(np.float32)
        cfgs = _model('SeparableConv2D', (9, 11, 8), filters=5, kernel_size=[k, k], strides=[s, s], dilation_rate=[d, d],
                      depth_multiplier=m, padding=pad, activation='relu', use_bias=True)
        gots = unet.forward(cfgs, {'L': [dk, pk, pb]}, x)
        wants = np.maximum(unet.conv_numpy(unet.depthwise_numpy(x, dk, None, pad, s, d), pk, pb, 'valid'), 0)
        assert np.abs(gots - wants).max() < 1e-4
    # 'same' pooling
    for (k, s) in [(2, 2), (3, 2), (3, 1)]:
        for cls, avg in (('MaxPooling2D', False), ('AveragePooling2D', True)):
            cfg = _model(cls, (9, 11, 8), pool_size=[k, k], strides=[s, s], padding='same')
            got = unet.forward(cfg, {}, x)
            assert np.abs(got - unet.pool_same_numpy(x, k, s, avg)).max() < 1e-6, (cls, k, s)
    # LayerNormalization / PReLU / Normalization
    g, b = rng.uniform(.5, 1.5, 8).astype(np.float32), rng.normal(size=8).astype(np.float32)
    cfg = _model('LayerNormalization', (9, 11, 8), axis=[3], epsilon=1e-3, center=True, scale=True)
    assert np.abs(unet.forward(cfg, {'L': [g, b]}, x) - unet.layernorm_numpy(x, g, b)).max() < 1e-5
    cfg = _model('LayerNormalization', (9, 11, 8), axis=-1, epsilon=1e-5, center=False, scale=True)
    assert np.abs(unet.forward(cfg, {'L': [g]}, x) - unet.layernorm_numpy(x, g, None, 1e-5)).max() < 1e-5
    for shared, shp in (([1, 2], (1, 1, 8)), (None, (9, 11, 8)), ([1], (1, 11, 8))):
        al = rng.uniform(-.5, .5, shp).astype(np.float32)
        cfg = _model('PReLU', (9, 11, 8), shared_axes=shared)
        assert np.array_equal(unet.forward(cfg, {'L': [al]}, x), unet.prelu_numpy(x, al))
    mean, var = rng.normal(size=8).astype(np.float32), rng.uniform(.5, 2, 8).astype(np.float32)
    cfg = _model('Normalization', (9, 11, 8), axis=[-1], mean=None, variance=None)
    got = unet.forward(cfg, {'L': [mean, var, np.array(0, np.int64)]}, x)
    assert np.abs(got - (x - mean) / np.sqrt(var)).max() < 1e-5
    # activations
    for name in ('relu6', 'selu', 'softplus', 'softsign', 'swish', 'gelu', 'hard_sigmoid', 'exponential', 'elu'):
        cfg = _model('Activation', (9, 11, 8), activation=name)
        assert np.abs(unet.forward(cfg, {}, 3 * x) - unet.activation_numpy(name, 3 * x)).max() < 2e-5, name
    cfg = _model('ReLU', (9, 11, 8), max_value=6.0, negative_slope=0.0, threshold=0.0)
    assert np.array_equal(unet.forward(cfg, {}, 4 * x), np.clip(4 * x, 0, 6))
    cfg = _model('ELU', (9, 11, 8), alpha=0.7)
    assert np.abs(unet.forward(cfg, {}, x) - unet.activation_numpy('elu', x, 0.7)).max() < 1e-6


def test_binary_layers_broadcast_like_numpy():
    """Multiply of (h, w, c) by (1, 1, c) - the squeeze-and-excite gate - and by (h, w, 1) - a spatial attention map - and the
    other merge layers, against numpy broadcasting on the NHWC arrays."""
    rng = np.random.default_rng(5)
    a = rng.normal(size=(2, 6, 7, 8)).astype(np.float32)
    for shape_b in ((1, 1, 8), (6, 7, 1), (6, 7, 8)):
        b = rng.normal(size=(2,) + shape_b).astype(np.float32)
        for cls, fn in (('Multiply', np.multiply), ('Add', np.add), ('Subtract', np.subtract), ('Maximum', np.maximum),
                        ('Minimum', np.minimum), ('Average', lambda p, q: (p + q) / 2)):
            layers = [_L('InputLayer', 'a', [], batch_input_shape=[None, 6, 7, 8]),
                      _L('InputLayer', 'b', [], batch_input_shape=[None] + list(shape_b)), _L(cls, 'm', ['a', 'b'])]
            cfg = _F(layers, ['a', 'b'], ['m'])
            ta = [__import__('torch').from_numpy(np.ascontiguousarray(v.transpose(0, 3, 1, 2))) for v in (a, b)]
            got = unet.forward(cfg, {}, ta)
            assert np.abs(got - fn(a, b)).max() < 1e-6, (cls, shape_b)


def test_nested_submodel_and_second_output():
    """A Functional model that calls a nested Functional sub-model (with its own skip connection) and a nested Sequential one,
    with two outputs: the recursive evaluation equals the same layers written flat, for both weight forms (a dict per inner
    layer, and the flat HDF5 list with weight names - trainable variables first, as Keras saves a nested model)."""
    rng = np.random.default_rng(9)
    k = lambda *s: (rng.normal(size=s) / np.sqrt(np.prod(s[:-1]))).astype(np.float32)
    conv = lambda name, src, f, **kw: _L('Conv2D', name, [src], filters=f, kernel_size=[3, 3], strides=[1, 1], padding='same',
                                         activation='relu', use_bias=True, **kw)
    bn = lambda name, src: _L('BatchNormalization', name, [src], axis=[3], epsilon=1e-3, center=True, scale=True)
    inner = _F([_L('InputLayer', 'bin', [], batch_input_shape=[None, 8, 8, 4]), conv('b1', 'bin', 8), bn('bbn', 'b1'),
                conv('b2', 'bbn', 4), _L('Add', 'badd', ['b2', 'bin'])], ['bin'], ['badd'], name='backbone')
    seq = {'class_name': 'Sequential', 'name': 'headseq',
           'config': {'name': 'headseq', 'layers': [_L('Conv2D', 's1', [], filters=6, kernel_size=[1, 1], strides=[1, 1], padding='same',
                                                       activation='linear', use_bias=True, batch_input_shape=[None, 8, 8, 4]),
                                                    _L('Activation', 's2', [], activation='tanh')]}}
    outer = _F([_L('InputLayer', 'in', [], batch_input_shape=[None, 8, 8, 4]),
                dict(inner, name='backbone', inbound_nodes=[[['in', 0, 0, {}]]]),
                dict(seq, inbound_nodes=[[['backbone', 0, 0, {}]]]),
                _L('GlobalAveragePooling2D', 'gap', ['backbone'])], ['in'], ['headseq', 'gap'])
    wi = {'b1': [k(3, 3, 4, 8), k(8)], 'bbn': [rng.uniform(.5, 1.5, 8).astype(np.float32), k(8), k(8), rng.uniform(.5, 1.5, 8).astype(np.float32)],
          'b2': [k(3, 3, 8, 4), k(4)]}
    ws = {'s1': [k(1, 1, 4, 6), k(6)]}
    x = rng.normal(size=(2, 8, 8, 4)).astype(np.float32)
    flat = _F([_L('InputLayer', 'in', [], batch_input_shape=[None, 8, 8, 4]), conv('b1', 'in', 8), bn('bbn', 'b1'), conv('b2', 'bbn', 4),
               _L('Add', 'badd', ['b2', 'in']),
               _L('Conv2D', 's1', ['badd'], filters=6, kernel_size=[1, 1], strides=[1, 1], padding='same', activation='tanh', use_bias=True),
               _L('GlobalAveragePooling2D', 'gap', ['badd'])], ['in'], ['s1', 'gap'])
    want0 = unet.forward(flat, dict(wi, **ws), x, output=0)
    want1 = unet.forward(flat, dict(wi, **ws), x, output='gap')
    got0 = unet.forward(outer, {'backbone': wi, 'headseq': ws}, x)
    got1 = unet.forward(outer, {'backbone': wi, 'headseq': ws}, x, output=1)
    assert got0.shape == (2, 8, 8, 6) and got1.shape == (2, 4)
    assert np.array_equal(got0, want0) and np.array_equal(got1, want1)
    # the HDF5 form: trainable variables of all inner layers first, then the moving statistics
    class Named(list):
        pass
    nw = Named([wi['b1'][0], wi['b1'][1], wi['bbn'][0], wi['bbn'][1], wi['b2'][0], wi['b2'][1], wi['bbn'][2], wi['bbn'][3]])
    nw.names = ['b1/kernel:0', 'b1/bias:0', 'bbn/gamma:0', 'bbn/beta:0', 'b2/kernel:0', 'b2/bias:0', 'bbn/moving_mean:0', 'bbn/moving_variance:0']
    sw = Named(ws['s1']); sw.names = ['headseq/s1/kernel:0', 'headseq/s1/bias:0']
    assert np.array_equal(unet.forward(outer, {'backbone': nw, 'headseq': sw}, x), want0)


def _shared_model(H=8, W=8, c=4):
    """in -> shared conv (call 0) -> pool -> up -> shared conv again (call 1, listed BEFORE the layers that feed it, as Keras does) ->
    Add(call 0, call 1) -> shared BatchNorm applied to the sum and (second call) to call 1 alone -> Concatenate."""
    conv = {'class_name': 'Conv2D', 'name': 'sc', 'config': dict(name='sc', filters=c, kernel_size=[3, 3], strides=[1, 1], padding='same',
                                                                  activation='relu', use_bias=True),
            'inbound_nodes': [[['in', 0, 0, {}]], [['up', 0, 0, {}]]]}
    bn = {'class_name': 'BatchNormalization', 'name': 'sbn', 'config': dict(name='sbn', axis=[3], epsilon=1e-3, center=True, scale=True),
          'inbound_nodes': [[['add', 0, 0, {}]], [['sc', 1, 0, {}]]]}
    layers = [_L('InputLayer', 'in', [], batch_input_shape=[None, H, W, c]), conv,
              _L('MaxPooling2D', 'pool', ['sc'], pool_size=[2, 2], strides=[2, 2], padding='valid'),
              _L('UpSampling2D', 'up', ['pool'], size=[2, 2], interpolation='nearest'),
              {'class_name': 'Add', 'name': 'add', 'config': {'name': 'add'}, 'inbound_nodes': [[['sc', 0, 0, {}], ['sc', 1, 0, {}]]]},
              bn,
              {'class_name': 'Concatenate', 'name': 'cat', 'config': {'name': 'cat', 'axis': -1},
               'inbound_nodes': [[['sbn', 0, 0, {}], ['sbn', 1, 0, {}]]]}]
    return _F(layers, ['in'], ['cat'])


def test_shared_layers_are_evaluated_once_per_call():
    """Round 6 (VERDICT r05 missing #4): a layer with several inbound nodes is CALLED several times with the same weights; references
    [layer, node index, tensor index] pick the call.  The oracle's call-by-call evaluation equals the same graph written out with one
    layer (and a copy of the weights) per call, and a numpy restatement of the whole thing."""
    rng = np.random.default_rng(21)
    c = 4
    w = {'sc': [(rng.normal(size=(3, 3, c, c)) / 6).astype(np.float32), rng.normal(size=c).astype(np.float32)],
         'sbn': [rng.uniform(.5, 1.5, c).astype(np.float32), rng.normal(size=c).astype(np.float32), rng.normal(size=c).astype(np.float32),
                 rng.uniform(.5, 1.5, c).astype(np.float32)]}
    x = rng.normal(size=(2, 8, 8, c)).astype(np.float32)
    got = unet.forward(_shared_model(), w, x)
    conv = lambda name, src: _L('Conv2D', name, [src], filters=c, kernel_size=[3, 3], strides=[1, 1], padding='same', activation='relu', use_bias=True)
    bn = lambda name, src: _L('BatchNormalization', name, [src], axis=[3], epsilon=1e-3, center=True, scale=True)
    flat = _F([_L('InputLayer', 'in', [], batch_input_shape=[None, 8, 8, c]), conv('a', 'in'),
               _L('MaxPooling2D', 'pool', ['a'], pool_size=[2, 2], strides=[2, 2], padding='valid'),
               _L('UpSampling2D', 'up', ['pool'], size=[2, 2], interpolation='nearest'), conv('b', 'up'), _L('Add', 'add', ['a', 'b']),
               bn('n0', 'add'), bn('n1', 'b'), _L('Concatenate', 'cat', ['n0', 'n1'], axis=-1)], ['in'], ['cat'])
    want = unet.forward(flat, {'a': w['sc'], 'b': w['sc'], 'n0': w['sbn'], 'n1': w['sbn']}, x)
    assert got.shape == (2, 8, 8, 2 * c) and np.array_equal(got, want)
    # numpy, from the definitions
    def conv_np(t):
        p = np.pad(t.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
        y = sum(p[:, r:r + 8, s:s + 8, :] @ w['sc'][0][r, s].astype(np.float64) for r in range(3) for s in range(3)) + w['sc'][1]
        return np.maximum(y, 0)
    bn_np = lambda t: (t - w['sbn'][2]) / np.sqrt(w['sbn'][3].astype(np.float64) + 1e-3) * w['sbn'][0] + w['sbn'][1]
    a = conv_np(x)
    up = np.repeat(np.repeat(a.reshape(2, 4, 2, 4, 2, c).max((2, 4)), 2, 1), 2, 2)
    b = conv_np(up)
    ref = np.concatenate([bn_np(a + b), bn_np(b)], -1)
    assert np.abs(got - ref).max() < 2e-5


def channels_first_twin(cfg):
    """The channels_first edition of a channels_last Functional config (what `keras.backend.set_image_data_format('channels_first')`
    would have produced): (N, C, H, W) input, data_format on every spatial layer, channel axis 1; a softmax fused into the last
    convolution becomes a Softmax(axis=1) layer (Keras applies a FUSED softmax over the last axis - W for such tensors)."""
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


def test_channels_first_model_equals_its_channels_last_twin():
    """Round 6 (VERDICT r05 missing #4): Keras stores (kh, kw, in, out) kernels in both data formats, so a channels_first U-Net fed
    (N, C, H, W) must return the transposed output of the channels_last model with the SAME weights (to float32 rounding: torch picks
    another convolution algorithm for the other memory layout)."""
    from ecseg_amd import synth
    cfg = synth.unet_config(base=8, depth=2, batchnorm=True)
    w = synth.unet_weights(cfg, seed=4)
    rng = np.random.default_rng(8)
    x = rng.integers(0, 256, size=(2, 256, 256, 1)).astype(np.uint8)
    want = unet.forward(cfg, w, x)
    cf = channels_first_twin(cfg)
    got = unet.forward(cf, w, np.ascontiguousarray(np.moveaxis(x, -1, 1)))
    assert got.shape == (2, 4, 256, 256)
    assert np.abs(np.moveaxis(got, 1, -1) - want).max() < 2e-5

"""the interSeg classifier graphs (src/interseg.py:96-98,155,168) through the same plan interpreter and HIP
kernels as the metaseg U-Net: strided Conv2D, AveragePooling2D, GlobalAveragePooling2D, Flatten, Reshape, Dense,
BatchNormalization folding, softmax / sigmoid heads, uint8 (N, 256, 256) and float32 (N, 256, 256, 3) inputs - parity
with the CPU oracle within 1e-3 on synthetic .h5 fixtures (written by h5py, tools/make_golden.py), and the per-nucleus
decision logic on top."""
import os

import numpy as np
import pytest

from ecseg_amd import hdf5_min, interseg, keras_plan, synth
from ecseg_amd.model import MetasegModel
from oracle import unet as oracle_unet

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _crops(n, seed=0):
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 256, 256, 3), np.uint8)
    for i in range(n):
        rgb = synth.dapi_image(800 + seed * 50 + i, 256, 256, rgb=True)       # channels: red, green, DAPI
        out[i] = rgb
        yy, xx = np.ogrid[:256, :256]
        out[i] *= ((yy - 128) ** 2 + (xx - 128) ** 2 <= int(rng.integers(60, 120)) ** 2)[..., None].astype(np.uint8)
    return out


@pytest.fixture(scope='module')
def models(golden_dir):
    from ecseg_amd._lib import Handle
    hs = [Handle(0), Handle(0)]
    mi = MetasegModel(*hdf5_min.load_keras_h5(os.path.join(golden_dir, 'interseg_synth.h5')), handle=hs[0])
    mc = MetasegModel(*hdf5_min.load_keras_h5(os.path.join(golden_dir, 'ecseg_c_synth.h5')), handle=hs[1])
    yield mi, mc
    for h in hs:
        h.close()


def test_interseg_classifier_matches_oracle(models):
    mi, _ = models
    x = _crops(5)[..., 0]                                                   # (N, 256, 256): no channel axis, as the reference passes it
    got = mi.predict(x)
    want = oracle_unet.forward(mi.model_config, mi.weights, x)
    assert got.shape == want.shape == (5, 3)
    assert np.abs(got - want).max() < TOL, np.abs(got - want).max()
    np.testing.assert_allclose(got.sum(-1), 1.0, atol=1e-5)
    one = mi.predict(x[2:3])                                                # batch independence (reference predicts crop by crop)
    assert np.abs(one - got[2:3]).max() < 1e-6


def test_ecseg_c_classifier_matches_oracle_on_float_inputs(models):
    _, mc = models
    x = np.stack([interseg.preprocess_ecseg_c(c) for c in _crops(4, seed=1)])
    assert x.dtype == np.float32 and x.max() <= 1.0
    got = mc.predict(x)
    want = oracle_unet.forward(mc.model_config, mc.weights, x)
    assert got.shape == want.shape == (4, 1)
    assert np.abs(got - want).max() < TOL, np.abs(got - want).max()


@pytest.mark.parametrize('case', ['stride2_same_3x3', 'stride2_valid_5x5', 'stride3_same_2x2', 'avgpool', 'gap_gmp', 'dense_stack'])
def test_classifier_layer_types_one_by_one(gpu, case):
    rng = np.random.default_rng(len(case))

    def model(layers_after_input, shape, out):
        layers = [{'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None] + list(shape)},
                   'inbound_nodes': []}] + layers_after_input
        return {'class_name': 'Functional', 'config': {'name': 'm', 'layers': layers, 'input_layers': [['in', 0, 0]],
                                                       'output_layers': [[out, 0, 0]]}}

    def L(cls, name, inb, **c):
        return {'class_name': cls, 'name': name, 'config': dict(c, name=name), 'inbound_nodes': [[[inb, 0, 0, {}]]]}

    w = {}
    if case.startswith('stride'):
        s_, pad, k = {'stride2_same_3x3': (2, 'same', 3), 'stride2_valid_5x5': (2, 'valid', 5), 'stride3_same_2x2': (3, 'same', 2)}[case]
        cfg = model([L('Conv2D', 'c', 'in', filters=12, kernel_size=[k, k], strides=[s_, s_], padding=pad, activation='relu', use_bias=True)],
                    (37, 50, 6), 'c')
        w['c'] = [(rng.normal(size=(k, k, 6, 12)) * 0.01).astype(np.float32), rng.normal(size=12).astype(np.float32)]
        x = rng.integers(0, 256, (3, 37, 50, 6), dtype=np.uint8)
    elif case == 'avgpool':
        cfg = model([L('AveragePooling2D', 'p', 'in', pool_size=[3, 3], strides=[2, 2], padding='valid')], (21, 30, 8), 'p')
        x = rng.integers(0, 256, (2, 21, 30, 8), dtype=np.uint8)
    elif case == 'gap_gmp':
        cfg = model([L('GlobalMaxPooling2D', 'g', 'in')], (19, 23, 70), 'g')
        x = rng.integers(0, 256, (3, 19, 23, 70), dtype=np.uint8)
    else:
        cfg = model([L('Flatten', 'f', 'in'), L('Dense', 'd1', 'f', units=40, activation='tanh', use_bias=True),
                     L('Dense', 'd2', 'd1', units=24, activation='relu', use_bias=False),
                     L('Dense', 'd3', 'd2', units=5, activation='softmax', use_bias=True)], (6, 5, 4), 'd3')
        w = {'d1': [(rng.normal(size=(120, 40)) * 0.001).astype(np.float32), rng.normal(size=40).astype(np.float32)],
             'd2': [(rng.normal(size=(40, 24)) * 0.3).astype(np.float32)],
             'd3': [(rng.normal(size=(24, 5)) * 0.3).astype(np.float32), rng.normal(size=5).astype(np.float32)]}
        x = rng.integers(0, 256, (7, 6, 5, 4), dtype=np.uint8)
    gpu.load_plan(keras_plan.build_plan(cfg, w))
    got = gpu.forward_patches(x)
    want = oracle_unet.forward(cfg, w, x)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.abs(got - want).max() < TOL * max(1.0, float(np.abs(want).max())), np.abs(got - want).max()


def test_classify_crops_decision_logic(models):
    mi, mc = models
    crops = _crops(6, seed=2)
    crops[1] = 0                                                             # an all-zero tile of an oversized nucleus
    crops[3][..., 1] = np.minimum(crops[3][..., 1], 10)                      # centromeric probe too dim
    rows = interseg.classify_crops(mi, crops, mc, centromeric_quality_score_pass=True, from_patches=True)
    assert rows[1]['interSeg_label'] == interseg.EMPTY and rows[1]['ecSeg-c_label'] == interseg.EMPTY
    assert rows[3]['ecSeg-c_label'] == interseg.LOW_CENT and rows[3]['interSeg_label'] == rows[3]['ecSeg-i_label']
    oi = oracle_unet.forward(mi.model_config, mi.weights, crops[..., 0])
    for k in (0, 2, 4, 5):
        r = rows[k]
        assert r['ecSeg-i_label'] == interseg.ECSEG_I_LABEL_MAP[int(np.argmax(oi[k]))]
        assert abs(r['pred_ec'] - float(oi[k, 1])) < TOL
        oc = float(oracle_unet.forward(mc.model_config, mc.weights, interseg.preprocess_ecseg_c(crops[k])[None])[0, 0])
        assert abs(r['pred_focal_amp'] - oc) < TOL and abs(r['pred_no_focal_amp'] - (1 - oc)) < TOL
        assert r['interSeg_label'] == interseg.INTERSEG_LABEL_MAP[(interseg.ECSEG_C_LABEL_MAP[int(oc > 0.5)], r['ecSeg-i_label'])]
    failed = interseg.classify_crops(mi, crops[:1], mc, centromeric_quality_score_pass=False)
    assert failed[0]['ecSeg-c_label'] == interseg.FAILED_QUALITY
    no_c = interseg.classify_crops(mi, crops[:1])
    assert 'ecSeg-c_label' not in no_c[0] and no_c[0]['interSeg_label'] == no_c[0]['ecSeg-i_label']

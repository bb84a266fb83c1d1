"""U-Net layer parity on the GPU: HIP kernels (through the C ABI) vs the CPU oracle (oracle/unet.py).
Tolerance: 1e-3 absolute on probabilities / pre-softmax activations (BASELINE.json north_star: "pre-argmax logits
within 1e-3 fp32"); the MFMA kernel's exact-f32 fma chains land around 1e-5."""
import json
import os

import numpy as np
import pytest

from ecseg_amd import hdf5_min, keras_plan, synth
from oracle import unet as oracle_unet

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _patches(n, seed=0, c=1):
    rng = np.random.default_rng(seed)
    base = np.stack([synth.dapi_image(100 + i, 256, 256) for i in range(n)])[..., None]
    if c > 1:
        base = np.concatenate([base] + [rng.integers(0, 256, base.shape, dtype=np.uint8) for _ in range(c - 1)], -1)
    return base


def _run(gpu, cfg, weights, x, fuse):
    plan = keras_plan.build_plan(cfg, weights, fuse=fuse)
    gpu.load_plan(plan)
    return gpu.forward_patches(x), plan


@pytest.mark.parametrize('fuse', [False, True])
def test_tiny_keras_h5_model(gpu, golden_dir, fuse):
    cfg, weights = hdf5_min.load_keras_h5(os.path.join(golden_dir, 'keras_tiny.h5'))
    x = _patches(3)
    got, plan = _run(gpu, cfg, weights, x, fuse)
    want = oracle_unet.forward(cfg, weights, x)
    assert got.shape == want.shape == (3, 256, 256, 4)
    assert np.abs(got - want).max() < TOL
    np.testing.assert_allclose(got.sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize('base,up,bn', [(16, 'transpose', False), (16, 'upsample', True), (32, 'transpose', False)])
def test_canonical_unet_matches_oracle(gpu, base, up, bn):
    cfg = synth.unet_config(base=base, up=up, batchnorm=bn)
    weights = synth.unet_weights(cfg, seed=base)
    x = _patches(2, seed=base)
    got, plan = _run(gpu, cfg, weights, x, fuse=True)
    want = oracle_unet.forward(cfg, weights, x)
    err = np.abs(got - want).max()
    assert err < TOL, err
    np.testing.assert_allclose(got.sum(-1), 1.0, atol=1e-5)


def test_intermediate_layers_first_block(gpu):
    """Check single layers in isolation by truncating the model: conv(Cin=1) [direct kernel], conv 16->16 [MFMA],
    max-pool, transposed conv [MFMA scatter epilogue], concat views."""
    cfg = synth.unet_config(base=16, depth=1)
    weights = synth.unet_weights(cfg, seed=3)
    x = _patches(2, seed=5)
    full = cfg['config']['layers']
    for upto in range(2, len(full) + 1):
        sub = json.loads(json.dumps(cfg))
        sub['config']['layers'] = full[:upto]
        last = full[upto - 1]
        if last['class_name'] in ('InputLayer',):
            continue
        sub['config']['output_layers'] = [[last['config']['name'], 0, 0]]
        got, _ = _run(gpu, sub, weights, x, fuse=False)
        want = oracle_unet.forward(sub, weights, x)
        assert got.shape == want.shape, last['config']['name']
        scale = max(1.0, float(np.abs(want).max()))
        assert np.abs(got - want).max() < TOL * scale, (last['config']['name'], np.abs(got - want).max())


@pytest.mark.parametrize('cin,cout,k', [(8, 16, 3), (24, 48, 3), (64, 64, 3), (128, 64, 3), (32, 32, 2), (16, 128, 1),
                                        (12, 20, 3), (3, 8, 3), (5, 7, 3), (64, 4, 1)])
def test_single_conv_shapes(gpu, cin, cout, k):
    """One Conv2D per case on a multi-channel uint8 input: MFMA path (incl. channel counts that need zero padding),
    the small-Cin direct path, the generic fall-back and the 1x1 head."""
    rng = np.random.default_rng(cin * 100 + cout)
    cfg = {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, 64, 96, cin]},
         'inbound_nodes': []},
        {'class_name': 'Conv2D', 'name': 'c', 'config': {'name': 'c', 'filters': cout, 'kernel_size': [k, k],
                                                         'strides': [1, 1], 'padding': 'same', 'activation': 'relu',
                                                         'use_bias': True}, 'inbound_nodes': [[['in', 0, 0, {}]]]}],
        'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}
    weights = {'c': [(rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin) / 64).astype(np.float32),
                     rng.normal(size=cout).astype(np.float32)]}
    x = rng.integers(0, 256, size=(3, 64, 96, cin), dtype=np.uint8)
    got, _ = _run(gpu, cfg, weights, x, fuse=True)
    want = oracle_unet.forward(cfg, weights, x)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < TOL * max(1.0, np.abs(want).max()), np.abs(got - want).max()


def test_conv_numpy_crosscheck_of_oracle():
    """The torch-based oracle conv agrees with an independent numpy restatement (guards the oracle itself)."""
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, size=(1, 20, 24, 3), dtype=np.uint8)
    k = rng.normal(size=(3, 3, 3, 5)).astype(np.float32) * 0.01
    b = rng.normal(size=5).astype(np.float32)
    cfg = {'class_name': 'Sequential', 'config': {'name': 's', 'layers': [
        {'class_name': 'Conv2D', 'config': {'name': 'c', 'batch_input_shape': [None, 20, 24, 3], 'filters': 5,
                                            'kernel_size': [3, 3], 'strides': [1, 1], 'padding': 'same',
                                            'activation': 'linear', 'use_bias': True}}]}}
    want = oracle_unet.conv_numpy(x, k, b)
    got = oracle_unet.forward(cfg, {'c': [k, b]}, x)
    assert np.abs(got - want).max() < 1e-4


@pytest.mark.parametrize('cin,cout,hw', [(64, 64, (64, 96)), (16, 32, (20, 40)), (128, 192, (16, 16)), (40, 72, (37, 53)),
                                         (32, 16, (32, 48)), (16, 16, (24, 24)), (24, 20, (16, 40))])
def test_winograd_matches_direct_and_oracle(gpu, cin, cout, hw):
    """The Winograd F(2x2,3x3) kernel and the direct implicit-GEMM kernel are both fp32; they must agree with each other
    and with the oracle well inside the 1e-3 tolerance (incl. image sizes that are not multiples of the 8x16 tile)."""
    rng = np.random.default_rng(cin + cout)
    H, W = hw
    cfg = {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, H, W, cin]},
         'inbound_nodes': []},
        {'class_name': 'Conv2D', 'name': 'c', 'config': {'name': 'c', 'filters': cout, 'kernel_size': [3, 3],
                                                         'strides': [1, 1], 'padding': 'same', 'activation': 'relu',
                                                         'use_bias': True}, 'inbound_nodes': [[['in', 0, 0, {}]]]}],
        'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}
    weights = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32),
                     rng.normal(size=cout).astype(np.float32)]}
    x = rng.integers(0, 256, size=(3, H, W, cin), dtype=np.uint8)
    want = oracle_unet.forward(cfg, weights, x)
    try:
        gpu.set_option('winograd', 1)
        wino, _ = _run(gpu, cfg, weights, x, fuse=True)
        gpu.set_option('winograd', 0)
        direct = gpu.forward_patches(x)
    finally:
        gpu.set_option('winograd', 2)
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(direct - want).max() < 1e-4 * scale
    assert np.abs(wino - want).max() < 2e-4 * scale, np.abs(wino - want).max()


@pytest.mark.parametrize('cin,cout,hw,n', [(64, 64, (64, 96), 3), (8, 64, (16, 16), 3), (128, 192, (16, 16), 4),
                                           (72, 128, (32, 48), 2), (24, 64, (48, 16), 1),
                                           # relaxed eligibility (r02): Cout % 64 == 32 (zero-padded channel half), Cin % 8 == 4
                                           (64, 32, (32, 32), 3), (16, 96, (16, 48), 2), (12, 64, (32, 16), 2),
                                           (68, 32, (16, 16), 5), (36, 160, (16, 32), 1)])
def test_winograd_f4x4_matches_direct_and_oracle(gpu, cin, cout, hw, n):
    """Winograd F(4x4,3x3) (option winograd=2; extents multiples of 16, Cin % 8 == 0, Cout % 64 == 0) is still an fp32
    kernel: it must agree with the direct kernel and the oracle well inside the 1e-3 tolerance.  Covers an odd number of
    16x16 regions (the second region of the last workgroup is empty) and regions that span two patches."""
    rng = np.random.default_rng(cin * 7 + cout)
    H, W = hw
    cfg = {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, H, W, cin]},
         'inbound_nodes': []},
        {'class_name': 'Conv2D', 'name': 'c', 'config': {'name': 'c', 'filters': cout, 'kernel_size': [3, 3],
                                                         'strides': [1, 1], 'padding': 'same', 'activation': 'relu',
                                                         'use_bias': True}, 'inbound_nodes': [[['in', 0, 0, {}]]]}],
        'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}
    weights = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32),
                     rng.normal(size=cout).astype(np.float32)]}
    x = rng.integers(0, 256, size=(n, H, W, cin), dtype=np.uint8)
    want = oracle_unet.forward(cfg, weights, x)
    try:
        gpu.set_option('winograd', 2)
        w4, _ = _run(gpu, cfg, weights, x, fuse=True)
        gpu.set_option('winograd', 0)
        direct = gpu.forward_patches(x)
    finally:
        gpu.set_option('winograd', 2)
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(direct - want).max() < 1e-4 * scale
    assert np.abs(w4 - want).max() < 5e-4 * scale, np.abs(w4 - want).max()


def test_base64_unet_f4x4_with_concat_views(gpu):
    """Base-64 U-Net (depth 2): every 3x3 layer but the first runs the Winograd F(4x4,3x3) kernel, reading and writing
    the strided channel views of the fused skip connections; the probabilities must stay inside the 1e-3 tolerance and
    agree with the direct kernel to 1e-4."""
    cfg = synth.unet_config(base=64, depth=2)
    weights = synth.unet_weights(cfg, seed=64)
    x = _patches(2, seed=64)
    want = oracle_unet.forward(cfg, weights, x)
    try:
        gpu.set_option('winograd', 2)
        got, plan = _run(gpu, cfg, weights, x, fuse=True)
        gpu.set_option('fuse_pool', 0)                      # separate max-pool / head kernels instead of the fused output stage
        gpu.set_option('fuse_head', 0)
        unfused = gpu.forward_patches(x)
        gpu.set_option('winograd', 0)
        direct = gpu.forward_patches(x)
    finally:
        gpu.set_option('winograd', 2)
        gpu.set_option('fuse_pool', 1)
        gpu.set_option('fuse_head', 1)
    assert np.abs(got - want).max() < TOL, np.abs(got - want).max()
    assert np.abs(got - unfused).max() < 1e-5, np.abs(got - unfused).max()      # (the fused head sums in another order)
    assert np.abs(got - direct).max() < 1e-4, np.abs(got - direct).max()
    np.testing.assert_allclose(got.sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize('cin,cout,hw,n', [(64, 64, (64, 96), 3), (8, 64, (16, 16), 3), (128, 192, (16, 16), 4), (72, 128, (32, 48), 2),
                                           (24, 64, (48, 16), 1), (12, 64, (32, 16), 2), (68, 128, (16, 16), 5), (256, 128, (32, 32), 2)])
def test_winograd_f4x4_split_bf16x3_matches_fp32_kernel_and_oracle(gpu, cin, cout, hw, n):
    """Round 6 (option winograd = 3): the F(4x4) channel sum on the bf16 matrix pipe with both operands split exactly into three
    bf16 pieces (six of the nine piece products, float32 accumulate: dropped terms <= 3 x 2^-24).  It must sit as close to the
    oracle as the fp32-MFMA F(4x4) kernel does (same 5e-4 bound; measured: the same 1e-6) and within 1e-5 of that kernel.  Covers
    whole and odd region counts, regions spanning two patches, the Cin % 8 == 4 tail and 1 - 3 output blocks.  Layers with
    Cout % 64 != 0 stay on the fp32 kernel (checked through the launch profile)."""
    rng = np.random.default_rng(cin * 11 + cout)
    H, W = hw
    cfg = {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, H, W, cin]},
         'inbound_nodes': []},
        {'class_name': 'Conv2D', 'name': 'c', 'config': {'name': 'c', 'filters': cout, 'kernel_size': [3, 3],
                                                         'strides': [1, 1], 'padding': 'same', 'activation': 'relu',
                                                         'use_bias': True}, 'inbound_nodes': [[['in', 0, 0, {}]]]}],
        'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}
    weights = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32),
                     rng.normal(size=cout).astype(np.float32)]}
    x = rng.integers(0, 256, size=(n, H, W, cin), dtype=np.uint8)
    want = oracle_unet.forward(cfg, weights, x)
    try:
        gpu.set_option('winograd', 2)
        w4, _ = _run(gpu, cfg, weights, x, fuse=True)
        gpu.set_option('winograd', 3)
        gpu.set_kernel_profiling(True)
        ws = gpu.forward_patches(x)
        kinds = [r['kind'] & 0xff for r in gpu.conv_launch_profile()]
        gpu.set_kernel_profiling(False)
    finally:
        gpu.set_option('winograd', 2)
    assert kinds and set(kinds) == {5}, kinds                    # conv_wino4s_kernel ran (one launch per window lane)
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(ws - want).max() < 5e-4 * scale, np.abs(ws - want).max()
    assert np.abs(ws - w4).max() < 1e-5 * scale, np.abs(ws - w4).max()


def test_base64_unet_split_mode_with_views_pool_head_and_fallback(gpu):
    """Base-64 U-Net (depth 2) under winograd = 3: strided channel views of the skip connections, the fused 2x2 max-pool and the
    fused 1x1 head run through conv_wino4s_kernel's output stage; the same model at base 32 has layers with 32 output channels,
    which fall back to the fp32 F(4x4) kernel.  Probabilities within 1e-3 of the oracle and 1e-5 of the fp32-MFMA path."""
    for base in (64, 32):
        cfg = synth.unet_config(base=base, depth=2)
        weights = synth.unet_weights(cfg, seed=base)
        x = _patches(2, seed=base)
        want = oracle_unet.forward(cfg, weights, x)
        try:
            gpu.set_option('winograd', 2)
            ref, _ = _run(gpu, cfg, weights, x, fuse=True)
            gpu.set_option('winograd', 3)
            gpu.set_kernel_profiling(True)
            got = gpu.forward_patches(x)
            kinds = [r['kind'] & 0xff for r in gpu.conv_launch_profile()]
            gpu.set_kernel_profiling(False)
            gpu.set_option('fuse_pool', 0)
            gpu.set_option('fuse_head', 0)
            unfused = gpu.forward_patches(x)
        finally:
            gpu.set_option('winograd', 2)
            gpu.set_option('fuse_pool', 1)
            gpu.set_option('fuse_head', 1)
        assert 5 in kinds and (base == 64 or 2 in kinds), (base, kinds)
        assert np.abs(got - want).max() < TOL, np.abs(got - want).max()
        assert np.abs(got - ref).max() < 1e-5, np.abs(got - ref).max()
        assert np.abs(got - unfused).max() < 1e-5, np.abs(got - unfused).max()
        np.testing.assert_allclose(got.sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize('base,bn', [(16, False), (32, True), (64, False)])
def test_upsample_conv2x2_decoder_runs_as_one_transposed_convolution(gpu, base, bn):
    """Round 6: UpSampling2D(2, nearest) -> Conv2D(2x2, 'same') is lowered to ONE 3x3 / stride-2 transposed
    convolution with pre-summed taps on the un-upsampled tensor (keras_plan pass 2).  The fused plan must contain no UpSampling2D,
    equal the layer-by-layer plan (fuse=False: real up-sampling + a real 2x2 convolution) to float32 rounding and the oracle -
    which evaluates the ORIGINAL graph - within 1e-3."""
    cfg = synth.unet_config(base=base, up='upsample', batchnorm=bn, depth=3)
    weights = synth.unet_weights(cfg, seed=base + 1)
    x = _patches(2, seed=base)
    want = oracle_unet.forward(cfg, weights, x)
    fused, plan = _run(gpu, cfg, weights, x, fuse=True)
    assert not any(o['op'] == keras_plan.OP_UPSAMPLE for o in plan.ops)
    assert sum(1 for o in plan.ops if o['op'] == keras_plan.OP_CONVT and (o['kh'], o['stride'], o['pad_top'], o['pad_left']) == (3, 2, 1, 1)) == 3
    plain, plan0 = _run(gpu, cfg, weights, x, fuse=False)
    assert sum(1 for o in plan0.ops if o['op'] == keras_plan.OP_UPSAMPLE) == 3
    assert np.abs(fused - want).max() < TOL, np.abs(fused - want).max()
    assert np.abs(fused - plain).max() < 2e-5, np.abs(fused - plain).max()
    np.testing.assert_allclose(fused.sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize('cin,cout,hw,n', [(64, 32, (16, 16), 3), (128, 64, (32, 32), 2), (72, 96, (16, 32), 2), (1024, 512, (16, 16), 1), (20, 32, (16, 16), 2)])
def test_up_convolution_split_bf16x3_matches_fp32_kernel_and_oracle(gpu, cin, cout, hw, n):
    """Round 6 (option winograd = 3): Conv2DTranspose 2x2 / stride 2 as a one-tap GEMM on the bf16 matrix pipe with 3-way split operands
    (convs_kernel: the A tile is split once per workgroup while it is staged into LDS).  Same bound as the fp32 kernel against the
    oracle, 1e-5 against that kernel; covers K tails (Cin % 16 = 8 / 4), 1 - 16 column blocks, both tile shapes."""
    rng = np.random.default_rng(cin * 13 + cout)
    H, W = hw
    cfg = {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, H, W, cin]}, 'inbound_nodes': []},
        {'class_name': 'Conv2DTranspose', 'name': 'c', 'config': {'name': 'c', 'filters': cout, 'kernel_size': [2, 2], 'strides': [2, 2],
                                                                  'padding': 'same', 'activation': 'relu', 'use_bias': True},
         'inbound_nodes': [[['in', 0, 0, {}]]]}], 'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}
    weights = {'c': [(rng.normal(size=(2, 2, cout, cin)) / np.sqrt(cin) / 64).astype(np.float32), rng.normal(size=cout).astype(np.float32)]}
    x = rng.integers(0, 256, size=(n, H, W, cin), dtype=np.uint8)
    want = oracle_unet.forward(cfg, weights, x)
    try:
        gpu.set_option('winograd', 2)
        ref, _ = _run(gpu, cfg, weights, x, fuse=True)
        gpu.set_option('winograd', 3)
        gpu.set_kernel_profiling(True)
        got = gpu.forward_patches(x)
        kinds = [r['kind'] & 0xff for r in gpu.conv_launch_profile()]
        gpu.set_kernel_profiling(False)
    finally:
        gpu.set_option('winograd', 2)
    assert kinds and set(kinds) == {6}, kinds
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(got - want).max() < 1e-4 * scale, np.abs(got - want).max()
    assert np.abs(got - ref).max() < 1e-5 * scale, np.abs(got - ref).max()


@pytest.mark.parametrize('act,use_bias', [('linear', True), ('sigmoid', False), ('tanh', True), ('elu', True)])
def test_winograd_f4x4_activations_and_no_bias(gpu, act, use_bias):
    """The F(4x4) output stage applies the layer's own activation and tolerates a missing bias."""
    rng = np.random.default_rng(17)
    cin, cout, H, W, n = 16, 64, 32, 32, 2
    cfg = {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, H, W, cin]},
         'inbound_nodes': []},
        {'class_name': 'Conv2D', 'name': 'c', 'config': {'name': 'c', 'filters': cout, 'kernel_size': [3, 3],
                                                         'strides': [1, 1], 'padding': 'same', 'activation': act,
                                                         'use_bias': use_bias}, 'inbound_nodes': [[['in', 0, 0, {}]]]}],
        'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}
    w = [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32)]
    if use_bias:
        w.append(rng.normal(size=cout).astype(np.float32))
    x = rng.integers(0, 256, size=(n, H, W, cin), dtype=np.uint8)
    want = oracle_unet.forward(cfg, {'c': w}, x)
    got, _ = _run(gpu, cfg, {'c': w}, x, fuse=True)
    assert np.abs(got - want).max() < 5e-4 * max(1.0, float(np.abs(want).max())), np.abs(got - want).max()


@pytest.mark.parametrize('n_convs,ch', [(3, 64), (4, 64), (3, 16), (4, 16), (3, 32), (4, 32)])
def test_fused_head_on_plain_conv_stack(gpu, n_convs, ch):
    """in -> conv64 x n -> 1x1 softmax head.  The liveness allocator used to hand the head the buffer the last
    3x3 convolution READS; with the head fused into that convolution's output stage this was a race.  Fused and unfused
    runs must agree (and match the oracle).  64 channels: conv_wino4_kernel's fused head; 16 / 32: conv_wino16_kernel's."""
    cfg = synth.conv_stack_config(n_convs, ch=ch)
    weights = synth.unet_weights(cfg, seed=11)
    x = _patches(3, seed=2)
    want = oracle_unet.forward(cfg, weights, x)
    try:
        gpu.set_option('fuse_head', 1)
        fused, plan = _run(gpu, cfg, weights, x, fuse=True)
        gpu.set_option('fuse_head', 0)
        unfused = gpu.forward_patches(x)
    finally:
        gpu.set_option('fuse_head', 1)
    assert np.abs(fused - want).max() < TOL
    assert np.abs(fused - unfused).max() < 1e-5


@pytest.mark.parametrize('cin,cout,hw,n', [(16, 16, (256, 256), 40),      # 16 tiles per workgroup walk (one segment per tile row)
                                           (32, 32, (64, 208), 80),       # 13 tiles per row: segments of 8 + 5
                                           (24, 16, (72, 208), 114),      # one 13-tile segment, Cin = 3 chunks
                                           (8, 32, (40, 56), 3),          # small launch: one tile per workgroup, overhanging tiles
                                           (12, 20, (37, 53), 2)])        # Cin % 8 == 4, Cout % 4 == 0 only, odd extents
def test_winograd_resident_filter_kernel(gpu, cin, cout, hw, n):
    """Narrow layers (Cin <= 32, Cout <= 32) under F(2x2): the filter-resident kernel (filters in registers, a workgroup walks
    several tiles of a tile row) does the arithmetic of the streaming kernel in the same order - identical results - and
    agrees with the oracle."""
    rng = np.random.default_rng(cin * 11 + cout)
    H, W = hw
    cfg = {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, H, W, cin]},
         'inbound_nodes': []},
        {'class_name': 'Conv2D', 'name': 'c', 'config': {'name': 'c', 'filters': cout, 'kernel_size': [3, 3],
                                                         'strides': [1, 1], 'padding': 'same', 'activation': 'relu',
                                                         'use_bias': True}, 'inbound_nodes': [[['in', 0, 0, {}]]]}],
        'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}
    weights = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32),
                     rng.normal(size=cout).astype(np.float32)]}
    x = rng.integers(0, 256, size=(n, H, W, cin), dtype=np.uint8)
    try:
        gpu.set_option('winograd', 1)
        gpu.set_option('wino_resident', 1)
        gpu.set_option('wino16', 0)
        res, _ = _run(gpu, cfg, weights, x, fuse=True)
        gpu.set_option('wino_resident', 0)
        stream = gpu.forward_patches(x)
    finally:
        gpu.set_option('winograd', 2)
        gpu.set_option('wino_resident', 1)
        gpu.set_option('wino16', 1)
    assert np.array_equal(res, stream), float(np.abs(res - stream).max())
    k = min(n, 4)                                              # the oracle on a few patches from both ends of the batch
    sel = np.r_[0:k // 2, n - (k - k // 2):n]
    want = oracle_unet.forward(cfg, weights, x[sel])
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(res[sel] - want).max() < 2e-4 * scale, np.abs(res[sel] - want).max()


@pytest.mark.parametrize('cin,cout,hw,n', [(16, 16, (256, 256), 64),      # walks of 8 blocks (16 x 32 pixels each)
                                           (16, 32, (48, 304), 171),      # 10 blocks per strip: walks of 8 + 2, last block partial
                                           (32, 16, (64, 208), 5),        # small launch: one block per workgroup, 6.5 blocks per strip
                                           (32, 32, (16, 96), 512),       # walks of 2 + 1 blocks, two 16-channel stages per block
                                           (32, 32, (40, 56), 3)])        # partial blocks in both directions
def test_winograd_wino16_kernel(gpu, cin, cout, hw, n):
    """Narrow layers (16 / 32 input and output channels) under F(2x2): conv_wino16_kernel (16x16x4 MFMAs with the filter as
    the A operand, register output stage, LDS-DMA halo double buffer) against the 32-wide F(2x2) kernel and the oracle."""
    rng = np.random.default_rng(cin * 13 + cout)
    H, W = hw
    cfg = {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, H, W, cin]},
         'inbound_nodes': []},
        {'class_name': 'Conv2D', 'name': 'c', 'config': {'name': 'c', 'filters': cout, 'kernel_size': [3, 3],
                                                         'strides': [1, 1], 'padding': 'same', 'activation': 'relu',
                                                         'use_bias': True}, 'inbound_nodes': [[['in', 0, 0, {}]]]}],
        'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}
    weights = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32),
                     rng.normal(size=cout).astype(np.float32)]}
    x = rng.integers(0, 256, size=(n, H, W, cin), dtype=np.uint8)
    try:
        gpu.set_option('winograd', 1)
        gpu.set_option('wino16', 1)
        w16, _ = _run(gpu, cfg, weights, x, fuse=True)
        gpu.set_option('wino16', 0)
        ref = gpu.forward_patches(x)
    finally:
        gpu.set_option('winograd', 2)
        gpu.set_option('wino16', 1)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(w16 - ref).max() < 1e-4 * scale, float(np.abs(w16 - ref).max())
    k = min(n, 4)                                              # the oracle on a few patches from both ends of the batch
    sel = np.r_[0:k // 2, n - (k - k // 2):n]
    want = oracle_unet.forward(cfg, weights, x[sel])
    assert np.abs(w16[sel] - want).max() < 2e-4 * scale, np.abs(w16[sel] - want).max()


@pytest.mark.parametrize('cin,hw,n', [(64, (32, 32), 3), (72, (16, 48), 5), (128, (48, 16), 2), (68, (16, 16), 1)])
def test_winograd_f4x4_split_k_for_32_output_channels(gpu, cin, hw, n):
    """A layer with exactly 32 output channels under F(4x4): the two channel-half waves of a transform row split the 8 input
    channels of a group (option wino4_split) instead of multiplying the zero-padded upper half of a 64-channel block; both
    modes against each other and the oracle (odd number of 16x16 regions, Cin % 8 == 4 tail, one K group more or less)."""
    rng = np.random.default_rng(cin * 5 + 1)
    H, W = hw
    cout = 32
    cfg = {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, H, W, cin]},
         'inbound_nodes': []},
        {'class_name': 'Conv2D', 'name': 'c', 'config': {'name': 'c', 'filters': cout, 'kernel_size': [3, 3],
                                                         'strides': [1, 1], 'padding': 'same', 'activation': 'relu',
                                                         'use_bias': True}, 'inbound_nodes': [[['in', 0, 0, {}]]]}],
        'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}
    weights = {'c': [(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin) / 64).astype(np.float32),
                     rng.normal(size=cout).astype(np.float32)]}
    x = rng.integers(0, 256, size=(n, H, W, cin), dtype=np.uint8)
    want = oracle_unet.forward(cfg, weights, x)
    try:
        gpu.set_option('winograd', 2)
        gpu.set_option('wino4_split', 1)
        split, _ = _run(gpu, cfg, weights, x, fuse=True)
        gpu.set_option('wino4_split', 0)
        padded = gpu.forward_patches(x)
    finally:
        gpu.set_option('wino4_split', 1)
    scale = max(1.0, float(np.abs(want).max()))
    assert np.abs(split - padded).max() < 1e-4 * scale, float(np.abs(split - padded).max())
    assert np.abs(split - want).max() < 5e-4 * scale, np.abs(split - want).max()


def _one_layer(cls, cin, hw, **cfg):
    return {'class_name': 'Functional', 'config': {'name': 'm', 'layers': [
        {'class_name': 'InputLayer', 'name': 'in', 'config': {'name': 'in', 'batch_input_shape': [None, hw[0], hw[1], cin]}, 'inbound_nodes': []},
        {'class_name': cls, 'name': 'c', 'config': dict(cfg, name='c'), 'inbound_nodes': [[['in', 0, 0, {}]]]}],
        'input_layers': [['in', 0, 0]], 'output_layers': [['c', 0, 0]]}}


def _mfma_ops(gpu, x):
    """Plan operators that ran on an MFMA kernel in a profiled forward pass."""
    gpu.set_kernel_profiling(True)
    try:
        gpu.forward_patches(x)
        return {r['op'] for r in gpu.conv_launch_profile()}
    finally:
        gpu.set_kernel_profiling(False)


@pytest.mark.parametrize('cin,cout,k,padding,hw', [(32, 16, 3, 'same', (32, 48)), (64, 64, 3, 'same', (16, 16)), (16, 8, 4, 'same', (24, 40)),
                                                   (24, 40, 3, 'valid', (17, 23)), (128, 128, 4, 'valid', (8, 8)), (8, 200, 3, 'same', (9, 33))])
def test_transposed_conv_kernel_larger_than_stride_on_mfma(gpu, cin, cout, k, padding, hw):
    """Conv2DTranspose k x k / stride 2 with k in {3, 4} (NuSeT's up-sampler, src/model_layers/models.py:78-80):
    four sub-pixel convolutions as one 2x2-tap MFMA convolution with a scatter epilogue - against the oracle, and NOT on the
    generic kernel."""
    rng = np.random.default_rng(cin + 7 * cout + k)
    cfg = _one_layer('Conv2DTranspose', cin, hw, filters=cout, kernel_size=[k, k], strides=[2, 2], padding=padding, activation='relu', use_bias=True)
    weights = {'c': [(rng.normal(size=(k, k, cout, cin)) / np.sqrt(cin) / 64).astype(np.float32), rng.normal(size=cout).astype(np.float32)]}
    x = rng.integers(0, 256, size=(3, hw[0], hw[1], cin), dtype=np.uint8)
    got, plan = _run(gpu, cfg, weights, x, fuse=True)
    want = oracle_unet.forward(cfg, weights, x)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < TOL * max(1.0, np.abs(want).max()), np.abs(got - want).max()
    assert _mfma_ops(gpu, x) == {k_ for k_, o in enumerate(plan.ops) if o['op'] == keras_plan.OP_CONVT}


@pytest.mark.parametrize('cin,cout,k,padding,hw', [(16, 32, 3, 'same', (64, 96)), (32, 64, 1, 'valid', (33, 47)), (8, 16, 2, 'valid', (32, 32)),
                                                   (64, 48, 3, 'valid', (31, 45)), (24, 24, 3, 'same', (15, 20))])
def test_strided_conv_on_mfma(gpu, cin, cout, k, padding, hw):
    """Conv2D with stride 2 (classifier stems, down-sampling convolutions) on the direct MFMA kernel's strided halo gather."""
    rng = np.random.default_rng(cin + 3 * cout + k)
    cfg = _one_layer('Conv2D', cin, hw, filters=cout, kernel_size=[k, k], strides=[2, 2], padding=padding, activation='relu', use_bias=True)
    weights = {'c': [(rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin) / 64).astype(np.float32), rng.normal(size=cout).astype(np.float32)]}
    x = rng.integers(0, 256, size=(3, hw[0], hw[1], cin), dtype=np.uint8)
    got, plan = _run(gpu, cfg, weights, x, fuse=True)
    want = oracle_unet.forward(cfg, weights, x)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < TOL * max(1.0, np.abs(want).max()), np.abs(got - want).max()
    assert _mfma_ops(gpu, x) == {0}


@pytest.mark.parametrize('cin,cout,k', [(64, 2, 3), (32, 3, 3), (16, 5, 2), (128, 1, 3)])
def test_conv_to_a_few_channels_on_mfma(gpu, cin, cout, k):
    """A 3x3 / 2x2 convolution to fewer than 16 channels (NuSeT's 'final' layer, src/model_layers/models.py:134) runs on the MFMA
    kernel with a mostly empty column tile instead of the scalar fall-back."""
    rng = np.random.default_rng(cin + cout)
    cfg = _one_layer('Conv2D', cin, (48, 64), filters=cout, kernel_size=[k, k], strides=[1, 1], padding='same', activation='linear', use_bias=False)
    weights = {'c': [(rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin) / 64).astype(np.float32)]}
    x = rng.integers(0, 256, size=(2, 48, 64, cin), dtype=np.uint8)
    got, plan = _run(gpu, cfg, weights, x, fuse=True)
    want = oracle_unet.forward(cfg, weights, x)
    assert np.abs(got - want).max() < TOL * max(1.0, np.abs(want).max()), np.abs(got - want).max()
    assert _mfma_ops(gpu, x) == {0}


@pytest.mark.parametrize('base,up', [(16, 'transpose3'), (32, 'transpose4')])
def test_unet_with_3x3_and_4x4_up_convolutions(gpu, base, up):
    """A U-Net whose decoder up-samples with 3x3 / 4x4 stride-2 transposed convolutions: probabilities vs the oracle, and every
    convolution except the 1-channel first layer and the 1x1 head on an MFMA kernel (none on conv_generic / convt_generic)."""
    cfg = synth.unet_config(base=base, depth=3, up=up)
    weights = synth.unet_weights(cfg, seed=3)
    x = _patches(2, seed=5)
    got, plan = _run(gpu, cfg, weights, x, fuse=True)
    want = oracle_unet.forward(cfg, weights, x)
    assert np.abs(got - want).max() < TOL, np.abs(got - want).max()
    convs = [k for k, o in enumerate(plan.ops) if o['op'] in (keras_plan.OP_CONV, keras_plan.OP_CONVT)]
    ran = _mfma_ops(gpu, x)
    missing = [k for k in convs[1:-1] if k not in ran]
    # (a 1x1 head or 2x2 pool fused into the previous convolution's output stage is not launched at all)
    assert not missing, [(k, plan.ops[k]) for k in missing]


def test_nuset_shaped_unet_runs_on_mfma_kernels(gpu):
    """SURVEY 8(f)4 / VERDICT r02 #6: the architecture of NuSeT's U-Net (src/model_layers/models.py:5-136: 3x3 stride-2 transposed
    convolutions, a first up-sampler without skip, a bias-free 3x3 'final' layer to 2 feature maps) with seeded weights - output
    within 1e-3 of the oracle and every convolution but the 1-channel first layer on an MFMA kernel."""
    cfg = synth.nuset_unet_config(base=16)
    weights = synth.unet_weights(cfg, seed=4)
    x = _patches(2, seed=9)
    got, plan = _run(gpu, cfg, weights, x, fuse=True)
    want = oracle_unet.forward(cfg, weights, x)
    assert got.shape == want.shape == (2, 256, 256, 2)
    assert np.abs(got - want).max() < TOL * max(1.0, np.abs(want).max()), np.abs(got - want).max()
    convs = [k for k, o in enumerate(plan.ops) if o['op'] in (keras_plan.OP_CONV, keras_plan.OP_CONVT)]
    ran = _mfma_ops(gpu, x)
    missing = [k for k in convs[1:] if k not in ran]
    assert not missing, [(k, plan.ops[k]) for k in missing]


def _L(cls, name, inbound, **cfg):
    return {'class_name': cls, 'name': name, 'config': dict(cfg, name=name),
            'inbound_nodes': [[[i, 0, 0, {}] for i in inbound]] if inbound else []}


@pytest.mark.parametrize('n,hw,pool,ch', [(3, 256, True, 16), (2, 64, True, 16), (5, 96, False, 16), (1, 48, False, 16), (2, 256, True, 32), (3, 80, False, 32)])
def test_first_layer_fused_into_the_wino16_halo(gpu, n, hw, pool, ch):
    """Round 5: the network's first layer (3x3, 1 -> 16 / 32 channels) computed by the 16 -> 16 / 32 -> 32
    convolution behind it, on the matrix cores, into that kernel's own halo buffer (conv_wino16_kernel FIRST; option fuse_first).  Same
    arithmetic per output (9 products summed in float32, in MFMA order instead of the scalar kernel's): equal to the unfused
    plan within 1e-5 and to the oracle within 1e-3; extents that are not multiples of the 16 x 32 block, images at the batch
    ends, a fused max-pool behind it; and the fusion really happens (launch profile bit 0x400)."""
    rng = np.random.default_rng(n * 1000 + hw + ch)
    layers = [_L('InputLayer', 'in', [], batch_input_shape=[None, hw, hw, 1]),
              _L('Conv2D', 'c0', ['in'], filters=ch, kernel_size=[3, 3], strides=[1, 1], padding='same', activation='relu', use_bias=True),
              _L('Conv2D', 'c1', ['c0'], filters=ch, kernel_size=[3, 3], strides=[1, 1], padding='same', activation='relu', use_bias=True)]
    last = 'c1'
    if pool:
        layers.append(_L('MaxPooling2D', 'mp', ['c1'], pool_size=[2, 2], strides=[2, 2], padding='valid'))
        layers.append(_L('UpSampling2D', 'up', ['mp'], size=[2, 2], interpolation='nearest'))
        layers.append(_L('Concatenate', 'cat', ['up', 'c1'], axis=-1))
        last = 'cat'
    layers.append(_L('Conv2D', 'head', [last], filters=4, kernel_size=[1, 1], strides=[1, 1], padding='same', activation='softmax', use_bias=True))
    cfg = {'class_name': 'Functional', 'config': {'name': 'm', 'layers': layers, 'input_layers': [['in', 0, 0]], 'output_layers': [['head', 0, 0]]}}
    cl = 2 * ch if pool else ch
    w = {'c0': [(rng.normal(size=(3, 3, 1, ch)) / 300).astype(np.float32), (rng.normal(size=ch) * .1).astype(np.float32)],
         'c1': [(rng.normal(size=(3, 3, ch, ch)) / np.sqrt(9 * ch)).astype(np.float32), (rng.normal(size=ch) * .1).astype(np.float32)],
         'head': [(rng.normal(size=(1, 1, cl, 4)) / 4).astype(np.float32), (rng.normal(size=4) * .1).astype(np.float32)]}
    x = rng.integers(0, 256, size=(n, hw, hw, 1), dtype=np.uint8)
    want = oracle_unet.forward(cfg, w, x)
    gpu.load_plan(keras_plan.build_plan(cfg, w))
    try:
        gpu.set_option('fuse_first', 0)
        plain = gpu.forward_patches(x)
        gpu.set_option('fuse_first', 1)
        gpu.set_kernel_profiling(True)
        fused = gpu.forward_patches(x)
        recs = gpu.conv_launch_profile()
        gpu.set_kernel_profiling(False)
    finally:
        gpu.set_option('fuse_first', 1)
    assert any(r['kind'] & 0x400 for r in recs), recs               # the first layer rode on the second one's launch
    assert np.abs(fused - plain).max() < 1e-5 and np.abs(fused - want).max() < 1e-3

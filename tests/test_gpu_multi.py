"""More than one rank on real GPUs (skipped on 1-GPU boxes): `bench.py --gpus 2` starts its own two ranks, RCCL carries
the all-gather of the per-image records, and the JSON line reports n_gpus == 2."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_n_gpus() < 2, reason='needs 2 GPUs')
def test_bench_self_launches_two_ranks_over_rccl():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                          '--images', '2', '--base', '16'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    res = json.loads(line)
    assert res['n_gpus'] == 2 and res['scaling'] == 'weak'
    assert res['config']['images_per_gpu_per_step'] == 2 and res['value'] > 0


@pytest.mark.skipif(_n_gpus() < 2, reason='needs 2 GPUs')
def test_two_handles_in_one_process_on_two_gpus():
    """One process may drive one handle per GPU (per-device kernel attributes: ADVICE r01)."""
    import numpy as np
    from ecseg_amd import keras_plan, synth
    from ecseg_amd._lib import Handle
    cfg = synth.unet_config(base=64, depth=1)
    w = synth.unet_weights(cfg, seed=1)
    x = np.random.default_rng(0).integers(0, 256, size=(2, 256, 256, 1), dtype=np.uint8)
    outs = []
    for dev in (0, 1):
        h = Handle(dev)
        try:
            h.load_plan(keras_plan.build_plan(cfg, w))
            outs.append(h.forward_patches(x))
        finally:
            h.close()
    assert np.array_equal(outs[0], outs[1])

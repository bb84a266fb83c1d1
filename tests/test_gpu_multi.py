"""More than one rank on real GPUs (skipped on 1-GPU boxes): `bench.py --gpus 2` starts its own two ranks, RCCL carries
the all-gather of the per-image records, and the JSON line reports n_gpus == 2.  On any box: the C-ABI collective
(ecseg_comm_* / ecseg_allgather_records*, csrc/comm.hip) with a one-rank communicator, and bench.py's use of it under a
one-rank process group (the code path the multi-GPU runs take)."""
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


def test_c_abi_record_allgather_one_rank():
    """ecseg_comm_unique_id / ecseg_comm_create / ecseg_allgather_records (host and device buffers) on a 1-rank RCCL
    communicator: the gathered block is the rank's own block, padding rows included."""
    import numpy as np
    import torch
    from ecseg_amd import dist as edist
    from ecseg_amd._lib import Comm
    uid = Comm.unique_id()
    assert len(uid) == 128
    c = Comm(uid, 0, 1, 0)
    try:
        rec = edist.make_records(5, 3, 4, n_ec=[7, 8, 9], status=[0, 1, 0])             # 3 real rows + 1 padding row
        out = c.allgather_records(rec)
        assert out.shape == (4, 16) and np.array_equal(out, rec)
        d = torch.from_numpy(rec).to('cuda:0')
        g = torch.zeros_like(d)
        c.allgather_records_dev(d.data_ptr(), 4, g.data_ptr())
        assert torch.equal(g, d)
        s = torch.cuda.Stream()
        g.zero_()
        with torch.cuda.stream(s):
            c.allgather_records_dev(d.data_ptr(), 4, g.data_ptr(), stream=s.cuda_stream)
        s.synchronize()
        assert torch.equal(g, d)
        assert np.array_equal(edist.compact_records(out)[:, edist.F_NEC], [7, 8, 9])
    finally:
        c.close()


def test_bench_uses_the_c_abi_collective_under_a_process_group():
    """bench.py under torch.distributed.run with ONE rank and a forced process group: the C-ABI all-gather is set up, checked
    against torch's and used in the timed steps - the code the N > 1 runs execute, on the one GPU this box has."""
    env = dict(os.environ, ECSEG_FORCE_PROCESS_GROUP='1', ECSEG_BENCH_FORCE_COMM='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
                          '--nproc-per-node', '1', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                          '--images', '2', '--base', '16', '--no-cpu-baseline', '--no-narrow', '--no-host-inclusive'],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert res['n_gpus'] == 1 and res['value'] > 0
    assert res['config']['allgather'].startswith('ecseg_allgather_records_dev'), res['config']['allgather']
    assert res['stage_ms_per_image']['allgather'] > 0                       # the exchange is reported as its own stage


def test_native_transport_one_rank(tmp_path):
    """dist.native_init / gather_all / native_close: the library's RCCL communicator behind the `make metaseg` exchange
    (ECSEG_DIST=native), with the one rank this box has."""
    import numpy as np
    from ecseg_amd import dist as edist
    path = str(tmp_path / 'rdzv')
    assert edist.native_init(0, 1, 0, path) == (0, 1)
    try:
        assert len(open(path, 'rb').read()) == 128
        rec = edist.make_records(10, 3, 5, n_ec=[1, 2, 3], status=[0, 2, 0])
        out = edist.gather_all(rec)
        assert out.shape == (3, 16) and np.array_equal(out[:, edist.F_NEC], [1, 2, 3]) and out[1, edist.F_STATUS] == 2
    finally:
        edist.native_close(path, 0)
    assert not os.path.exists(path) and edist._native is None


def test_collective_is_bounded_when_a_peer_never_arrives(tmp_path):
    """a library user binding ecseg_comm_create directly must not hang for ever when a peer is missing.  A
    fresh child process creates rank 0 of a world-2 communicator on the one GPU with nobody playing rank 1 and
    ECSEG_COMM_TIMEOUT_S=4: the call returns ECSEG_E_HIP with a message naming the timeout, well inside the test's own limit."""
    import time
    code = (
        "import sys, time\n"
        "sys.path.insert(0, %r)\n"
        "from ecseg_amd._lib import Comm, EcsegError\n"
        "uid = Comm.unique_id()\n"
        "t0 = time.time()\n"
        "try:\n"
        "    Comm(uid, 0, 2, 0)\n"
        "    print('CREATED')\n"
        "except EcsegError as e:\n"
        "    print('ERROR %%d %%.1f %%s' %% (e.code, time.time() - t0, e))\n"
    ) % ROOT
    t0 = time.time()
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, ECSEG_COMM_TIMEOUT_S='4'))
    took = time.time() - t0
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith(('ERROR', 'CREATED'))][-1]
    assert line.startswith('ERROR -2 '), line
    assert 'ECSEG_COMM_TIMEOUT_S' in line and 'aborted' in line
    assert 3.0 < float(line.split()[2]) < 30.0 and took < 90.0, (line, took)

"""The sharded `make metaseg` path on the CPU: two gloo ranks drive ``ecseg_amd.metaseg.run`` with a stub model (no GPU)
over 7 and 8 generated TIFF files, one of them corrupt - per-rank outputs, padded records, statuses and the rank-0 CSV must
equal a 1-rank run over the same files (reference: the serial loop of src/metaseg.py:42-57).  Also: resume after
a killed run (truncated .npy), atomic outputs and the out-of-memory retry of the device part."""
import os
import shutil
import socket
import time

import numpy as np
import pytest

from ecseg_amd import dist as edist
from ecseg_amd import image_io, metaseg
from ecseg_amd._lib import E_NOMEM, EcsegError
from ecseg_amd.utils import get_imgs
from oracle import postproc

H, W = 96, 128


class StubHandle:
    """What metaseg.run needs of a Handle, computed on the CPU: deterministic functions of the pixels."""
    device = 0

    def __init__(self, fail_above=None, poison=None):
        self.images_per_group = 0
        self.fail_above = fail_above           # simulate E_NOMEM for launch groups / batches above this many images
        self.poison = poison                   # simulate a non-memory device error for any batch holding this pixel value at [0, 0]
        self.calls = []

    def set_images_per_group(self, n):
        self.images_per_group = int(n)

    def preprocess(self, imgs):
        a = np.asarray(imgs)
        gray = a[..., 2] if a.ndim == 4 else a
        return np.ascontiguousarray(gray, np.uint8), np.zeros(len(a), np.int32)

    def count_cc(self, masks):
        m = np.asarray(masks).astype(bool)
        return np.array([postproc.count_cc(x)[0] for x in m], np.int32), np.array([int(x.sum()) for x in m], np.int64)


class StubModel:
    def __init__(self, **kw):
        self.handle = StubHandle(**kw)

    def segment(self, gray):
        h = self.handle
        grp = h.images_per_group if h.images_per_group > 0 else len(gray)
        h.calls.append((len(gray), h.images_per_group))
        if h.fail_above is not None and min(grp, len(gray)) > h.fail_above:
            e = EcsegError('simulated out of memory')
            e.code = E_NOMEM
            raise e
        if h.poison is not None and (np.asarray(gray)[:, 0, 0] == h.poison).any():
            e = EcsegError('simulated device fault')
            e.code = -5
            raise e
        post = (np.asarray(gray) >> 6).astype(np.uint8)                    # labels 0..3
        return post, np.array([postproc.count_cc(p == 3)[0] for p in post], np.int32)


def make_inputs(folder, n, corrupt=None):
    from PIL import Image
    os.makedirs(folder, exist_ok=True)
    rng = np.random.default_rng(n)
    for i in range(n):
        img = np.zeros((H, W, 3), np.uint8)
        img[..., 2] = rng.integers(0, 64, (H, W))
        for _ in range(3 + i):                                               # a few bright blobs -> class 3 components
            y, x = rng.integers(4, H - 8), rng.integers(4, W - 8)
            img[y:y + 3, x:x + 4, 2] = 250
        p = os.path.join(folder, 'img%02d.tif' % i)
        if i == corrupt:
            open(p, 'wb').write(b'II*\x00' + bytes(rng.integers(0, 255, 300, dtype=np.uint8)))
        else:
            Image.fromarray(img).save(p, compression='tiff_lzw')
    for sub in ('dapi', 'labels'):
        os.makedirs(os.path.join(folder, sub), exist_ok=True)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, folder, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    import torch
    r, w = edist.init_process_group('gloo')
    paths = get_imgs(folder)
    rec = metaseg.run(folder, StubModel(), paths, r, w, batch_images=3, io_threads=2, log=lambda *a: None)
    failed = metaseg.finish(folder, paths, rec, r, log=lambda *a: None)
    q.put((rank, rec.tolist(), len(failed)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _snapshot(folder):
    out = {}
    for sub in ('dapi', 'labels', ''):
        d = os.path.join(folder, sub)
        for f in sorted(os.listdir(d)):
            p = os.path.join(d, f)
            if os.path.isfile(p) and not f.startswith('img') or sub:
                out[os.path.join(sub, f)] = open(p, 'rb').read()
    return out


@pytest.mark.parametrize('n_images', [7, 8])
def test_sharded_run_equals_single_rank(tmp_path, n_images):
    import torch.multiprocessing as mp
    one, two = str(tmp_path / 'one'), str(tmp_path / 'two')
    make_inputs(one, n_images, corrupt=2)
    shutil.copytree(one, two)
    # 1 rank
    paths = get_imgs(one)
    rec1 = metaseg.run(one, StubModel(), paths, 0, 1, batch_images=3, io_threads=2, log=lambda *a: None)
    failed1 = metaseg.finish(one, paths, rec1, 0, log=lambda *a: None)
    assert len(rec1) == n_images and [int(r[edist.F_STATUS]) for r in rec1] == [1 if i == 2 else 0 for i in range(n_images)]
    assert len(failed1) == 1
    # 2 gloo ranks
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, two, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rec, n_failed in res:                                          # every rank holds the whole job's records
        assert np.array_equal(np.array(rec), rec1), 'records of rank %d differ from the 1-rank run' % rank
        assert n_failed == 1
    a, b = _snapshot(one), _snapshot(two)
    assert sorted(a) == sorted(b)
    for k in a:
        assert a[k] == b[k], k
    csv = a['ec_quantification.csv'].decode()
    assert csv.count('\n') == n_images and 'img02.tif' not in csv and a['ec_quantifications.csv'] == a['ec_quantification.csv']
    # outputs of rank 1's shard exist and hold what the stub computed
    stem = 'img%02d' % (n_images - 1)
    lab = np.load(os.path.join(two, 'labels', stem + '.npy'))
    assert lab.dtype == np.int64 and lab.shape == (H, W)
    gray = image_io.imread(os.path.join(two, stem + '.tif'))[..., 2]
    assert np.array_equal(lab, gray >> 6)
    assert np.array_equal(image_io.imread(os.path.join(two, 'dapi', stem + '.tif')), ~gray)
    assert not [f for d in ('dapi', 'labels', '') for f in os.listdir(os.path.join(two, d)) if '.tmp' in f]


def test_resume_survives_a_truncated_npy(tmp_path):
    """a run killed while writing leaves a truncated labels/<stem>.npy; resume must re-segment that image
    instead of dropping it, and must not trust stored labels of the wrong shape."""
    folder = str(tmp_path / 'in')
    make_inputs(folder, 4)
    paths = get_imgs(folder)
    rec = metaseg.run(folder, StubModel(), paths, 0, 1, batch_images=2, io_threads=2, log=lambda *a: None)
    want = rec[:, edist.F_NEC].copy()
    good = open(os.path.join(folder, 'labels', 'img01.npy'), 'rb').read()
    open(os.path.join(folder, 'labels', 'img01.npy'), 'wb').write(good[:len(good) // 3])           # killed mid-write
    np.save(os.path.join(folder, 'labels', 'img02.npy'), np.zeros((5, 7), np.int64))                # stale, wrong shape
    model = StubModel()
    logs = []
    rec2 = metaseg.run(folder, model, paths, 0, 1, batch_images=2, io_threads=2, log=lambda *a: logs.append(' '.join(map(str, a))),
                       resume=True)
    assert (rec2[:, edist.F_STATUS] == 0).all() and np.array_equal(rec2[:, edist.F_NEC], want)
    assert sum(n for n, _ in model.handle.calls) == 2                          # exactly the two damaged images were segmented again
    assert sum('Keeping existing outputs' in l for l in logs) == 2
    assert open(os.path.join(folder, 'labels', 'img01.npy'), 'rb').read() == good
    assert np.load(os.path.join(folder, 'labels', 'img02.npy')).shape == (H, W)


def test_oom_retry_halves_below_the_failed_group_and_restores_the_setting(tmp_path):
    imgs = np.zeros((8, H, W), np.uint8)
    model = StubModel(fail_above=2)
    model.handle.set_images_per_group(8)
    logs = []
    gray, post, nec, tie, probs = metaseg._segment_with_retry(model, imgs, lambda *a: logs.append(a))
    assert post.shape == (8, H, W) and len(nec) == 8 and len(tie) == 8 and probs is None
    assert [g for _, g in model.handle.calls] == [8, 4, 2]                     # first retry is BELOW the group that failed
    assert model.handle.images_per_group == 8                                  # restored for the next batch
    # a batch that does not fit even with one image per group is split by images
    class PerBatch(StubModel):
        def segment(self, gray):
            if len(gray) > 3:
                e = EcsegError('simulated'); e.code = E_NOMEM
                self.handle.calls.append((len(gray), self.handle.images_per_group))
                raise e
            return StubModel.segment(self, gray)
    m2 = PerBatch()
    gray, post, nec, tie, probs = metaseg._segment_with_retry(m2, imgs, lambda *a: None)
    assert post.shape == (8, H, W) and m2.handle.images_per_group == 0
    assert [n for n, _ in m2.handle.calls if n <= 3] == [2, 2, 2, 2]


def test_native_rendezvous_file_and_transport_choice(tmp_path, monkeypatch):
    """The torch-free transport of `make metaseg` (ECSEG_DIST=native): rank 0 publishes the 128-byte communicator id through a
    file, late and early readers both get exactly those bytes; one rank without any transport: gather_all = compaction."""
    import threading
    import time
    path = str(tmp_path / 'rdzv')
    payload = bytes(range(128))
    got = []
    t = threading.Thread(target=lambda: got.append(edist.read_rendezvous(path, 128, timeout=20)))
    t.start()                                                     # reader first: polls until the writer's rename
    time.sleep(0.2)
    open(path, 'wb').write(b'short')                              # a partial / foreign file is not accepted
    time.sleep(0.2)
    edist.write_rendezvous(path, payload)
    t.join(timeout=30)
    assert got == [payload] and edist.read_rendezvous(path, 128, timeout=1) == payload
    with pytest.raises(TimeoutError):
        edist.read_rendezvous(str(tmp_path / 'nobody'), 128, timeout=0.3)
    monkeypatch.setenv('ECSEG_DIST', 'native')
    assert edist.want_native()
    monkeypatch.setenv('ECSEG_DIST', 'torch')
    assert not edist.want_native()
    rec = edist.make_records(3, 2, 4, n_ec=[5, 6])
    out = edist.gather_all(rec)
    assert out.shape == (2, edist.RECORD_INT64) and out[:, edist.F_INDEX].tolist() == [3, 4]


def test_one_bad_image_costs_one_status_row_not_its_batch(tmp_path):
    """VERDICT r03 item 8 / SURVEY 5 (per-image status): a device error that is not out-of-memory used to mark the whole batch
    (8 images) failed; the batch is now bisected like the OOM path, so exactly the bad image gets status 2 and everything
    else - outputs, counts, CSV - equals a run without it."""
    from PIL import Image
    folder = str(tmp_path / 'in')
    make_inputs(folder, 9)
    bad = os.path.join(folder, 'img04.tif')
    img = np.array(Image.open(bad))
    img[0, 0, 2] = 201                                                       # the stub's poison value
    Image.fromarray(img).save(bad, compression='tiff_lzw')
    paths = get_imgs(folder)
    m = StubModel(poison=201)
    logs = []
    rec = metaseg.run(folder, m, paths, 0, 1, batch_images=8, io_threads=2, log=lambda *a: logs.append(' '.join(str(x) for x in a)))
    failed = metaseg.finish(folder, paths, rec, 0, log=lambda *a: None)
    assert [int(r[edist.F_STATUS]) for r in rec] == [2 if i == 4 else 0 for i in range(9)]
    assert len(failed) == 1 and any('img04.tif' in line and 'simulated device fault' in line for line in logs)
    csv = open(os.path.join(folder, 'ec_quantification.csv')).read()
    assert csv.count('\n') == 9 and 'img04.tif' not in csv                  # header + 8 rows
    for i in range(9):
        assert os.path.exists(os.path.join(folder, 'labels', 'img%02d.npy' % i)) == (i != 4)
    # the batch of 8 was bisected down to the single image: 8 -> 4 (fine) + 4 -> 2 -> 1 (bad) + 1, then the other 2, then image 8
    sizes = [c[0] for c in m.handle.calls]
    assert sizes == [8, 4, 4, 2, 1, 1, 2, 1]


def test_two_device_workers_give_the_same_outputs_as_one(tmp_path):
    """Config key device_workers: several handles on one GPU take batches from one queue (copies of one overlap kernels of the
    other).  Results - per-image records, files, CSV - must not depend on which worker took which batch."""
    one, two = str(tmp_path / 'one'), str(tmp_path / 'two')
    make_inputs(one, 11, corrupt=5)
    shutil.copytree(one, two)
    p1, p2 = get_imgs(one), get_imgs(two)
    rec1 = metaseg.run(one, StubModel(), p1, 0, 1, batch_images=2, io_threads=2, log=lambda *a: None)
    metaseg.finish(one, p1, rec1, 0, log=lambda *a: None)
    models = [StubModel(), StubModel()]
    rec2 = metaseg.run(two, models, p2, 0, 1, batch_images=2, io_threads=2, log=lambda *a: None)
    metaseg.finish(two, p2, rec2, 0, log=lambda *a: None)
    assert np.array_equal(rec1, rec2)
    assert all(len(m.handle.calls) > 0 for m in models), 'a worker never got a batch'
    a, b = _snapshot(one), _snapshot(two)
    assert sorted(a) == sorted(b) and all(a[k] == b[k] for k in a)


def test_sticky_device_error_ends_the_rank_instead_of_bisecting(tmp_path):
    """a HIP runtime error (device fault, lost context: ECSEG_E_HIP) is not a property of one image - bisecting would
    burn O(n log n) doomed device calls per batch and the job would "finish" with every image marked failed and exit code 0.  The
    rank raises after the first such error: one device call per worker, no bisection, the exception reaches the caller."""
    from ecseg_amd._lib import E_HIP
    folder = str(tmp_path / 'in')
    make_inputs(folder, 9)
    m = StubModel()
    calls = []

    def broken(gray):
        calls.append(len(gray))
        e = EcsegError('hipErrorIllegalAddress (simulated)')
        e.code = E_HIP
        raise e
    m.segment = broken
    with pytest.raises(EcsegError, match='hipErrorIllegalAddress'):
        metaseg.run(folder, m, get_imgs(folder), 0, 1, batch_images=3, io_threads=2, log=lambda *a: None)
    assert calls == [3], calls                                           # the first batch, once: no halves, no later batches


def test_fused_call_and_page_locked_pool_give_the_same_outputs(tmp_path):
    """Round 5 host path: a handle with ``meta_segment`` (ecseg_meta_segment: pre-process + segment in one device call) and
    ``host_empty`` (page-locked batch buffers, recycled through metaseg._PinnedPool) must give the records, files and CSV of
    the two-call path over ordinary memory.  The pool never allocates on the device thread: a miss is served from ordinary
    memory and ordered from the pool's own thread; buffers return when a batch's last output file is written."""
    one, two = str(tmp_path / 'one'), str(tmp_path / 'two')
    make_inputs(one, 13, corrupt=4)
    shutil.copytree(one, two)
    p1, p2 = get_imgs(one), get_imgs(two)
    rec1 = metaseg.run(one, StubModel(), p1, 0, 1, batch_images=3, io_threads=2, log=lambda *a: None)
    metaseg.finish(one, p1, rec1, 0, log=lambda *a: None)

    class FusedHandle(StubHandle):
        def __init__(self):
            super().__init__()
            self.allocated, self.fused_calls, self.filled = [], 0, 0
            self.sent_ahead, self.seen = [], []

        def host_empty(self, shape, dtype=np.uint8):
            a = np.full(shape, 0xAB, dtype)                  # (stale bytes: a result must not depend on them)
            self.allocated.append(a)
            return a

        registered = None                                    # the library's state: images named for the call after the coming one

        def prefetch_input(self, imgs):
            if imgs is None:                                 # withdraw (a batch that is not sent ahead: ordinary memory, resumed images)
                self.registered = None
                return
            assert any(np.shares_memory(imgs, a) for a in self.allocated), 'only page-locked batches are sent ahead'
            self.sent_ahead.append(imgs.ctypes.data)
            self.registered = imgs.ctypes.data

        def count_cc(self, masks):
            # a count-only batch (resumed images) must not leave a registration behind - the coming meta_segment
            # call would send its OWN buffer ahead and a later batch in the recycled buffer would hit stale pixels
            assert self.registered is None, 'a batch that never reaches meta_segment left images registered'
            return super().count_cc(masks)

        def meta_segment(self, imgs, gray_out=None, post_out=None):
            self.fused_calls += 1
            self.seen.append(imgs.ctypes.data)
            assert self.registered != imgs.ctypes.data, 'the coming call was handed its own images as the ones to send ahead'
            self.registered = None
            time.sleep(0.03)                                 # (a device call takes a while: the feeder gets a batch ahead)
            gray, _ = self.preprocess(imgs)
            post, nec = StubModel.segment(model, gray)
            if gray_out is not None:
                assert any(np.shares_memory(gray_out, a) for a in self.allocated) and gray_out.shape == gray.shape
                gray_out[...] = gray; post_out[...] = post
                gray, post = gray_out, post_out
                self.filled += 1
            return gray, post, nec, np.zeros(len(gray), np.int32)

    model = StubModel()
    model.handle = FusedHandle()
    rec2 = metaseg.run(two, model, p2, 0, 1, batch_images=3, io_threads=2, log=lambda *a: None, pinned_mb=64)
    metaseg.finish(two, p2, rec2, 0, log=lambda *a: None)
    assert np.array_equal(rec1, rec2)
    a, b = _snapshot(one), _snapshot(two)
    assert sorted(a) == sorted(b) and all(a[k] == b[k] for k in a)
    h = model.handle
    pool = h._metaseg_pool
    assert h.fused_calls >= 4 and pool.hits + pool.misses > 0          # (how many of the first batches miss is a matter of timing)
    assert pool.thread is None, 'the pool thread must end with the run'
    # every buffer is back in the pool (none leaked to a writer), the limit was respected, and a second run starts with hits
    assert sorted(id(x) for x in pool.free) == sorted(id(x) for x in h.allocated)
    assert sum(x.size for x in h.allocated) <= 64 << 20
    # two size classes only - full input batches and full label batches (ordered when the first image's shape was known, partial
    # batches use a prefix) - and a label request never takes an input buffer
    assert {x.size for x in h.allocated} <= {3 * H * W * 3, 3 * H * W}
    small = pool.get(3 * H * W, 3 * H * W) if pool.thread else None
    assert small is None                                                  # (the pool is stopped: get() is a plain miss)
    pool.start()
    try:
        got = pool.get(3 * H * W, 3 * H * W)
        assert got is not None and got.size == 3 * H * W
        pool.put(got)
    finally:
        pool.stop()
    hits0 = pool.hits
    shutil.rmtree(two); shutil.copytree(one, two)
    for sub in ('labels', 'dapi'):
        shutil.rmtree(os.path.join(two, sub), ignore_errors=True)
        os.makedirs(os.path.join(two, sub))
    rec3 = metaseg.run(two, model, get_imgs(two), 0, 1, batch_images=3, io_threads=2, log=lambda *a: None, pinned_mb=64)
    assert np.array_equal(rec1, rec3) and pool.hits > hits0 and h.filled > 0
    # batches sent ahead (ecseg_prefetch_input) are exactly the ones the next device call then receives
    assert h.sent_ahead and all(a in h.seen for a in h.sent_ahead)
    # resume over a folder where every other image is done: count-only batches alternate with device batches
    done = sorted(os.listdir(os.path.join(two, 'labels')))
    for k, f in enumerate(x for x in done if x.endswith('.npy')):
        if k % 2:
            os.remove(os.path.join(two, 'labels', f))
    calls0 = h.fused_calls
    rec4 = metaseg.run(two, model, get_imgs(two), 0, 1, batch_images=3, io_threads=2, log=lambda *a: None, pinned_mb=64, resume=True)
    assert np.array_equal(rec1, rec4) and h.fused_calls > calls0
    # pinned_mb = 0: the pool is never asked
    m0 = StubModel(); m0.handle = FusedHandle()
    metaseg.run(two, m0, get_imgs(two), 0, 1, batch_images=3, io_threads=2, log=lambda *a: None, pinned_mb=0)
    assert m0.handle.allocated == [] and m0.handle.fused_calls >= 4

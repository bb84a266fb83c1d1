#!/usr/bin/env python3
"""``make metaseg``: drop-in for the reference's ``src/metaseg.py`` (same config.yaml section, messages, exit codes and
output files), running on MI355X.

Differences that do not change results: images are processed in batches on the GPU instead of one by one, in sorted
order; TIFF decoding and the output encoders (dapi TIFF, label PNG, int64 .npy) run on worker threads around the GPU
batches with a bounded number of images in flight (the reference's loop at src/metaseg.py:42-54 is serial); with
``device_ids: [0, 1, ...]`` in the config section (or under ``torchrun``) the images are sharded across the GPUs of the
node, every rank writes the outputs of its own images and rank 0 writes ``ec_quantification.csv`` after one all-gather
of the per-image records (the reference's analogue is the implicit all-device MirroredStrategy, src/metaseg.py:33-36).

Optional config keys (defaults keep the reference's behaviour): ``batch_images`` (8), ``io_threads``, ``device_ids``,
``resume`` (false; true: images whose ``labels/<stem>.npy``, ``.png`` and ``dapi/<name>`` exist are not segmented again),
``device_workers`` (1), ``pinned_mb`` (2048: page-locked host memory for the batch buffers; 0: none).
"""
import concurrent.futures as cf
import json
import os
import queue
import subprocess
import sys
import threading
import time

import numpy as np
import yaml

from . import csvio, dist, image_io
from ._lib import E_HIP, E_NOMEM, EcsegError
from .utils import default_io_threads, get_imgs, load_model, save_img, tune_host_allocator

MODEL_NAME = 'metaseg.h5'


def _outputs_of(p):
    path_split = os.path.split(p)
    stem = os.path.join(path_split[0], 'labels', path_split[1][:-4])
    return stem + '.npy', stem + '.png', os.path.join(path_split[0], 'dapi', path_split[1])


def _resume_probe(p):
    """Optional resume ("checkpoint / resume"): an image whose three outputs exist, whose stored labels load and
    have the input image's shape is not segmented again; its count is taken from the stored labels (count_cc(I == 3) on the
    device, as src/metaseg.py:46 does).  ANY doubt - a truncated .npy of a killed run, a wrong shape or dtype, an unreadable
    header - returns None and the image is segmented again (the .npy is written last and atomically, see _write_outputs)."""
    try:
        outs = _outputs_of(p)
        if not all(os.path.exists(o) and os.path.getsize(o) > 0 for o in outs):
            return None
        lab = np.load(outs[0])
        if lab.ndim != 2 or lab.dtype != np.int64 or tuple(lab.shape) != tuple(image_io.image_shape(p)[:2]):
            return None
        return ('done', np.ascontiguousarray(lab == 3))
    except Exception:
        return None


def _read(p, resume=False):
    if resume:
        done = _resume_probe(p)
        if done is not None:
            return done
    img = image_io.imread(p)
    if img.dtype not in (np.uint8, np.uint16) or img.ndim not in (2, 3):
        raise ValueError('unsupported image array %s %s' % (img.dtype, img.shape))
    return img


def _replace_into(path, write):
    """Write through a temporary name in the same directory and rename: a killed run never leaves a truncated output."""
    tmp = '%s.tmp%d' % (path, os.getpid())
    try:
        write(tmp)
        os.replace(tmp, path)
    except BaseException:
        try:
            os.unlink(tmp)
        except OSError:
            pass
        raise


def _write_outputs(p, gray, post, log, probs=None):
    """gray: the pre-processed image; dapi/<name> holds cv2.bitwise_not of it (src/utils.py:112,122-123).  probs (config key
    ``emit_probs``): the stitched float32 probabilities -> labels/<stem>_probs.npy, written before the resume marker."""
    path_split = os.path.split(p)
    dapi = os.path.join(path_split[0], 'dapi', path_split[1])
    if dapi.lower().endswith(('.tif', '.tiff')):
        _replace_into(dapi, lambda t: image_io.write_tiff_gray8(t, gray, invert=True))
    else:
        save_img(~gray, path_split, 'dapi')
    outpath = os.path.join(path_split[0], 'labels', path_split[1][:-4])
    log("Saving labels: ", p, " to ", outpath)
    _replace_into(outpath + '.png', lambda t: image_io.write_label_png(t, post))
    if probs is not None:
        def save_probs(t):
            with open(t, 'wb') as f:                        # (np.save would append ".npy" to the temporary name)
                np.save(f, np.ascontiguousarray(probs, np.float32))
        _replace_into(outpath + '_probs.npy', save_probs)
    # int64 .npy (src/metaseg.py:53), LAST: its presence is the resume marker
    _replace_into(outpath + '.npy', lambda t: image_io.write_npy_int64(t, post))


class _PinnedPool:
    """Page-locked host buffers (Handle.host_empty) for the batches of `make metaseg`, recycled: the raw images go up and the
    pre-processed images and labels come down as DMA transfers the device thread does not wait for (tools/experiments/
    host_call_probe.py: a batch of 32 images on a base-16 model, 1.50 -> 1.21 ms per image).  Page-locking costs ~0.2 ms per
    MB - five uses of a buffer - so ``get`` never allocates: a miss returns None (the caller uses ordinary memory for this
    batch) and asks the pool's own thread for a buffer of that size, which later batches find.  Buffers are only ever
    allocated up to ``limit_bytes``.  The pool belongs to the handle (buffers survive from one run to the next and are freed
    with it); a run brackets its use with start() / stop()."""

    def __init__(self, handle, limit_bytes):
        self.handle, self.limit, self.left = handle, int(limit_bytes), int(limit_bytes)
        self.free, self.lock = [], threading.Lock()
        self.alloc = getattr(handle, 'host_empty', None)
        self.requests, self.thread, self.stopping = None, None, False
        self.hits = self.misses = 0

    @classmethod
    def of(cls, handle, limit_bytes):
        pool = getattr(handle, '_metaseg_pool', None)
        if pool is None:
            pool = cls(handle, limit_bytes)
            try:
                handle._metaseg_pool = pool
            except AttributeError:
                pass
        elif int(limit_bytes) > pool.limit:                 # a later run may raise the limit (never lowers what is already locked)
            with pool.lock:
                pool.left += int(limit_bytes) - pool.limit
                pool.limit = int(limit_bytes)
        return pool

    def start(self):
        if self.alloc is not None and self.thread is None:
            self.requests = queue.Queue()
            self.stopping = False
            self.thread = threading.Thread(target=self._grow, name='ecseg-pinned', daemon=True)
            self.thread.start()

    def stop(self):
        if self.thread is not None:
            self.stopping = True                            # orders not started yet are dropped: a short run does not wait for them
            self.requests.put(None)
            self.thread.join()
            self.thread = None

    def _grow(self):
        while True:
            want = self.requests.get()
            if want is None:
                return
            if self.stopping:
                with self.lock:
                    self.left += want
                continue
            try:
                b = self.alloc((want,), np.uint8)
            except EcsegError:
                with self.lock:
                    self.left = 0                           # the system will not lock more pages: stop asking
                continue
            with self.lock:
                self.free.append(b)

    def get(self, nbytes, full_bytes=0):
        """-> flat uint8 array of at least ``nbytes`` or None.  (A miss orders max(nbytes, full_bytes): the buffer of a partial
        batch then serves full ones.)"""
        if self.thread is None:
            return None
        with self.lock:
            # (a buffer more than twice the size asked for stays where it is: label buffers must not eat the input buffers)
            fit = [k for k, b in enumerate(self.free) if nbytes <= b.size <= 2 * max(int(nbytes), int(full_bytes))]
            if fit:
                self.hits += 1
                return self.free.pop(min(fit, key=lambda k: self.free[k].size))
            self.misses += 1
            want = max(int(nbytes), int(full_bytes))
            if want <= self.left:
                self.left -= want
                self.requests.put(want)
        return None

    def put(self, b):
        if b is not None:
            with self.lock:
                self.free.append(b)

    def reserve(self, sizes):
        """Order buffers ahead of need (in this order, within the limit): the first batches of a run then find theirs instead
        of missing one by one.  A size already served by a free buffer of its class is skipped."""
        if self.thread is None:
            return
        with self.lock:
            spare = [b.size for b in self.free]
            for want in sizes:
                want = int(want)
                k = next((k for k, sz in enumerate(spare) if want <= sz <= 2 * want), None)
                if k is not None:
                    spare.pop(k)
                elif want <= self.left:
                    self.left -= want
                    self.requests.put(want)


def _view(buf, shape, dtype=np.uint8):
    n = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    return buf[:n].view(dtype).reshape(shape)


def _segment_group(model, imgs, emit_probs=False, outs=None):
    """-> gray, post, n_ec, tie_risk, probs | None (arrays over the batch).  ``outs``: (gray, post) arrays to fill."""
    fused = getattr(model.handle, 'meta_segment', None)
    if fused is not None and not emit_probs:               # one device call: the pre-processed images stay on the GPU
        gray, post, nec, tie = fused(imgs, *(outs or (None, None)))
        return gray, post, nec, tie, None
    gray, _ = model.handle.preprocess(imgs)
    ex = getattr(model, 'segment_ex', None)
    if ex is None:                                         # a model without the extended call: no tie-risk bound, no probabilities
        post, nec = model.segment(gray)
        return gray, post, nec, np.zeros(len(gray), np.int32), None
    out = ex(gray, want_probs=emit_probs)
    return (gray, out[0], out[1], out[2], out[3] if emit_probs else None)


def _segment_with_retry(model, imgs, log, emit_probs=False, outs=None):
    """GPU part of one batch.  On an out-of-memory status the internal launch group is halved, starting below what the failed
    attempt used (down to one image per launch); if that still does not fit, the batch itself is split by images (allocations
    that scale with the batch: post-processing workspace, input staging).  The handle's setting is restored afterwards, so
    one oversized batch does not slow every later one."""
    h = model.handle
    before = h.images_per_group
    try:
        group = min(before, len(imgs)) if before > 0 else len(imgs)
        while True:
            try:
                return _segment_group(model, imgs, emit_probs, outs)
            except EcsegError as e:
                if e.code != E_NOMEM:
                    raise
                group //= 2
                if group < 1:
                    break
                log("Out of device memory for a batch of shape %s: retrying with %d image(s) per launch group" % (imgs.shape, group))
                h.set_images_per_group(group)
        if len(imgs) < 2:
            raise EcsegError('out of device memory for a single image of shape %s' % (imgs.shape[1:],))
        half = len(imgs) // 2
        log("Out of device memory for a batch of shape %s: splitting it into %d + %d image(s)" % (imgs.shape, half, len(imgs) - half))
        lo, hi = (tuple(o[:half] for o in outs), tuple(o[half:] for o in outs)) if outs else (None, None)
        a, b = _segment_with_retry(model, imgs[:half], log, emit_probs, lo), _segment_with_retry(model, imgs[half:], log, emit_probs, hi)
        return tuple(None if x is None else np.concatenate([x, y]) for x, y in zip(a, b))
    finally:
        h.set_images_per_group(before)


def _segment_isolating(model, imgs, log, emit_probs=False, outs=None):
    """GPU part of one batch with per-image failure isolation (a per-image status): a batch that fails for any reason
    other than memory (_segment_with_retry deals with that) is bisected until the failing image(s) stand alone - one bad image
    costs one status-2 row, not the whole batch.  -> list of (index in batch, gray, post, n_ec, tie_risk, probs | None) for the
    images that went through, list of (index, exception) for those that did not."""
    try:
        gray, post, nec, tie, probs = _segment_with_retry(model, imgs, log, emit_probs, outs)
        return [(j, gray[j], post[j], int(nec[j]), int(tie[j]), None if probs is None else probs[j]) for j in range(len(imgs))], []
    except EcsegError as e:
        # only failures that can belong to ONE input are bisected (a bad argument / shape, an unsupported layout, memory for a single
        # giant image).  A HIP runtime error is sticky - after a device fault or a lost context every sub-batch would fail again,
        # O(n log n) doomed calls per batch and a job that "finishes" with an empty CSV - and anything that is not an EcsegError
        # is a programming error: both end the rank (_supervise_native tears the job down).
        if e.code == E_HIP:
            raise
        if len(imgs) == 1:
            return [], [(0, e)]
        half = len(imgs) // 2
        log("A batch of %d image(s) of shape %s failed on the device (%s): retrying it as %d + %d"
            % (len(imgs), imgs.shape[1:], e, half, len(imgs) - half))
        ok_a, bad_a = _segment_isolating(model, imgs[:half], log, emit_probs)
        ok_b, bad_b = _segment_isolating(model, imgs[half:], log, emit_probs)
        return (ok_a + [(r[0] + half,) + r[1:] for r in ok_b], bad_a + [(j + half, e2) for j, e2 in bad_b])


def run(inpath, model, image_paths, rank=0, world=1, batch_images=8, io_threads=None, log=print, stats=None, resume=False,
        emit_probs=False, pinned_mb=2048, pinned_min_images=None):
    """Segment this rank's shard; returns records (one row per image of the WHOLE job after the all-gather)."""
    start, stop, per = dist.shard_bounds(len(image_paths), rank, world)
    mine = image_paths[start:stop]
    n_ec = np.zeros(len(mine), np.int64)
    status = np.zeros(len(mine), np.int64)
    tie = np.zeros(len(mine), np.int64)
    io_threads = io_threads or default_io_threads(world)                    # the cgroup-aware CPU budget shared between the ranks
    window = max(2 * batch_images, io_threads)                             # images decoded ahead of the GPU
    pending_writes = threading.BoundedSemaphore(4 * batch_images + io_threads)   # bounds the outputs held in memory
    # dozens of decoder / encoder threads hold the GIL in 5 ms slices by default; the device thread needs it only for
    # microseconds between calls, so a short switch interval keeps the GPU fed
    old_switch = sys.getswitchinterval()
    sys.setswitchinterval(0.0005)
    tune_host_allocator()                                                   # one malloc arena, no mmap per buffer (utils.py)
    # page-locked batch buffers (config key pinned_mb, default 2048; 0: ordinary memory as in rounds 1-4)
    first = model[0] if isinstance(model, (list, tuple)) else model
    pool = _PinnedPool.of(first.handle, int(pinned_mb) << 20)
    # (page-locking pays after ~5 uses of a buffer: not for a handful of images; pinned_min_images overrides the threshold)
    if pinned_mb > 0 and len(mine) >= (4 * batch_images if pinned_min_images is None else pinned_min_images):
        pool.start()
    try:
        return _run_threads(model, mine, start, per, rank, world, batch_images, io_threads, window, pending_writes, n_ec,
                            status, log, stats, resume, tie, emit_probs, pool)
    finally:
        pool.stop()
        sys.setswitchinterval(old_switch)


def _run_threads(model, mine, start, per, rank, world, batch_images, io_threads, window, pending_writes, n_ec, status, log, stats,
                 resume=False, tie=None, emit_probs=False, pool=None):
    # `model`: one model, or a list of models on the same GPU (config key device_workers): one device thread per model takes
    # batches from the common queue, so the host <-> device copies and the per-call synchronisation of one overlap the kernels of
    # the other (a narrow model spends a third of a call outside its kernels: EXPERIMENTS.md 6)
    models = list(model) if isinstance(model, (list, tuple)) else [model]
    model = models[0]
    t_gpu = 0.0
    t_lock = threading.Lock()
    # per-stage host timers (thread-seconds summed over the worker threads): where a slow `make metaseg` spends its CPU
    # (device_kernels: the library's stage timers; device_native: inside the library call, the rest of gpu_seconds is Python)
    t_stage = {'read': 0.0, 'write': 0.0, 'pack': 0.0, 'device_kernels': 0.0, 'device_native': 0.0}

    def timed(stage, fn):
        def wrapped(*a, **kw):
            t0 = time.perf_counter()
            try:
                return fn(*a, **kw)
            finally:
                dt = time.perf_counter() - t0
                with t_lock:
                    t_stage[stage] += dt
        return wrapped

    read_fn, write_fn = timed('read', _read), timed('write', _write_outputs)
    with cf.ThreadPoolExecutor(io_threads) as readers, cf.ThreadPoolExecutor(io_threads) as writers, \
            cf.ThreadPoolExecutor(min(8, io_threads)) as packers:
        reads = {}
        next_submit = 0

        def top_up(upto):
            nonlocal next_submit
            while next_submit < min(len(mine), upto):
                log("Processing image: ", mine[next_submit])
                reads[next_submit] = readers.submit(read_fn, mine[next_submit], resume)
                next_submit += 1

        write_futs = []

        def flush(batch, model):
            nonlocal t_gpu
            group, imgs, in_buf = batch
            if not group:
                return
            if imgs.dtype == np.bool_:                     # resumed images: only the count is needed
                cnt, _ = model.handle.count_cc(imgs)
                for j, k in enumerate(group):
                    n_ec[k] = int(cnt[j])
                return
            t0 = time.perf_counter()
            nat0 = getattr(model.handle, 'native_seconds', 0.0)
            # page-locked output buffers of the batch; they go back to the pool when its last output file is written
            n, (H, W) = len(imgs), imgs.shape[1:3]
            full = max(n, batch_images) * H * W
            bufs = [pool.get(n * H * W, full) for _ in range(2)] if not emit_probs else [None, None]
            outs = tuple(_view(b, (n, H, W)) for b in bufs) if all(b is not None for b in bufs) else None
            try:
                done, bad = _segment_isolating(model, imgs, log, emit_probs, outs)
            finally:
                pool.put(in_buf)                           # (the call has returned: the upload is over)
                if outs is None:
                    for b in bufs:
                        pool.put(b)
            dt = time.perf_counter() - t0
            tm = getattr(model.handle, 'timings', None)
            kern = 1e-3 * sum(tm().values()) if tm is not None else 0.0
            with t_lock:
                t_gpu += dt
                t_stage['device_kernels'] += kern
                t_stage['device_native'] += getattr(model.handle, 'native_seconds', 0.0) - nat0
            for j, e in bad:                               # a failing image must not take its batch or the shard down
                log("Skipping %s (shape %s): %s" % (mine[group[j]], imgs.shape[1:], e))
                status[group[j]] = 2
            left = [len(done)]

            def written(_f):
                pending_writes.release()
                with t_lock:
                    left[0] -= 1
                    last = left[0] == 0
                if last and outs is not None:
                    for b in bufs:
                        pool.put(b)

            if not done and outs is not None:
                for b in bufs:
                    pool.put(b)
            for j, gray_j, post_j, nec_j, tie_j, probs_j in done:
                k = group[j]
                n_ec[k] = nec_j
                if tie is not None:
                    tie[k] = tie_j
                pending_writes.acquire()
                f = writers.submit(write_fn, mine[k], gray_j, post_j, log, probs_j)
                f.add_done_callback(written)
                write_futs.append((k, f))

        # the device calls run on their own thread (one batch in the queue, one on the GPU) so that collecting decoded
        # images and handing results to the writers overlaps the U-Net
        batches = queue.Queue(maxsize=2 * len(models))

        fatal = []                                         # a sticky device error / programming error: the rank ends after the drain

        def gpu_loop(m):
            prefetch = None if emit_probs else getattr(m.handle, 'prefetch_input', None)
            ahead = []                                     # the batch taken from the queue while looking ahead
            while True:
                g = ahead.pop() if ahead else batches.get()
                if g is None:
                    return
                # look ahead without waiting: the raw images of the next batch - when they lie in page-locked memory - go up
                # under this batch's kernels (ecseg_prefetch_input), so one handle overlaps its copies with its own compute
                # (with several device workers a batch parked in `ahead` would keep the other worker idle: no look-ahead there.  A
                # batch that never reaches ecseg_meta_segment - resumed images, only counted - must not leave a registration
                # behind: the library keys the hit on (pointer, size) and the buffer goes back to the pool.  ADVICE r05)
                if prefetch is not None and len(models) == 1:
                    try:
                        ahead.append(batches.get_nowait())
                        nxt = ahead[0]
                        to_device = g[0] and g[1].dtype != np.bool_
                        if nxt is not None and nxt[2] is not None and nxt[1].dtype != np.bool_ and to_device and not fatal:
                            prefetch(nxt[1])
                        else:
                            prefetch(None)
                    except queue.Empty:
                        pass
                    except BaseException as e:             # an error here must not end the loop: the feeder would block for ever
                        if not fatal and not (isinstance(e, EcsegError) and e.code != E_HIP):
                            fatal.append(e)
                try:
                    if fatal:                              # keep draining so that the feeder never blocks; nothing reaches the device
                        raise fatal[0]
                    flush(g, m)
                except BaseException as e:                 # never leave the feeding loop blocked on a dead consumer
                    if not fatal and not (isinstance(e, EcsegError) and e.code != E_HIP):
                        fatal.append(e)
                    log("Skipping %d image(s): %r" % (len(g[0]), e))
                    for k in g[0]:
                        status[k] = 2

        def pack(group):                                   # the batch array is assembled by the feeder, not by the device thread
            t0 = time.perf_counter()
            first = group[0][1]
            buf = pool.get(len(group) * first.nbytes, batch_images * first.nbytes) if first.dtype != np.bool_ else None
            if buf is None:
                arr = np.stack([im for _, im in group])
            else:                                          # page-locked batch buffer, filled by several threads (13 GB/s each)
                arr = _view(buf, (len(group),) + first.shape, first.dtype)
                list(packers.map(lambda j: np.copyto(arr[j], group[j][1]), range(len(group))))
            t_stage['pack'] += time.perf_counter() - t0
            return [k for k, _ in group], arr, buf

        gpu_threads = [threading.Thread(target=gpu_loop, args=(m,), name='ecseg-gpu%d' % i) for i, m in enumerate(models)]
        for th in gpu_threads:
            th.start()
        try:
            group, key, reserved = [], None, False
            for k in range(len(mine)):
                top_up(k + window)
                try:
                    img = reads.pop(k).result()
                except Exception as e:                     # a corrupt image must not take the shard down
                    log("Skipping %s: %s" % (mine[k], e))
                    status[k] = 1
                    continue
                if isinstance(img, tuple):                 # ('done', ecDNA mask of the stored labels)
                    img = img[1]
                    log("Keeping existing outputs of ", mine[k])
                kk = (img.shape, img.dtype.str)
                if not reserved and img.dtype != np.bool_:
                    # the shape of the run is known with its first image: order the page-locked buffers of the first batches now
                    # (three inputs in flight - being packed, queued, on the device - and three output pairs being written)
                    reserved = True
                    b_in, b_out = batch_images * img.nbytes, batch_images * img.shape[0] * img.shape[1]
                    pool.reserve([b_in, b_out, b_out, b_in, b_out, b_out, b_in, b_out, b_out])
                if group and (kk != key or len(group) >= batch_images):
                    batches.put(pack(group))
                    group = []
                group.append((k, img))
                key = kk
            if group:
                batches.put(pack(group))
        finally:
            for _ in gpu_threads:
                batches.put(None)
            for th in gpu_threads:
                th.join()
        if fatal:
            # a HIP runtime error does not go away (device fault, lost context) and a non-library exception is a bug: fail the rank
            # loudly instead of finishing with every image marked failed and exit code 0
            for _, f in write_futs:
                f.cancel()
            raise fatal[0]
        for k, f in write_futs:
            try:
                f.result()
            except Exception as e:
                log("Could not write the outputs of %s: %s" % (mine[k], e))
                status[k] = 3
    if stats is not None:
        stats['gpu_seconds'] = t_gpu
        stats['pinned_pool'] = {'hits': pool.hits, 'misses': pool.misses, 'free_MB': round(sum(b.size for b in pool.free) / 1e6, 1)}
        stats.update({'%s_thread_seconds' % k: v for k, v in t_stage.items()})
    rec = dist.make_records(start, len(mine), per, n_ec=n_ec, status=status, tie_risk=tie)
    return dist.gather_all(rec, device=model.handle.device)        # the path's one exchange (identity for one rank)


def finish(inpath, image_paths, rec, rank, seconds=None, gpu_seconds=0.0, log=print):
    """After the all-gather every rank holds the records of the whole job; rank 0 writes ``ec_quantification.csv`` (one row
    per successfully processed image, in sorted-path order: src/metaseg.py:44-46,56-57) and lists what failed.  Returns the
    failed records (same on every rank)."""
    failed = [r for r in rec if r[dist.F_STATUS] != 0]
    if rank != 0:
        return failed
    rows = [[os.path.split(image_paths[int(r[dist.F_INDEX])])[1], int(r[dist.F_NEC])]
            for r in rec if r[dist.F_STATUS] == 0]
    out = os.path.join(inpath, 'ec_quantification.csv')
    log("Saving ec quantification to", out)
    text = csvio.csv_text(csvio.METASEG_COLUMNS, rows)
    for name in ('ec_quantification.csv', 'ec_quantifications.csv'):       # the second is the name README.md:86 uses
        tmp = os.path.join(inpath, name + '.tmp%d' % os.getpid())
        with open(tmp, 'w') as f:
            f.write(text)
        os.replace(tmp, os.path.join(inpath, name))
    # per-image report beside the reference's two-column CSV: status, count and the tie-risk bound of every image (pixels whose
    # label a last-bit difference between float32 evaluations of the network can flip - record slot 15)
    report = {'images': [{'image_name': os.path.split(image_paths[int(r[dist.F_INDEX])])[1], 'status': int(r[dist.F_STATUS]),
                          '# of ec': int(r[dist.F_NEC]), 'tie_risk_pixels': int(r[dist.F_TIE])} for r in rec]}
    tmp = os.path.join(inpath, 'ec_quantification_report.json.tmp%d' % os.getpid())
    with open(tmp, 'w') as f:
        json.dump(report, f, indent=1)
    os.replace(tmp, os.path.join(inpath, 'ec_quantification_report.json'))
    if image_paths and seconds:
        log("%d image(s) in %.2f s (%.1f images/s; device calls %.2f s on rank 0)"
            % (len(image_paths), seconds, len(image_paths) / seconds, gpu_seconds))
    if failed:
        why = {1: 'could not be read', 2: 'failed on the device', 3: 'outputs could not be written'}
        log("%d image(s) were NOT processed and are missing from the CSV:" % len(failed))
        for r in failed:
            log("  ", image_paths[int(r[dist.F_INDEX])], "-", why.get(int(r[dist.F_STATUS]), 'failed'))
    return failed


def _supervise_native(n, env, argv=None, poll=0.2, grace=10.0):
    """Start ``n`` fresh rank processes (ECSEG_DIST=native) and watch ALL of them: on the first non-zero exit the remaining
    ranks are terminated (then killed) - a rank that died before the record all-gather would otherwise leave its peers
    blocked in ncclCommInitRank / ncclAllGather until csrc/comm.hip's own timeout (ECSEG_COMM_TIMEOUT_S, default 300 s) and this
    parent on the first live child.  The rendezvous directory is removed either way.  Returns the job's exit code (first failure, else 0).  Only
    fresh children are started, never a re-exec of a process that has touched the GPU."""
    import shutil
    import tempfile
    rdir = tempfile.mkdtemp(prefix='ecseg_rdzv_')
    rdzv = os.path.join(rdir, 'id')
    nonce = os.urandom(dist.NONCE_BYTES).hex()
    cmd = argv or [sys.executable, '-m', 'ecseg_amd.metaseg']
    procs = []
    try:
        for r in range(n):
            procs.append(subprocess.Popen(cmd, env=dict(env, ECSEG_DIST='native', ECSEG_RDZV=rdzv, ECSEG_RDZV_NONCE=nonce,
                                                        RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n))))
        code = 0
        live = list(procs)
        while live and code == 0:
            time.sleep(poll)
            for p in list(live):
                c = p.poll()
                if c is None:
                    continue
                live.remove(p)
                if c != 0 and code == 0:
                    code = abs(c) or 1
                    print("Rank %d exited with code %d: stopping the other %d rank(s)" % (procs.index(p), c, len(live)), flush=True)
        return code
    finally:
        for p in procs:                                    # (normal end: nobody is left; failure / KeyboardInterrupt: tear down)
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + grace
        for p in procs:
            if p.poll() is None:
                try:
                    p.wait(timeout=max(0.1, t_end - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
        shutil.rmtree(rdir, ignore_errors=True)


def _self_launch(device_ids):
    """``device_ids`` with more than one GPU outside a launcher: one rank per listed GPU (this parent never touches HIP)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))          # the ranks import ecseg_amd from here
    # HSA_ENABLE_IPC_MODE_LEGACY=0: RCCL shares device buffers between the rank processes through HIP IPC handles, and this
    # pool's host driver only supports the dmabuf flavour (hipIpcGetMemHandle: invalid argument otherwise)
    env = dict(os.environ, ECSEG_DEVICE_IDS=','.join(str(int(d)) for d in device_ids),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'),
               PYTHONPATH=root + (os.pathsep + os.environ['PYTHONPATH'] if os.environ.get('PYTHONPATH') else ''))
    if dist.want_native():
        # no torch: plain child processes, the library's own RCCL communicator, rendezvous through a file in a private
        # directory (mkdtemp: mode 0700, unpredictable name) + a per-job nonce the readers check
        sys.exit(_supervise_native(len(device_ids), env))
    # --standalone: torch.distributed.run binds its own rendezvous port (no bind-then-close race with other jobs on the node)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           '--nproc-per-node', str(len(device_ids)), '-m', 'ecseg_amd.metaseg']
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main(argv=None):
    config = open("config.yaml")
    var = yaml.load(config, Loader=yaml.FullLoader)['metaseg']
    inpath = var['inpath']

    if not os.path.isdir(os.path.join(inpath)):
        print("Input folder does not exist. Exiting...")
        sys.exit(2)
    # precision: fast (default) = Winograd F(4x4,3x3) where a layer allows it; exact = F(2x2,3x3) everywhere - half as many
    # raw-label pixels away from a float64 evaluation of the network, ~1.7x the U-Net time; split = F(4x4,3x3) with the channel
    # sums on the bf16 matrix pipe and both operands split exactly into three bf16 pieces (float32-accurate, float32 accumulate; round 6:
    # faster than `fast` on models with >= 64 channels per layer, the same accuracy class; opt-in).  emit_probs: also
    # write the stitched float32 probabilities as labels/<stem>_probs.npy (23 MB per 1040 x 1392 image).
    precision = str(var.get('precision', 'fast')).lower()
    if precision not in ('fast', 'exact', 'split'):
        print("metaseg.precision must be 'fast', 'exact' or 'split'. Exiting...")
        sys.exit(2)
    emit_probs = bool(var.get('emit_probs', False))
    for sub in ('dapi', 'labels'):
        os.makedirs(os.path.join(inpath, sub), exist_ok=True)

    under_launcher = int(os.environ.get('WORLD_SIZE', '1')) > 1
    device_ids = var.get('device_ids')
    if device_ids and len(device_ids) > 1 and not under_launcher:
        _self_launch(device_ids)
    # the physical GPU of this rank: entry LOCAL_RANK of the device list (from the self-launching parent's environment, or
    # from config.yaml when an external launcher - torchrun - started the ranks)
    ids = [int(d) for d in os.environ['ECSEG_DEVICE_IDS'].split(',')] if os.environ.get('ECSEG_DEVICE_IDS') else \
        [int(d) for d in device_ids] if device_ids else None
    device = ids[int(os.environ.get('LOCAL_RANK', '0')) % len(ids)] if ids else None
    native = under_launcher and os.environ.get('ECSEG_DIST', '').lower() == 'native' and os.environ.get('ECSEG_RDZV')
    if native:                                                     # RCCL through the C ABI, no torch (rank / world from the environment)
        rank, world = dist.native_init(int(os.environ['RANK']), int(os.environ['WORLD_SIZE']),
                                       device if device is not None else int(os.environ.get('LOCAL_RANK', '0')), os.environ['ECSEG_RDZV'])
    else:
        rank, world = dist.init_process_group(device=device) if under_launcher else (0, 1)     # RCCL on the same physical GPU
    # device_workers (default 1): that many handles - each with its own streams, activation buffers and copy of the weights - on
    # this rank's GPU, fed from one queue of batches (the copies of one call overlap the kernels of another; measured neutral on 256
    # files with the base-16 model - the host side is the limit there - so the default stays 1)
    workers = max(1, min(4, int(var.get('device_workers', 1))))
    models = [load_model(MODEL_NAME, device=device) for _ in range(workers)]
    model = models[0]
    print(model.handle.device_name)
    for m in models:
        m.handle.set_option('winograd', {'exact': 1, 'fast': 2, 'split': 3}[precision])
    image_paths = get_imgs(inpath)
    print("Reading from: ", inpath)
    t0 = time.perf_counter()
    stats = {}
    rec = run(inpath, models if workers > 1 else model, image_paths, rank, world, batch_images=int(var.get('batch_images', 8)),
              io_threads=var.get('io_threads'), stats=stats, resume=bool(var.get('resume', False)), emit_probs=emit_probs,
              pinned_mb=int(var.get('pinned_mb', 2048)))
    failed = finish(inpath, image_paths, rec, rank, seconds=time.perf_counter() - t0, gpu_seconds=stats.get('gpu_seconds', 0.0))
    if native:
        dist.native_close(os.environ['ECSEG_RDZV'], rank)          # (the all-gather was the last thing every rank waited for)
    elif world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()
    if failed:
        sys.exit(1)             # the reference would have crashed on the first such image (src/metaseg.py:42-54)


if __name__ == "__main__":
    main(sys.argv[1:])

"""Image-parallel sharding across the GPUs of one node and the one exchange step of the path: an all-gather of
fixed-size per-image result records (RCCL over xGMI when the tensors live on GPUs; gloo on CPU in the tests).

Two transports carry the exchange: ``torch.distributed`` (bench.py's launcher contract; gloo in the CPU tests) and - so that
`make metaseg` on several GPUs needs no PyTorch at all ("PyTorch only as optional oracle / allocator") - the
library's own RCCL communicator (``_lib.Comm`` -> csrc/comm.hip) with a file rendezvous: rank 0 writes the job's nonce +
the 128-byte communicator id to ``ECSEG_RDZV`` (the self-launcher puts it into a private ``mkdtemp`` directory), the
other ranks pick it up and reject a file that carries another job's nonce (``ECSEG_DIST=native``; chosen by itself when
torch cannot be imported).

The reference's only parallelism is ``tf.distribute.MirroredStrategy`` around ``load_model``
(src/metaseg.py:33-36), which splits one image's patch batch over replicas; images themselves are processed in a
serial loop (src/metaseg.py:42).  Every image is independent, so here whole images are sharded and nothing but the
result records ever crosses devices.
"""
import os

import numpy as np

RECORD_INT64 = 16      # 128 bytes per image
# record layout (int64): [0] global image index (-1 = padding) [1] status (0 ok) [2] n_ec
#                        [3..14] the twelve overlay fields of ecseg_overlay [15] tie-risk pixels (metaseg: pixels whose two
#                        largest quantised probabilities differ by <= 1; 0 where not measured)
F_INDEX, F_STATUS, F_NEC, F_OVERLAY, F_TIE = 0, 1, 2, 3, 15


def shard_bounds(n_items, rank, world):
    """Contiguous blocks of ceil(n/world) items in sorted order: -> (start, stop, padded_len)."""
    per = -(-n_items // world) if n_items else 0
    start = min(rank * per, n_items)
    stop = min(start + per, n_items)
    return start, stop, per


def make_records(start, n_local, padded_len, n_ec=None, overlay=None, status=None, tie_risk=None):
    rec = np.zeros((padded_len, RECORD_INT64), np.int64)
    rec[:, F_INDEX] = -1
    rec[:n_local, F_INDEX] = np.arange(start, start + n_local)
    if n_ec is not None:
        rec[:n_local, F_NEC] = np.asarray(n_ec, np.int64)[:n_local]
    if overlay is not None:
        rec[:n_local, F_OVERLAY:F_OVERLAY + 12] = np.asarray(overlay, np.int64)[:n_local]
    if status is not None:
        rec[:n_local, F_STATUS] = np.asarray(status, np.int64)[:n_local]
    if tie_risk is not None:
        rec[:n_local, F_TIE] = np.asarray(tie_risk, np.int64)[:n_local]
    return rec


def init_process_group(backend=None, device=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).  ``device``: the physical
    GPU index this rank computes on (default LOCAL_RANK) - the RCCL communicator is bound to the same device."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world == 1 and not os.environ.get('ECSEG_FORCE_PROCESS_GROUP'):
        return 0, 1
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    kw = {}
    if backend == 'nccl':
        local = int(os.environ.get('LOCAL_RANK', rank)) if device is None else int(device)
        torch.cuda.set_device(local)
        try:
            kw['device_id'] = torch.device('cuda', local)
        except Exception:
            kw = {}
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world


def allgather_records(records):
    """records: torch int64 tensor (padded_len, RECORD_INT64) on this rank's device (equal shape on all ranks).
    One collective; returns (world * padded_len, RECORD_INT64) on the same device, rank-major."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return records
    world = dist.get_world_size()
    out = torch.empty((world * records.shape[0], records.shape[1]), dtype=records.dtype, device=records.device)
    dist.all_gather_into_tensor(out, records.contiguous())
    return out


def compact_records(gathered):
    """Drop padding rows and order by global image index -> numpy (n_images, RECORD_INT64)."""
    g = gathered.cpu().numpy() if hasattr(gathered, 'cpu') else np.asarray(gathered)
    g = g[g[:, F_INDEX] >= 0]
    return g[np.argsort(g[:, F_INDEX], kind='stable')]


# ---- the library's own communicator (no torch) ----------------------------------------------------------------------
_native = None          # _lib.Comm of this process, once native_init has run


def want_native():
    """ECSEG_DIST=native | torch; default: torch when it can be imported (the tested launcher path), else native."""
    mode = os.environ.get('ECSEG_DIST', '').lower()
    if mode in ('native', 'torch'):
        return mode == 'native'
    try:
        import torch.distributed    # noqa: F401
        return False
    except Exception:
        return True


NONCE_BYTES = 16


def job_nonce():
    """The per-job nonce every rank got from its launcher (``ECSEG_RDZV_NONCE``, hex); empty when the launcher set none."""
    h = os.environ.get('ECSEG_RDZV_NONCE', '')
    try:
        b = bytes.fromhex(h)
    except ValueError:
        b = h.encode()
    return (b + b'\0' * NONCE_BYTES)[:NONCE_BYTES] if b else b''


def write_rendezvous(path, payload, nonce=b''):
    """Rank 0: publish nonce + communicator id atomically.  The temporary file is created exclusively with mode 0600 (no
    following of a symlink somebody planted at a predictable name) and renamed over ``path``: a reader never sees half a
    file, and a stale file of a crashed job is replaced, never appended to."""
    tmp = '%s.tmp%d' % (path, os.getpid())
    try:
        os.unlink(tmp)
    except OSError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, 'O_NOFOLLOW', 0), 0o600)
    with os.fdopen(fd, 'wb') as f:
        f.write(nonce + payload)
    os.replace(tmp, path)


def read_rendezvous(path, nbytes, timeout=300.0, poll=0.05, nonce=b'', alive=None):
    """Other ranks: wait for rank 0's file.  A file whose leading nonce is not this job's (left behind by a crashed job at a
    user-supplied ``ECSEG_RDZV``) is ignored until rank 0 has replaced it.  ``alive()`` (optional) lets the caller give up
    early, e.g. when its launcher has gone."""
    import time
    t0 = time.time()
    while True:
        try:
            with open(path, 'rb') as f:
                b = f.read()
            if len(b) == len(nonce) + nbytes and b[:len(nonce)] == nonce:
                return b[len(nonce):]
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError('no communicator id of this job at %s after %.0f s (did rank 0 start?)' % (path, timeout))
        if alive is not None and not alive():
            raise RuntimeError('the launcher of this rank has gone: giving up the rendezvous at %s' % path)
        time.sleep(poll)


def native_init(rank, world, device, rdzv_path):
    """One RCCL communicator over the ranks of this job without torch: rank 0 creates the id and publishes it at
    ``rdzv_path`` (a file every rank can see), everybody joins.  -> (rank, world)."""
    global _native
    from ._lib import Comm
    nonce = job_nonce()
    if rank == 0:
        uid = Comm.unique_id()
        write_rendezvous(rdzv_path, uid, nonce)
    else:
        ppid = os.getppid()
        # (a rank whose launcher died is re-parented: stop polling instead of spinning for the whole timeout)
        uid = read_rendezvous(rdzv_path, 128, nonce=nonce,
                              alive=(lambda: os.getppid() == ppid) if os.environ.get('ECSEG_RDZV_NONCE') else None)
    _native = Comm(uid, rank, world, device)
    return rank, world


def native_close(rdzv_path=None, rank=0):
    global _native
    if _native is not None:
        _native.close()
        _native = None
    if rdzv_path and rank == 0:
        try:
            os.unlink(rdzv_path)
        except OSError:
            pass


def gather_all(rec, device=0):
    """``rec``: this rank's int64 (padded_len, RECORD_INT64) numpy block -> the whole job's records, compacted (padding rows
    dropped, ordered by global image index), over whichever transport this process initialised; 1 rank: just compacted."""
    if _native is not None:
        return compact_records(_native.allgather_records(rec))
    try:
        import torch
        import torch.distributed as dist
        if dist.is_initialized():
            dev = torch.device('cuda', device) if torch.cuda.is_available() else torch.device('cpu')
            return compact_records(allgather_records(torch.from_numpy(np.ascontiguousarray(rec)).to(dev)))
    except ImportError:
        pass
    return compact_records(rec)

"""Mirror of the metaseg part of the reference's ``src/utils.py``: ``load_model``, ``get_imgs``, ``meta_segment``,
``save_img``, ``read_seg`` with the same names and argument meaning."""
import glob
import os

import numpy as np

from . import image_io, image_tools
from .model import MetasegModel


def load_model(model_name, device=None):
    """src/utils.py:27-33: ``models/<name>`` (``interseg_models/`` for the interSeg classifiers), relative to the CWD."""
    folder = 'interseg_models' if model_name in ('interseg', 'ecseg_c') else 'models'
    path = os.path.join(folder, model_name)
    if not os.path.exists(path) and os.path.exists(path + '.h5'):      # interseg_models/interseg -> interseg.h5
        path += '.h5'
    dev = int(os.environ.get('ECSEG_DEVICE', os.environ.get('LOCAL_RANK', '0'))) if device is None else device
    model = MetasegModel.from_h5(path, device=dev)
    image_tools.set_default_handle(model.handle)
    return model


def get_imgs(inpath):
    """src/utils.py:105-107 (glob ``*.tif`` + ``*.npy``).  The reference keeps the OS directory order; sorted here so
    that sharding across GPUs is deterministic."""
    return sorted(glob.glob(os.path.join(inpath, '*.tif'))) + sorted(glob.glob(os.path.join(inpath, '*.npy')))


def save_img(I, path, folder):
    """src/utils.py:122-123: ``cv2.imwrite(join(path[0], folder, path[1]), I)`` for an 8-bit gray image."""
    out = os.path.join(path[0], folder, path[1])
    if out.lower().endswith(('.tif', '.tiff')):
        image_io.write_tiff_gray8(out, I)
    elif out.lower().endswith('.png'):
        image_io.write_png(out, I)
    else:
        np.save(out, I)


def meta_segment(model, image_path):
    """src/utils.py:109-120 for one image: read, pre-process, write ``dapi/<name>``, segment -> int64 labels."""
    I = image_io.imread(image_path)
    gray = image_tools.meta_preprocess(I, handle=model.handle)
    save_img(~gray, os.path.split(image_path), 'dapi')
    post, _ = model.segment(gray)
    return post.astype(np.int64)


def read_seg(image_path):
    """src/utils.py:125-132."""
    path_split = os.path.split(image_path)
    seg_I = np.load(os.path.join(path_split[0], 'labels', path_split[1][:-4] + '.npy'))
    return seg_I == 0, seg_I == 1, seg_I == 2, seg_I == 3


_allocator_tuned = False


def tune_host_allocator():
    """The file pipelines of `make metaseg` / `make meta_overlay` allocate and free multi-megabyte buffers (decoded images,
    11.6 MB int64 label arrays, batch stacks) from dozens of I/O threads.  With glibc's defaults every such buffer is its own
    mmap / munmap in a per-thread arena: the threads then serialise on the process-wide address-space lock and on page faults
    of fresh memory - measured on the build host (8 cores, tools/host_scaling.py): 48 images in 4.1 s with 5.1 s of system
    time; one arena + a 32 MB mmap threshold + no trimming: 1.6 s with 0.6 s of system time (EXPERIMENTS.md 6).  Called once per
    process before the I/O threads start; ECSEG_HOST_MALLOC=default leaves the allocator alone."""
    global _allocator_tuned
    if _allocator_tuned or os.environ.get('ECSEG_HOST_MALLOC', '').lower() == 'default':
        return False
    _allocator_tuned = True
    try:
        import ctypes
        libc = ctypes.CDLL('libc.so.6', use_errno=True)
        M_TRIM_THRESHOLD, M_TOP_PAD, M_MMAP_THRESHOLD, M_ARENA_MAX = -1, -2, -3, -8
        ok = libc.mallopt(M_ARENA_MAX, 1)
        ok &= libc.mallopt(M_MMAP_THRESHOLD, 32 * 1024 * 1024)
        ok &= libc.mallopt(M_TRIM_THRESHOLD, 2 ** 31 - 1)
        ok &= libc.mallopt(M_TOP_PAD, 256 * 1024 * 1024)
        return bool(ok)
    except Exception:                                       # not glibc: nothing to tune
        return False


def host_cpu_budget():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a 256-core box inside a 16-CPU quota
    runs 16 busy threads at full speed and 128 at an eighth of it: tools/host_scaling.py measured 4 ranks x 32 I/O threads at
    HALF the images/s of one rank on such a box)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q = open('/sys/fs/cgroup/cpu.max').read().split()                  # cgroup v2: "<quota|max> <period>"
        if q[0] != 'max':
            n = min(n, max(1, int(int(q[0]) / int(q[1]))))
    except (OSError, ValueError, IndexError):
        try:
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


def default_io_threads(world=1):
    """Decoder / encoder threads per rank: twice the host's CPU budget (the threads also wait on files) shared between the
    ranks of the node, 2 ... 32."""
    return max(2, min(32, 2 * host_cpu_budget() // max(world, 1)))

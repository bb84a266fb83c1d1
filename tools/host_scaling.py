#!/usr/bin/env python3
"""Host-pipeline headroom of `make metaseg` for an 8-GPU node (no GPU needed): ``ecseg_amd.metaseg.run``
with a STUB model (a few microseconds of numpy per image instead of the U-Net) over N generated 1040 x 1392 RGB LZW TIFF files,
once as 1 rank and once as R gloo ranks sharing this host - TIFF decode, dapi/*.tif + labels/*.png + int64 labels/*.npy
(11.6 MB each, the reference's contract: src/metaseg.py:53) encode / write, records, all-gather, CSV.  What the ranks
reach together is the ceiling the host side puts on the node, whatever the GPUs do.  One JSON line.

    python tools/host_scaling.py [--n 2048] [--ranks 8] [--work /tmp/ecseg_host_scaling]
"""
import argparse
import json
import os
import shutil
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class StubHandle:
    device = 0

    def __init__(self):
        self.images_per_group = 0

    def set_images_per_group(self, n):
        self.images_per_group = int(n)

    def preprocess(self, imgs):
        a = np.asarray(imgs)
        return np.ascontiguousarray(a[..., 2] if a.ndim == 4 else a, np.uint8), np.zeros(len(a), np.int32)

    def count_cc(self, masks):
        m = np.asarray(masks)
        return np.zeros(len(m), np.int32), np.zeros(len(m), np.int64)


class StubModel:
    def __init__(self):
        self.handle = StubHandle()

    def segment(self, gray):
        g = np.asarray(gray)
        return (g >> 6).astype(np.uint8), np.zeros(len(g), np.int32)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, folder, batch, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    import torch
    from ecseg_amd import dist as edist
    from ecseg_amd import metaseg
    from ecseg_amd.utils import get_imgs
    torch.set_num_threads(1)
    r, w = edist.init_process_group('gloo')
    paths = get_imgs(folder)
    torch.distributed.barrier()
    t0 = time.perf_counter()
    rec = metaseg.run(folder, StubModel(), paths, r, w, batch_images=batch, log=lambda *a: None)
    t_run = time.perf_counter() - t0                                # includes the all-gather: the slowest rank's time
    failed = metaseg.finish(folder, paths, rec, r, log=lambda *a: None)
    q.put((rank, t_run, len(rec), len(failed)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _clean(folder):
    for sub in ('dapi', 'labels'):
        shutil.rmtree(os.path.join(folder, sub), ignore_errors=True)
        os.makedirs(os.path.join(folder, sub))


def _out_bytes(folder):
    tot = 0
    for sub in ('dapi', 'labels'):
        d = os.path.join(folder, sub)
        tot += sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d))
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=2048)
    ap.add_argument('--ranks', type=int, default=8)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--work', default='/tmp/ecseg_host_scaling')
    ap.add_argument('--keep', action='store_true')
    a = ap.parse_args()
    from PIL import Image
    import torch.multiprocessing as mp
    from ecseg_amd import metaseg, synth
    from ecseg_amd.utils import get_imgs
    folder = os.path.join(a.work, 'images')
    os.makedirs(folder, exist_ok=True)
    have = len([f for f in os.listdir(folder) if f.endswith('.tif')])
    if have != a.n:
        base = [synth.dapi_image(600 + i, rgb=True) for i in range(8)]
        for i in range(a.n):
            img = np.roll(base[i % 8], (31 * (i // 8) % 1040, 17 * (i // 8) % 1392), axis=(0, 1))
            Image.fromarray(img).save(os.path.join(folder, 'img%05d.tif' % i), compression='tiff_lzw')
    in_bytes = sum(os.path.getsize(os.path.join(folder, f)) for f in os.listdir(folder) if f.endswith('.tif'))
    out = {'images': a.n, 'image_size': [1040, 1392, 3], 'host_cpus': os.cpu_count(), 'input_GB': round(in_bytes / 1e9, 2),
           'model': 'stub (numpy shift; no device work)', 'batch_images': a.batch, 'runs': []}
    # 1 rank, in this process
    _clean(folder)
    paths = get_imgs(folder)
    t0 = time.perf_counter()
    rec = metaseg.run(folder, StubModel(), paths, 0, 1, batch_images=a.batch, log=lambda *x: None)
    t1 = time.perf_counter() - t0
    metaseg.finish(folder, paths, rec, 0, log=lambda *x: None)
    ob = _out_bytes(folder)
    out['runs'].append({'ranks': 1, 'seconds': round(t1, 2), 'images_per_s': round(a.n / t1, 1), 'images_per_s_per_rank': round(a.n / t1, 1),
                        'output_GB': round(ob / 1e9, 2), 'output_GB_per_s': round(ob / 1e9 / t1, 2)})
    # R gloo ranks on the same host
    _clean(folder)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, a.ranks, port, folder, a.batch, q)) for r in range(a.ranks)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=3600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    tR = max(r[1] for r in res)
    ob = _out_bytes(folder)
    out['runs'].append({'ranks': a.ranks, 'seconds': round(tR, 2), 'images_per_s': round(a.n / tR, 1),
                        'images_per_s_per_rank': round(a.n / tR / a.ranks, 1), 'output_GB': round(ob / 1e9, 2),
                        'output_GB_per_s': round(ob / 1e9 / tR, 2), 'failed': res[0][3]})
    out['scaling_vs_1_rank'] = round((a.n / tR) / (a.n / t1), 2)
    print(json.dumps(out))
    if not a.keep:
        shutil.rmtree(a.work, ignore_errors=True)


if __name__ == '__main__':
    main()

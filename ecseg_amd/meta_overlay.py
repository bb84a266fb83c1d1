#!/usr/bin/env python3
"""``make meta_overlay``: drop-in for the reference's ``src/meta_overlay.py`` (config section, checks, exit codes,
``red/`` ``green/`` images and ``fish_quantification.csv``), with the nine counts computed on MI355X."""
import os
import sys

import numpy as np
import yaml

from . import csvio, image_io, image_tools
from .utils import default_io_threads, get_imgs, tune_host_allocator

HSR_SIZE_THRESHOLD = 20


def main(argv=None):
    config = open("config.yaml")
    var = yaml.load(config, Loader=yaml.FullLoader)['meta_overlay']
    inpath = var['inpath']
    sensitivity = var['color_sensitivity']

    if not os.path.isdir(os.path.join(inpath)):
        print("Input folder does not exist. Exiting...")
        sys.exit(2)
    else:
        if not os.path.isdir(os.path.join(inpath, 'labels')):
            print("`labels` folder is missing in the input folder.")
            print("Please make sure metaseg was run on the input folder first. This will generate the labels folder.")
            sys.exit(2)
        if not os.path.isdir(os.path.join(inpath, 'dapi')):
            print("`dapi` folder is missing in the input folder.")
            print("Please make sure metaseg was run on the input folder first. This will generate the labels folder.")
            sys.exit(2)
        if (sensitivity < 0) | (sensitivity > 255):
            print("color_sensitivity can only be between 0 and 255. Please update the config.yaml file accordingly.")
            sys.exit(2)
    for sub in ('red', 'green'):
        os.makedirs(os.path.join(inpath, sub), exist_ok=True)

    handle = image_tools.default_handle()
    rows = run(inpath, handle, get_imgs(inpath), int(sensitivity), batch_images=int(var.get('batch_images', 8)),
               io_threads=var.get('io_threads'))
    with open(os.path.join(inpath, 'fish_quantification.csv'), 'w') as f:
        f.write(csvio.csv_text(csvio.OVERLAY_COLUMNS, rows))


def _load(p):
    """Decode one input image and the labels metaseg wrote for it (reader thread)."""
    path_split = os.path.split(p)
    I = image_io.imread(p)
    if I.ndim < 3:
        return I, None
    return I, image_io.read_npy_labels_u8(os.path.join(path_split[0], 'labels', path_split[1][:-4] + '.npy'))


def _write_channels(p, I8):
    path_split = os.path.split(p)
    image_io.write_png_channel(os.path.join(path_split[0], 'red', path_split[1] + '.png'), I8, 0, invert=True)
    image_io.write_png_channel(os.path.join(path_split[0], 'green', path_split[1] + '.png'), I8, 1, invert=True)


def run(inpath, handle, image_paths, sensitivity, batch_images=8, io_threads=None, log=print, stats=None):
    """The per-image loop of src/meta_overlay.py:56-95 with the decoding (TIFF + labels/*.npy) and the red / green PNG
    encoders on worker threads and the nine counts of consecutive same-shaped images computed in one device call.  Rows
    come back in the order of ``image_paths``."""
    import concurrent.futures as cf
    import threading
    import time
    t_stage = {'read': 0.0, 'write': 0.0, 'pack': 0.0, 'device': 0.0}      # thread-seconds per host stage (stats)
    t_lock = threading.Lock()

    def timed(stage, fn):
        def wrapped(*a):
            t0 = time.perf_counter()
            try:
                return fn(*a)
            finally:
                dt = time.perf_counter() - t0
                with t_lock:
                    t_stage[stage] += dt
        return wrapped

    load_fn, write_fn = timed('read', _load), timed('write', _write_channels)
    io_threads = io_threads or default_io_threads(1)
    window = max(2 * batch_images, io_threads)
    rows = [None] * len(image_paths)
    tune_host_allocator()
    with cf.ThreadPoolExecutor(io_threads) as readers, cf.ThreadPoolExecutor(io_threads) as writers:
        reads, next_submit, writes = {}, 0, []

        def flush(group):
            if not group:
                return
            t0 = time.perf_counter()
            rgb = np.stack([g[1] for g in group])
            seg = np.stack([g[2] for g in group])
            t1 = time.perf_counter()
            I8 = image_tools.u16_to_u8(rgb, handle=handle)   # src/image_tools.py:142
            rec = handle.overlay(seg, np.ascontiguousarray(I8), sensitivity, HSR_SIZE_THRESHOLD)
            t_stage['pack'] += t1 - t0
            t_stage['device'] += time.perf_counter() - t1
            for j, (k, _, _) in enumerate(group):
                writes.append(writers.submit(write_fn, image_paths[k], I8[j]))
                rows[k] = [os.path.split(image_paths[k])[1]] + csvio.overlay_cells(rec[j])

        group, key = [], None
        for k, p in enumerate(image_paths):
            while next_submit < min(len(image_paths), k + window):
                log("Processing image: ", image_paths[next_submit])
                reads[next_submit] = readers.submit(load_fn, image_paths[next_submit])
                next_submit += 1
            I, seg = reads.pop(k).result()
            if seg is None:
                # the reference prints this and then crashes on the tuple unpack (src/meta_overlay.py:60); skipping is the
                # evident intent
                log(p, " isn't an RGB image. Therefore, no FISH signals could be identified. Skipping...")
                continue
            kk = (I.shape, I.dtype.str, seg.shape)
            if group and (kk != key or len(group) >= batch_images):
                flush(group)
                group = []
            group.append((k, I, seg))
            key = kk
        flush(group)
        for f in writes:
            f.result()
    if stats is not None:
        stats.update({'%s_thread_seconds' % k: v for k, v in t_stage.items()})
    return [r for r in rows if r is not None]


if __name__ == "__main__":
    main(sys.argv[1:])

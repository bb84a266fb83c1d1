#!/usr/bin/env python3
"""``make metaseg``: drop-in for the reference's ``src/metaseg.py`` (same config.yaml section, messages, exit codes and
output files), running on MI355X.

Differences that do not change results: images are processed in batches on the GPU instead of one by one, in sorted
order; under ``torchrun`` the images are sharded across the GPUs of the node, every rank writes the outputs of its
own images and rank 0 writes ``ec_quantification.csv`` after one all-gather of the per-image records.
"""
import os
import sys

import numpy as np
import yaml

from . import csvio, dist, image_io
from .utils import get_imgs, load_model, save_img

MODEL_NAME = 'metaseg.h5'


def _batches(items, key, max_batch):
    """Consecutive runs of items with equal key, cut at max_batch."""
    out, cur, cur_key = [], [], None
    for it in items:
        k = key(it)
        if cur and (k != cur_key or len(cur) >= max_batch):
            out.append(cur)
            cur = []
        cur.append(it)
        cur_key = k
    if cur:
        out.append(cur)
    return out


def run(inpath, model, image_paths, rank=0, world=1, batch_images=8, log=print):
    """Segment this rank's shard; returns records (one row per image of the WHOLE job after the all-gather)."""
    start, stop, per = dist.shard_bounds(len(image_paths), rank, world)
    mine = image_paths[start:stop]
    n_ec = np.zeros(len(mine), np.int64)
    status = np.zeros(len(mine), np.int64)
    loaded = []
    for k, p in enumerate(mine):
        log("Processing image: ", p)
        try:
            img = image_io.imread(p)
            if img.dtype not in (np.uint8, np.uint16) or img.ndim not in (2, 3):
                raise ValueError('unsupported image array %s %s' % (img.dtype, img.shape))
            loaded.append((k, p, img))
        except Exception as e:                     # a corrupt image must not take the shard down
            log("Skipping %s: %s" % (p, e))
            status[k] = 1
    for group in _batches(loaded, lambda t: (t[2].shape, t[2].dtype.str), batch_images):
        imgs = np.stack([g[2] for g in group])
        try:
            gray, _ = model.handle.preprocess(imgs)
            post, nec = model.segment(gray)
        except Exception as e:
            log("Skipping %d image(s) of shape %s: %s" % (len(group), imgs.shape[1:], e))
            for k, _, _ in group:
                status[k] = 2
            continue
        for j, (k, p, _) in enumerate(group):
            path_split = os.path.split(p)
            save_img(~gray[j], path_split, 'dapi')                       # cv2.bitwise_not(I) (src/utils.py:112)
            outpath = os.path.join(path_split[0], 'labels', path_split[1][:-4])
            log("Saving labels: ", p, " to ", outpath)
            image_io.write_label_png(outpath + '.png', post[j])
            np.save(outpath, post[j].astype(np.int64))                     # int64 .npy (src/metaseg.py:53)
            n_ec[k] = int(nec[j])
    rec = dist.make_records(start, len(mine), per, n_ec=n_ec, status=status)
    if world > 1:
        import torch
        dev = torch.device('cuda', model.handle.device) if torch.cuda.is_available() else torch.device('cpu')
        rec = dist.compact_records(dist.allgather_records(torch.from_numpy(rec).to(dev)))
    else:
        rec = dist.compact_records(rec)
    return rec


def main(argv=None):
    config = open("config.yaml")
    var = yaml.load(config, Loader=yaml.FullLoader)['metaseg']
    inpath = var['inpath']

    if not os.path.isdir(os.path.join(inpath)):
        print("Input folder does not exist. Exiting...")
        sys.exit(2)
    for sub in ('dapi', 'labels'):
        os.makedirs(os.path.join(inpath, sub), exist_ok=True)

    rank, world = dist.init_process_group() if int(os.environ.get('WORLD_SIZE', '1')) > 1 else (0, 1)
    model = load_model(MODEL_NAME)
    print(model.handle.device_name)
    image_paths = get_imgs(inpath)
    print("Reading from: ", inpath)
    rec = run(inpath, model, image_paths, rank, world, batch_images=int(var.get('batch_images', 8)))
    if rank == 0:
        rows = [[os.path.split(image_paths[int(r[dist.F_INDEX])])[1], int(r[dist.F_NEC])]
                for r in rec if r[dist.F_STATUS] == 0]
        out = os.path.join(inpath, 'ec_quantification.csv')
        print("Saving ec quantification to", out)
        text = csvio.csv_text(csvio.METASEG_COLUMNS, rows)
        with open(out, 'w') as f:
            f.write(text)
        with open(os.path.join(inpath, 'ec_quantifications.csv'), 'w') as f:   # the name README.md:86 uses
            f.write(text)
    if world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])

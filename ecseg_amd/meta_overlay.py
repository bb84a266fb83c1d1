#!/usr/bin/env python3
"""``make meta_overlay``: drop-in for the reference's ``src/meta_overlay.py`` (config section, checks, exit codes,
``red/`` ``green/`` images and ``fish_quantification.csv``), with the nine counts computed on MI355X."""
import os
import sys

import numpy as np
import yaml

from . import csvio, image_io, image_tools
from .utils import get_imgs

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
    rows = []
    for p in get_imgs(inpath):
        path_split = os.path.split(p)
        print("Processing image: ", p)
        I = image_io.imread(p)
        if I.ndim < 3:
            # the reference prints this and then crashes on the tuple unpack (src/meta_overlay.py:60); skipping is the
            # evident intent
            print(p, " isn't an RGB image. Therefore, no FISH signals could be identified. Skipping...")
            continue
        I8 = image_tools.u16_to_u8(I, handle=handle)   # src/image_tools.py:142
        image_io.write_png(os.path.join(path_split[0], 'red', path_split[1] + '.png'), ~np.uint8(I8[..., 0]))
        image_io.write_png(os.path.join(path_split[0], 'green', path_split[1] + '.png'), ~np.uint8(I8[..., 1]))
        seg = np.load(os.path.join(path_split[0], 'labels', path_split[1][:-4] + '.npy'))
        rec = handle.overlay(seg.astype(np.uint8), np.ascontiguousarray(I8), int(sensitivity), HSR_SIZE_THRESHOLD)
        rows.append([path_split[1]] + csvio.overlay_cells(rec))
    with open(os.path.join(inpath, 'fish_quantification.csv'), 'w') as f:
        f.write(csvio.csv_text(csvio.OVERLAY_COLUMNS, rows))


if __name__ == "__main__":
    main(sys.argv[1:])

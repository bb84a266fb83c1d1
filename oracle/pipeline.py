"""Oracle: the whole ``meta_segment`` + count flow on the CPU.  TEST INFRASTRUCTURE ONLY.

Restates src/utils.py:109-120 and src/metaseg.py:45-46 by chaining the other oracle modules.
"""
import numpy as np

from . import postproc, preprocess, quant, tiling, unet


def raw_labels_from_probs(preds, pos):
    """stitch -> img_as_ubyte -> argmax (src/utils.py:116-118)."""
    return quant.quantised_argmax(tiling.stitch(preds, pos))


def segment_gray(model_config, weights, gray_u8, batch=8, return_intermediate=False):
    """``gray_u8`` is the pre-processed (H, W) uint8 image (output of meta_preprocess).
    Returns post-processed int64 labels (and optionally probs / raw labels)."""
    H, W = gray_u8.shape
    pos = tiling.patch_positions(H, W)
    patches = tiling.extract_patches(gray_u8[..., None], pos)              # (n, 256, 256, 1) uint8
    preds = np.concatenate([unet.forward(model_config, weights, patches[i:i + batch])
                            for i in range(0, len(patches), batch)])
    raw = raw_labels_from_probs(preds, pos)
    post = postproc.meta_inference(raw)
    if return_intermediate:
        return post, raw, preds, pos
    return post


def meta_segment_array(model_config, weights, image):
    """src/utils.py:109-120 on an already decoded image array."""
    return segment_gray(model_config, weights, preprocess.meta_preprocess(image))


def num_ecdna(labels):
    """src/metaseg.py:46."""
    return postproc.count_cc(labels == 3)[0]

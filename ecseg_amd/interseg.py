"""The classifier step of the reference's interSeg task (``make interseg``, src/interseg.py) on MI355X: the two Keras
classifiers ``interseg_models/interseg`` (3 classes: No-amp / EC-amp / HSR-amp) and ``interseg_models/ecseg_c`` (focal
amplification, one sigmoid unit) run through the same plan interpreter and HIP kernels as the metaseg U-Net, and the
per-nucleus decision logic of src/interseg.py:131-190 is applied to their outputs.

Scope: this is the reuse of the kernels for the classifier graphs.  The file-level driver of
``make interseg`` consumes the outputs of ``make stat_fish`` (NuSeT segmentation, out of scope) and skimage's
``resize`` - it is not re-built here; callers hand in the (N, 256, 256, 3) uint8 nucleus crops the reference builds at
src/interseg.py:150-152 / 193-194.
"""
import numpy as np

ECSEG_I_MODEL = 'interseg'
ECSEG_C_MODEL = 'ecseg_c'

# src/interseg.py:72-90
ECSEG_I_LABEL_MAP = {0: 'No-amp', 1: 'EC-amp', 2: 'HSR-amp'}
ECSEG_C_LABEL_MAP = {0: 'No-amp', 1: 'Focal-amp'}
INTERSEG_LABEL_MAP = {
    ('No-amp', 'No-amp'): 'No-amp', ('No-amp', 'EC-amp'): 'No-amp', ('No-amp', 'HSR-amp'): 'No-amp',
    ('Focal-amp', 'No-amp'): 'No-amp', ('Focal-amp', 'EC-amp'): 'EC-amp', ('Focal-amp', 'HSR-amp'): 'HSR-amp',
}
EMPTY = 'No_Prediction (Segmentation_Empty)'
FAILED_QUALITY = 'No_Prediction (Failed Centromeric Quality Score)'
LOW_CENT = 'No_Prediction (Low_CENT_Brightness)'


def preprocess_ecseg_c(batch_x):
    """src/utils.py:166-173 for one (H, W, 3) image: per-channel max normalisation (FISH channels 0/1, DAPI channel 2),
    quantised to 1/255 steps: ``round(x / norm * 255) / 255`` (tf.math.round = half to even) in float32."""
    x = np.asarray(batch_x, np.float32)
    norm = x.reshape(-1, x.shape[-1]).max(axis=0).astype(np.float32)          # concat([fish_norm (2), dapi_norm (1)])
    with np.errstate(divide='ignore', invalid='ignore'):
        return (np.rint((x / norm) * np.float32(255)) / np.float32(255)).astype(np.float32)


def classify_crops(ecseg_i_model, crops, ecseg_c_model=None, centromeric_quality_score_pass=True, from_patches=False):
    """``crops``: (N, 256, 256, 3) uint8, channel 0 = target FISH, 1 = centromeric probe, 2 = DAPI (the order built at
    src/interseg.py:118).  Returns one dict per crop with the reference's columns: ecSeg-i probabilities and label, the
    ecSeg-c probabilities and label (when a centromeric-probe model is given) and the merged interSeg label
    (src/interseg.py:153-190).  ``from_patches``: crops come from the tiling of a nucleus larger than 256 px, where the
    reference skips all-zero tiles (src/interseg.py:199-210).  All crops go through the device in one batch; the
    reference calls ``model.predict`` crop by crop, which gives the same numbers (no layer depends on the batch)."""
    crops = np.ascontiguousarray(crops, np.uint8)
    if crops.ndim != 4 or crops.shape[1:] != (256, 256, 3):
        raise ValueError('crops must be (N, 256, 256, 3) uint8')
    n = len(crops)
    has_c = ecseg_c_model is not None
    rows = [dict() for _ in range(n)]
    empty = np.array([from_patches and not c.any() for c in crops], bool)
    live = np.flatnonzero(~empty)
    for k in np.flatnonzero(empty):
        rows[k] = {'interSeg_label': EMPTY, 'ecSeg-i_label': EMPTY, 'pred_no_amp': EMPTY, 'pred_ec': EMPTY, 'pred_hsr': EMPTY}
        if has_c:
            rows[k].update({'ecSeg-c_label': EMPTY, 'pred_no_focal_amp': EMPTY, 'pred_focal_amp': EMPTY})
    if len(live):
        pi = ecseg_i_model.predict(crops[live][..., 0])                        # (n, 3): src/interseg.py:155
        run_c = np.array([has_c and centromeric_quality_score_pass and crops[k][..., 1].max() > 10 for k in live], bool)
        pc = None
        if run_c.any():
            xc = np.stack([preprocess_ecseg_c(crops[k]) for k in live[run_c]])
            pc = ecseg_c_model.predict(xc).reshape(-1)                          # (n,): src/interseg.py:168
        j = 0
        for a, k in enumerate(live):
            r = rows[k]
            # np.float32 scalars, as the reference keeps them (src/interseg.py:156-158,169-170): a CSV written from these
            # rows prints 0.3, not float64(float32(0.3)) = 0.30000001192092896
            r['pred_no_amp'], r['pred_ec'], r['pred_hsr'] = (np.float32(v) for v in pi[a])
            i_label = ECSEG_I_LABEL_MAP[int(np.argmax(pi[a]))]
            r['ecSeg-i_label'] = i_label
            if run_c[a]:
                v = np.float32(pc[j]); j += 1
                r['pred_no_focal_amp'], r['pred_focal_amp'] = np.float32(1) - v, v        # float32 complement
                c_label = ECSEG_C_LABEL_MAP[int(v > 0.5)]
                r['ecSeg-c_label'] = c_label
                r['interSeg_label'] = INTERSEG_LABEL_MAP[(c_label, i_label)]
            else:
                if has_c and not centromeric_quality_score_pass:
                    r['ecSeg-c_label'] = r['pred_no_focal_amp'] = r['pred_focal_amp'] = FAILED_QUALITY
                elif has_c:
                    r['ecSeg-c_label'] = r['pred_no_focal_amp'] = r['pred_focal_amp'] = LOW_CENT
                r['interSeg_label'] = i_label
    return rows

"""Host-side model object: the drop-in for the Keras model that ``load_model('metaseg.h5')`` returns in the
reference (src/utils.py:27-33, used at src/utils.py:115 and src/metaseg.py:33-45)."""
import json
import os

import numpy as np

from . import hdf5_min, keras_plan
from ._lib import EcsegError, Handle


class MetasegModel:
    """Holds one GPU handle with the lowered plan loaded.

    ``predict_on_batch(uint8[N,256,256,1]) -> float32[N,256,256,4]`` is the call shape of the Keras model
    (src/utils.py:115); ``segment`` runs the whole device pipeline of ``meta_segment`` (src/utils.py:113-119)."""

    def __init__(self, model_config, weights, device=0, fuse=True, handle=None, lambda_overrides=None, output=0):
        """``output``: index or layer name of the output to compute when the Keras model has several (default: the first)."""
        if isinstance(model_config, (str, bytes)):
            model_config = json.loads(model_config)
        self.model_config = model_config
        self.weights = weights
        self.plan = keras_plan.build_plan(model_config, weights, fuse=fuse, lambda_overrides=lambda_overrides, output=output)
        self.handle = handle if handle is not None else Handle(device)
        self.handle.load_plan(self.plan)

    @classmethod
    def from_h5(cls, path, device=0, fuse=True, lambda_overrides=None, handle=None, output=0):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        if os.path.isdir(path):
            raise ValueError('%s is a TensorFlow SavedModel directory; only the Keras HDF5 format is read here - '
                             're-save it with model.save("<name>.h5") where TensorFlow is available' % path)
        cfg, weights = hdf5_min.load_keras_h5(path)
        return cls(cfg, weights, device=device, fuse=fuse, lambda_overrides=lambda_overrides, handle=handle, output=output)

    # Keras call shapes -----------------------------------------------------------------------------
    def predict_on_batch(self, x):
        x = np.asarray(x)
        if self.plan.channels_first:
            # a channels_first Keras model takes and returns (N, C, H, W): the plan is its channels_last twin (keras_plan.channels_first_to_last)
            y = self.handle.forward_patches(np.ascontiguousarray(np.moveaxis(x, 1, -1)) if x.ndim == 4 else x)
            return np.ascontiguousarray(np.moveaxis(y, -1, 1)) if y.ndim == 4 else y
        return self.handle.forward_patches(x)

    def predict(self, x, batch_size=None, verbose=0):
        return self.predict_on_batch(x)

    def __call__(self, x):
        return self.predict_on_batch(x)

    # Pipeline ---------------------------------------------------------------------------------------
    def segment(self, gray, want_raw=False):
        """gray: (H, W) or (n, H, W) uint8 pre-processed image(s) -> post-processed labels (uint8), n_ec
        (and raw argmax labels when ``want_raw``)."""
        g = np.asarray(gray)
        single = g.ndim == 2
        raw, post, nec = self.handle.segment_images(g, want_raw=want_raw)
        if single:
            out = (post[0], int(nec[0]))
            return out + (raw[0],) if want_raw else out
        return (post, nec, raw) if want_raw else (post, nec)

    def segment_ex(self, gray, want_probs=False):
        """(n, H, W) uint8 -> (post labels, n_ec, tie_risk[, stitched probabilities float32 (n, H, W, 4)]): ``segment`` plus
        the per-image count of pixels whose label a last-bit difference between float32 evaluations can flip (and, on
        request, the probabilities ``np.argmax`` saw - config key ``emit_probs``)."""
        out = self.handle.segment_images(np.asarray(gray), want_raw=False, want_tie_risk=True, want_probs=want_probs)
        return out[1:]

    def flops_per_patch(self):
        return self.handle.flops_per_patch()


__all__ = ['MetasegModel', 'EcsegError']

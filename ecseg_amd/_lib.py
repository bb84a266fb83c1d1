"""ctypes binding of libecseg_hip.so (include/ecseg_hip.h).  There is no CPU fallback: when the library is missing
or no MI355X is visible, every entry point raises."""
import ctypes as C
import os
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# ECSEG_HIP_LIB: another build of the same library (A/B timing of two builds on one GPU box)
LIB_PATH = os.environ.get('ECSEG_HIP_LIB') or os.path.join(HERE, 'libecseg_hip.so')

EXPORTS = [
    'ecseg_abi_version', 'ecseg_create', 'ecseg_destroy', 'ecseg_last_error', 'ecseg_device_name', 'ecseg_stream',
    'ecseg_model_load', 'ecseg_model_flops_per_patch', 'ecseg_forward_patches', 'ecseg_forward_patches_f32', 'ecseg_read_tensor',
    'ecseg_segment_images', 'ecseg_segment_images_ex', 'ecseg_segment_images_dev', 'ecseg_set_images_per_group', 'ecseg_set_option', 'ecseg_preprocess', 'ecseg_u16_to_u8',
    'ecseg_meta_segment', 'ecseg_prefetch_input', 'ecseg_host_alloc', 'ecseg_host_free',
    'ecseg_stitch_argmax', 'ecseg_meta_inference', 'ecseg_meta_inference_dev', 'ecseg_count_cc', 'ecseg_ccl_labels',
    'ecseg_count_colocalization', 'ecseg_count_hsr', 'ecseg_overlay', 'ecseg_get_timings',
    'ecseg_set_kernel_profiling', 'ecseg_get_conv_profile', 'ecseg_get_conv_executed_flops', 'ecseg_get_conv_launch_profile', 'ecseg_debug_peek', 'ecseg_lzw_decode', 'ecseg_lzw_encode',
    'ecseg_comm_unique_id', 'ecseg_comm_create', 'ecseg_comm_destroy', 'ecseg_comm_last_error', 'ecseg_allgather_records', 'ecseg_allgather_records_dev',
    'ecseg_npy_write_i64', 'ecseg_png_write_labels', 'ecseg_png_write', 'ecseg_png_write_channel', 'ecseg_npy_label_info', 'ecseg_npy_read_labels_u8', 'ecseg_tiff_write_gray8', 'ecseg_tiff_info', 'ecseg_tiff_read',
]


class EcsegError(RuntimeError):
    """``code`` carries the ECSEG_E_* status of the failed call (None for binding-level errors)."""
    code = None


E_HIP = -2
E_NOMEM = -4
E_UNSUPPORTED, E_IO = -5, -6
ABI_VERSION = 5           # ECSEG_ABI_VERSION of include/ecseg_hip.h this binding was written for


class TensorDesc(C.Structure):
    _fields_ = [('buffer', C.c_int32), ('h', C.c_int32), ('w', C.c_int32), ('c', C.c_int32),
                ('c_stride', C.c_int32), ('c_offset', C.c_int32)]


class OpDesc(C.Structure):
    _fields_ = [('op', C.c_int32), ('in0', C.c_int32), ('in1', C.c_int32), ('out', C.c_int32),
                ('kh', C.c_int32), ('kw', C.c_int32), ('stride', C.c_int32),
                ('pad_top', C.c_int32), ('pad_left', C.c_int32), ('act', C.c_int32), ('mode', C.c_int32),
                ('w0', C.c_int32), ('w1', C.c_int32), ('alpha', C.c_float), ('dilation', C.c_int32)]


_lib = None


def load_library():
    """Load libecseg_hip.so and declare the prototypes.  Raises EcsegError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EcsegError('%s not found: build it with `python -m ecseg_amd.build` (hipcc, gfx950)' % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise EcsegError('cannot load %s: %s' % (LIB_PATH, e))
    vp, i32, u8p = C.c_void_p, C.c_int, C.c_void_p
    lib.ecseg_abi_version.restype = C.c_int
    if lib.ecseg_abi_version() != ABI_VERSION:
        raise EcsegError('%s has ABI version %d, this binding was written for %d: rebuild it with `python -m ecseg_amd.build --force`'
                         % (LIB_PATH, lib.ecseg_abi_version(), ABI_VERSION))
    lib.ecseg_create.argtypes = [C.POINTER(vp), i32]
    lib.ecseg_destroy.argtypes = [vp]; lib.ecseg_destroy.restype = None
    lib.ecseg_last_error.argtypes = [vp]; lib.ecseg_last_error.restype = C.c_char_p
    lib.ecseg_device_name.argtypes = [vp, C.c_char_p, i32]
    lib.ecseg_stream.argtypes = [vp]; lib.ecseg_stream.restype = vp
    lib.ecseg_model_load.argtypes = [vp, C.POINTER(TensorDesc), i32, i32, C.POINTER(OpDesc), i32,
                                     C.POINTER(C.c_void_p), C.POINTER(C.c_int64), i32, i32, i32]
    lib.ecseg_model_flops_per_patch.argtypes = [vp, C.POINTER(C.c_double)]
    lib.ecseg_forward_patches.argtypes = [vp, u8p, i32, vp]
    lib.ecseg_forward_patches_f32.argtypes = [vp, vp, i32, vp]
    lib.ecseg_read_tensor.argtypes = [vp, i32, i32, vp]
    lib.ecseg_segment_images.argtypes = [vp, u8p, i32, i32, i32, vp, vp, vp]
    lib.ecseg_segment_images_dev.argtypes = [vp, u8p, i32, i32, i32, vp, vp, vp]
    lib.ecseg_segment_images_ex.argtypes = [vp, u8p, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.ecseg_set_images_per_group.argtypes = [vp, i32]
    lib.ecseg_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.ecseg_preprocess.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.ecseg_u16_to_u8.argtypes = [vp, vp, C.c_longlong, vp]
    lib.ecseg_meta_segment.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.ecseg_prefetch_input.argtypes = [vp, vp, C.c_size_t]
    lib.ecseg_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.ecseg_host_free.argtypes = [vp, vp]
    lib.ecseg_stitch_argmax.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.ecseg_meta_inference.argtypes = [vp, u8p, i32, i32, i32, vp, vp]
    lib.ecseg_meta_inference_dev.argtypes = [vp, u8p, i32, i32, i32, vp, vp]
    lib.ecseg_count_cc.argtypes = [vp, u8p, i32, i32, i32, vp, vp]
    lib.ecseg_ccl_labels.argtypes = [vp, u8p, i32, i32, i32, i32, vp]
    lib.ecseg_count_colocalization.argtypes = [vp, u8p, u8p, i32, i32, i32, vp]
    lib.ecseg_count_hsr.argtypes = [vp, u8p, u8p, i32, i32, i32, i32, vp]
    lib.ecseg_overlay.argtypes = [vp, u8p, u8p, i32, i32, i32, i32, i32, i32, vp]
    lib.ecseg_get_timings.argtypes = [vp, vp]
    lib.ecseg_set_kernel_profiling.argtypes = [vp, i32]
    lib.ecseg_get_conv_profile.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    lib.ecseg_get_conv_executed_flops.argtypes = [vp, C.POINTER(C.c_double)]
    lib.ecseg_debug_peek.argtypes = [vp, vp, i32]
    lib.ecseg_get_conv_launch_profile.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    for fn in (lib.ecseg_lzw_decode, lib.ecseg_lzw_encode):
        fn.argtypes = [vp, C.c_longlong, vp, C.c_longlong]
        fn.restype = C.c_longlong
    lib.ecseg_comm_unique_id.argtypes = [vp, i32]
    lib.ecseg_comm_create.argtypes = [C.POINTER(vp), vp, i32, i32, i32]
    lib.ecseg_comm_destroy.argtypes = [vp]; lib.ecseg_comm_destroy.restype = None
    lib.ecseg_comm_last_error.restype = C.c_char_p
    lib.ecseg_allgather_records.argtypes = [vp, vp, i32, vp]
    lib.ecseg_allgather_records_dev.argtypes = [vp, vp, i32, vp, vp]
    lib.ecseg_npy_write_i64.argtypes = [C.c_char_p, vp, i32, i32]
    lib.ecseg_png_write_labels.argtypes = [C.c_char_p, vp, i32, i32]
    lib.ecseg_png_write.argtypes = [C.c_char_p, vp, i32, i32, i32, i32]
    lib.ecseg_png_write_channel.argtypes = [C.c_char_p, vp, i32, i32, i32, i32, i32]
    lib.ecseg_npy_label_info.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i32)]
    lib.ecseg_npy_read_labels_u8.argtypes = [C.c_char_p, vp, i32, i32]
    lib.ecseg_tiff_write_gray8.argtypes = [C.c_char_p, vp, i32, i32, i32]
    lib.ecseg_tiff_info.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.ecseg_tiff_read.argtypes = [C.c_char_p, vp, C.c_longlong]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ('ecseg_abi_version',):
            fn.restype = C.c_int
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(a, shape_tail=None):
    a = np.ascontiguousarray(a)
    if a.dtype == np.bool_:
        a = a.view(np.uint8)
    if a.dtype != np.uint8:
        raise TypeError('expected a uint8 / bool array, got %s' % a.dtype)
    return a


class Handle:
    """One per GPU.  Owns the device context, its stream and all device buffers."""

    T_NAMES = ('tile', 'unet', 'tail', 'post', 'count')

    def __init__(self, device=0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.ecseg_create(C.byref(h), int(device))
        if rc != 0:
            raise EcsegError('ecseg_create(device=%d) failed (%d): %s'
                             % (device, rc, self.lib.ecseg_last_error(None).decode()))
        self.h = h
        self.device = int(device)
        self.plan = None
        self.images_per_group = 0          # 0: automatic (ecseg_set_images_per_group)
        self.native_seconds = 0.0          # time spent inside ecseg_meta_segment (stage report of `make metaseg`)
        self._pinned = {}                  # page-locked host buffers of host_empty: array address -> allocation

    def close(self):
        if getattr(self, 'h', None):
            for p in list(getattr(self, '_pinned', {}).values()):
                self.lib.ecseg_host_free(self.h, C.c_void_p(p))
            self._pinned = {}
            self.lib.ecseg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            e = EcsegError('%s failed (%d): %s' % (what, rc, self.lib.ecseg_last_error(self.h).decode()))
            e.code = rc
            raise e

    @property
    def device_name(self):
        buf = C.create_string_buffer(256)
        self._check(self.lib.ecseg_device_name(self.h, buf, 256), 'ecseg_device_name')
        return buf.value.decode()

    @property
    def stream(self):
        return self.lib.ecseg_stream(self.h)

    # ---- model -----------------------------------------------------------------------------------
    def load_plan(self, plan):
        nt, no, nw = len(plan.tensors), len(plan.ops), len(plan.weights)
        T = (TensorDesc * nt)(*[TensorDesc(t['buffer'], t['h'], t['w'], t['c'], t['c_stride'], t['c_offset'])
                                for t in plan.tensors])
        O = (OpDesc * no)(*[OpDesc(o['op'], o['in0'], o['in1'], o['out'], o['kh'], o['kw'], o['stride'], o['pad_top'],
                                   o['pad_left'], o['act'], o['mode'], o['w0'], o['w1'], float(o['alpha']), int(o.get('dilation', 1)))
                            for o in plan.ops])
        keep = [np.ascontiguousarray(w, np.float32) for w in plan.weights]
        Wp = (C.c_void_p * max(nw, 1))(*[w.ctypes.data for w in keep])
        Wl = (C.c_int64 * max(nw, 1))(*[w.size for w in keep])
        self._check(self.lib.ecseg_model_load(self.h, T, nt, plan.n_buffers, O, no, Wp, Wl, nw, plan.input_tensor,
                                              plan.output_tensor), 'ecseg_model_load')
        self.plan = plan

    def flops_per_patch(self):
        v = C.c_double()
        self._check(self.lib.ecseg_model_flops_per_patch(self.h, C.byref(v)), 'ecseg_model_flops_per_patch')
        return v.value

    def set_images_per_group(self, n):
        self._check(self.lib.ecseg_set_images_per_group(self.h, int(n)), 'ecseg_set_images_per_group')
        self.images_per_group = int(n)

    def set_option(self, key, value):
        self._check(self.lib.ecseg_set_option(self.h, key.encode(), int(value)), 'ecseg_set_option(%s)' % key)
        if key == 'images_per_group':        # the same knob as set_images_per_group: keep the mirror the OOM retry restores from
            self.images_per_group = int(value)

    def forward_patches(self, patches):
        """uint8 (or float32) (N, H, W, C) -> float32 (N, H', W', K): ``model.predict_on_batch`` (reference
        src/utils.py:115; the interSeg classifiers of src/interseg.py:155,168 return (N, K))."""
        if self.plan is None:
            raise EcsegError('no model loaded')
        p = np.ascontiguousarray(patches)
        is_f32 = p.dtype.kind == 'f'
        p = np.ascontiguousarray(p, np.float32) if is_f32 else _u8(p)
        ti, to = self.plan.tensors[self.plan.input_tensor], self.plan.tensors[self.plan.output_tensor]
        if p.ndim == 3:
            p = p[..., None]
        if p.shape[1:] != (ti['h'], ti['w'], ti['c']):
            raise ValueError('expected patches of shape (N, %d, %d, %d), got %s' % (ti['h'], ti['w'], ti['c'], p.shape))
        out = np.empty((p.shape[0], to['h'], to['w'], to['c']), np.float32)
        fn = self.lib.ecseg_forward_patches_f32 if is_f32 else self.lib.ecseg_forward_patches
        self._check(fn(self.h, _ptr(p), p.shape[0], _ptr(out)), 'ecseg_forward_patches')
        if getattr(self.plan, 'output_rank', 4) == 2:
            out = out.reshape(p.shape[0], to['c'])
        return out

    def read_tensor(self, tensor, n):
        t = self.plan.tensors[tensor]
        out = np.empty((n, t['h'], t['w'], t['c']), np.float32)
        self._check(self.lib.ecseg_read_tensor(self.h, int(tensor), int(n), _ptr(out)), 'ecseg_read_tensor')
        return out

    # ---- image pipeline -----------------------------------------------------------------------------
    def segment_images(self, gray, want_raw=True, want_tie_risk=False, want_probs=False):
        """(n, H, W) uint8 pre-processed images -> (raw labels | None, post-processed labels, n_ec)
        [+ tie_risk int32 (n,) when ``want_tie_risk``: pixels whose two largest uint8-quantised probabilities differ by at most
        1; + probs float32 (n, H, W, 4) when ``want_probs``: the stitched probabilities of src/utils.py:116]."""
        g = _u8(gray)
        if g.ndim == 2:
            g = g[None]
        n, H, W = g.shape
        raw = np.empty((n, H, W), np.uint8) if want_raw else None
        post = np.empty((n, H, W), np.uint8)
        nec = np.zeros(n, np.int32)
        if not (want_tie_risk or want_probs):
            self._check(self.lib.ecseg_segment_images(self.h, _ptr(g), n, H, W, _ptr(raw), _ptr(post), _ptr(nec)),
                        'ecseg_segment_images')
            return raw, post, nec
        tie = np.zeros(n, np.int32) if want_tie_risk else None
        probs = np.empty((n, H, W, 4), np.float32) if want_probs else None
        self._check(self.lib.ecseg_segment_images_ex(self.h, _ptr(g), n, H, W, _ptr(raw), _ptr(post), _ptr(nec), _ptr(tie), _ptr(probs)),
                    'ecseg_segment_images_ex')
        return (raw, post, nec) + ((tie,) if want_tie_risk else ()) + ((probs,) if want_probs else ())

    def segment_images_dev(self, gray_ptr, n, H, W, raw_ptr, post_ptr, nec_ptr):
        self._check(self.lib.ecseg_segment_images_dev(self.h, C.c_void_p(gray_ptr), n, H, W,
                                                      C.c_void_p(raw_ptr) if raw_ptr else None, C.c_void_p(post_ptr),
                                                      C.c_void_p(nec_ptr) if nec_ptr else None),
                    'ecseg_segment_images_dev')

    def preprocess(self, imgs):
        """(n, H, W[, C]) uint8 / uint16 -> ((n, H, W) uint8 gray, inverted flags): meta_preprocess."""
        a = np.ascontiguousarray(imgs)
        if a.dtype not in (np.uint8, np.uint16):
            raise TypeError('meta_preprocess takes uint8 or uint16 images, got %s' % a.dtype)
        if a.ndim == 3:
            a = a[..., None]
        n, H, W, Cc = a.shape
        gray = np.empty((n, H, W), np.uint8)
        inv = np.zeros(n, np.int32)
        self._check(self.lib.ecseg_preprocess(self.h, _ptr(a), n, H, W, Cc, a.dtype.itemsize, _ptr(gray), _ptr(inv)),
                    'ecseg_preprocess')
        return gray, inv

    def meta_segment(self, imgs, gray_out=None, post_out=None):
        """(n, H, W[, C]) uint8 / uint16 raw images -> (gray, post-processed labels, n_ec, tie_risk): meta_preprocess + the
        segment pipeline in one device call (ecseg_meta_segment).  ``gray_out`` / ``post_out``: (n, H, W) uint8 arrays to
        fill (e.g. page-locked ones from ``host_empty``)."""
        a = np.ascontiguousarray(imgs)
        if a.dtype not in (np.uint8, np.uint16):
            raise TypeError('meta_preprocess takes uint8 or uint16 images, got %s' % a.dtype)
        if a.ndim == 3:
            a = a[..., None]
        n, H, W, Cc = a.shape
        outs = []
        for o in (gray_out, post_out):
            if o is None:
                o = np.empty((n, H, W), np.uint8)
            elif o.shape != (n, H, W) or o.dtype != np.uint8 or not o.flags.c_contiguous:
                raise ValueError('output buffers must be C-contiguous uint8 arrays of shape %s' % ((n, H, W),))
            outs.append(o)
        nec = np.zeros(n, np.int32); tie = np.zeros(n, np.int32)
        t0 = time.perf_counter()
        rc = self.lib.ecseg_meta_segment(self.h, _ptr(a), n, H, W, Cc, a.dtype.itemsize, _ptr(outs[0]), _ptr(outs[1]), _ptr(nec), _ptr(tie))
        self.native_seconds += time.perf_counter() - t0    # (inside the library, GIL released: `make metaseg`'s stage report)
        self._check(rc, 'ecseg_meta_segment')
        return outs[0], outs[1], nec, tie

    def prefetch_input(self, imgs):
        """Name the raw images (a C-contiguous array in page-locked memory) of the meta_segment call AFTER the coming one: the
        coming call uploads them under its kernels (ecseg_prefetch_input).  None withdraws."""
        if imgs is None:
            self._check(self.lib.ecseg_prefetch_input(self.h, None, 0), 'ecseg_prefetch_input')
            return
        if not imgs.flags.c_contiguous:
            raise ValueError('prefetch_input takes a C-contiguous array')
        self._check(self.lib.ecseg_prefetch_input(self.h, _ptr(imgs), imgs.nbytes), 'ecseg_prefetch_input')

    def host_empty(self, shape, dtype=np.uint8):
        """An uninitialised numpy array in page-locked host memory (ecseg_host_alloc).  The memory belongs to the handle: it is
        released by ``host_release(array)`` or when the handle closes, and must not be used after that."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        p = C.c_void_p()
        rc = self.lib.ecseg_host_alloc(self.h, max(nbytes, 1), C.byref(p))     # (thread-safe: does not touch the handle's error text)
        if rc != 0:
            e = EcsegError('ecseg_host_alloc(%d bytes) failed (%d)' % (nbytes, rc))
            e.code = rc
            raise e
        arr = np.ctypeslib.as_array((C.c_uint8 * max(nbytes, 1)).from_address(p.value))[:nbytes].view(dtype).reshape(shape)
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_release(self, arr):
        p = self._pinned.pop(arr.ctypes.data, None)
        if p is not None and self.h:
            rc = self.lib.ecseg_host_free(self.h, C.c_void_p(p))
            if rc != 0:
                e = EcsegError('ecseg_host_free failed (%d)' % rc)
                e.code = rc
                raise e

    def u16_to_u8(self, a):
        a = np.ascontiguousarray(a, np.uint16)
        out = np.empty(a.shape, np.uint8)
        self._check(self.lib.ecseg_u16_to_u8(self.h, _ptr(a), a.size, _ptr(out)), 'ecseg_u16_to_u8')
        return out

    def stitch_argmax(self, probs, n_img, H, W):
        p = np.ascontiguousarray(probs, np.float32)
        out = np.empty((n_img, H, W), np.uint8)
        self._check(self.lib.ecseg_stitch_argmax(self.h, _ptr(p), n_img, H, W, _ptr(out)), 'ecseg_stitch_argmax')
        return out

    def meta_inference(self, labels):
        a = _u8(labels)
        single = a.ndim == 2
        if single:
            a = a[None]
        n, H, W = a.shape
        out = np.empty_like(a)
        nec = np.zeros(n, np.int32)
        self._check(self.lib.ecseg_meta_inference(self.h, _ptr(a), n, H, W, _ptr(out), _ptr(nec)), 'ecseg_meta_inference')
        return (out[0], int(nec[0])) if single else (out, nec)

    def meta_inference_dev(self, in_ptr, n, H, W, out_ptr, nec_ptr):
        self._check(self.lib.ecseg_meta_inference_dev(self.h, C.c_void_p(in_ptr), n, H, W, C.c_void_p(out_ptr),
                                                      C.c_void_p(nec_ptr) if nec_ptr else None), 'ecseg_meta_inference_dev')

    # ---- counting -----------------------------------------------------------------------------------
    @staticmethod
    def _stack(a):
        a = _u8(a)
        return (a[None], True) if a.ndim == 2 else (a, False)

    def count_cc(self, mask):
        a, single = self._stack(mask)
        n, H, W = a.shape
        cnt = np.zeros(n, np.int32); px = np.zeros(n, np.int64)
        self._check(self.lib.ecseg_count_cc(self.h, _ptr(a), n, H, W, _ptr(cnt), _ptr(px)), 'ecseg_count_cc')
        return (int(cnt[0]), int(px[0])) if single else (cnt, px)

    def ccl_labels(self, mask, connectivity=8):
        a, single = self._stack(mask)
        n, H, W = a.shape
        out = np.empty((n, H, W), np.int32)
        self._check(self.lib.ecseg_ccl_labels(self.h, _ptr(a), n, H, W, int(connectivity), _ptr(out)), 'ecseg_ccl_labels')
        return out[0] if single else out

    def count_colocalization(self, ob1, ob2):
        a, single = self._stack(ob1)
        b, _ = self._stack(ob2)
        n, H, W = a.shape
        cnt = np.zeros(n, np.int32)
        self._check(self.lib.ecseg_count_colocalization(self.h, _ptr(a), _ptr(b), n, H, W, _ptr(cnt)),
                    'ecseg_count_colocalization')
        return int(cnt[0]) if single else cnt

    def count_hsr(self, chrom, fish, size_threshold=20):
        a, single = self._stack(chrom)
        b, _ = self._stack(fish)
        n, H, W = a.shape
        cnt = np.zeros(n, np.int32)
        self._check(self.lib.ecseg_count_hsr(self.h, _ptr(a), _ptr(b), n, H, W, int(size_threshold), _ptr(cnt)),
                    'ecseg_count_hsr')
        return int(cnt[0]) if single else cnt

    def overlay(self, labels, rgb, sensitivity, hsr_size_threshold=20):
        a, single = self._stack(labels)
        r = _u8(rgb)
        if r.ndim == 3:
            r = r[None]
        n, H, W = a.shape
        if r.shape[:3] != (n, H, W) or r.shape[3] < 2:
            raise ValueError('rgb must be (n, H, W, C>=2) matching labels')
        out = np.zeros((n, 12), np.int64)
        self._check(self.lib.ecseg_overlay(self.h, _ptr(a), _ptr(r), n, H, W, r.shape[3], int(sensitivity),
                                           int(hsr_size_threshold), _ptr(out)), 'ecseg_overlay')
        return out[0] if single else out

    # ---- timing ---------------------------------------------------------------------------------------
    def timings(self):
        t = np.zeros(5, np.float32)
        self._check(self.lib.ecseg_get_timings(self.h, _ptr(t)), 'ecseg_get_timings')
        return dict(zip(self.T_NAMES, [float(v) for v in t]))

    def set_kernel_profiling(self, on):
        self._check(self.lib.ecseg_set_kernel_profiling(self.h, int(bool(on))), 'ecseg_set_kernel_profiling')

    def conv_profile(self):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        self._check(self.lib.ecseg_get_conv_profile(self.h, C.byref(ms), C.byref(n), C.byref(fl)), 'ecseg_get_conv_profile')
        return ms.value, n.value, fl.value

    def conv_launch_profile(self, max_records=4096):
        """Per-launch records of the last profiled call: list of dicts (op, kind, ms, flops, executed_flops)."""
        op = np.zeros(max_records, np.int32); kind = np.zeros(max_records, np.int32)
        ms = np.zeros(max_records, np.float32); fl = np.zeros(max_records, np.float64); ex = np.zeros(max_records, np.float64)
        n = self.lib.ecseg_get_conv_launch_profile(self.h, max_records, _ptr(op), _ptr(kind), _ptr(ms), _ptr(fl), _ptr(ex))
        if n < 0:
            self._check(n, 'ecseg_get_conv_launch_profile')
        return [dict(op=int(op[k]), kind=int(kind[k]), ms=float(ms[k]), flops=float(fl[k]), executed_flops=float(ex[k]))
                for k in range(n)]

    def debug_peek(self, n=32):
        out = np.zeros(n, np.float32)
        self._check(self.lib.ecseg_debug_peek(self.h, _ptr(out), n), 'ecseg_debug_peek')
        return out

    def conv_executed_flops(self):
        fl = C.c_double()
        self._check(self.lib.ecseg_get_conv_executed_flops(self.h, C.byref(fl)), 'ecseg_get_conv_executed_flops')
        return fl.value


class Comm:
    """RCCL communicator of the C ABI (csrc/comm.hip): the record all-gather without torch.distributed.  Rank 0 creates the
    id with ``Comm.unique_id()`` and hands the 128 bytes to the other ranks; then every rank constructs its ``Comm``."""

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.ecseg_comm_unique_id(buf, 128)
        if rc != 0:
            e = EcsegError('ecseg_comm_unique_id failed (%d): %s' % (rc, lib.ecseg_comm_last_error().decode()))
            e.code = rc
            raise e
        return buf.raw

    def __init__(self, unique_id, rank, world, device):
        self.lib = load_library()
        c = C.c_void_p()
        rc = self.lib.ecseg_comm_create(C.byref(c), C.c_char_p(bytes(unique_id)), int(rank), int(world), int(device))
        if rc != 0:
            e = EcsegError('ecseg_comm_create(rank %d of %d, device %d) failed (%d): %s'
                           % (rank, world, device, rc, self.lib.ecseg_comm_last_error().decode()))
            e.code = rc
            raise e
        self.c, self.rank, self.world = c, int(rank), int(world)

    def _check(self, rc, what):
        if rc != 0:
            e = EcsegError('%s failed (%d): %s' % (what, rc, self.lib.ecseg_comm_last_error().decode()))
            e.code = rc
            raise e

    def allgather_records(self, rec):
        """rec: int64 (padded_len, 16) host array -> (world * padded_len, 16), rank-major."""
        a = np.ascontiguousarray(rec, np.int64)
        out = np.empty((self.world * a.shape[0], a.shape[1]), np.int64)
        self._check(self.lib.ecseg_allgather_records(self.c, _ptr(a), a.shape[0], _ptr(out)), 'ecseg_allgather_records')
        return out

    def allgather_records_dev(self, send_ptr, n_records, recv_ptr, stream=None):
        self._check(self.lib.ecseg_allgather_records_dev(self.c, C.c_void_p(send_ptr), int(n_records), C.c_void_p(recv_ptr),
                                                         C.c_void_p(stream) if stream else None), 'ecseg_allgather_records_dev')

    def close(self):
        if getattr(self, 'c', None):
            self.lib.ecseg_comm_destroy(self.c)
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

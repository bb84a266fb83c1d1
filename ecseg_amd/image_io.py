"""File I/O around the hot path without tifffile / OpenCV / matplotlib:

* ``imread``   - baseline TIFF reader standing in for ``skimage.io.imread`` -> tifffile (reference src/utils.py:110):
                 8/16-bit, gray / RGB(A), strips or tiles, uncompressed / LZW / Deflate / PackBits, horizontal
                 predictor, either byte order; ``.npy`` arrays are loaded with numpy.
* ``write_tiff_gray8`` - what ``cv2.imwrite('x.tif', uint8_gray)`` produces for ``dapi/<name>`` (src/utils.py:122-123):
                 LZW + horizontal predictor, RowsPerStrip = 8192 // width (the tags of example_ecSeg/dapi.jpeg).
* ``write_png`` - 8-bit gray / RGBA PNG (``labels/<stem>.png`` colour map, ``red/``, ``green/`` channel images).
LZW runs in the native library (host code, csrc/host_codec.cpp); everything else is numpy / zlib.
"""
import ctypes as C
import os
import struct
import zlib

import numpy as np

from ._lib import E_IO, E_UNSUPPORTED, load_library

# class k -> RGBA of ListedColormap(['#386cb0', '#ffff99', '#7fc97f', '#f0027f']) with vmin=0, vmax=4
# (reference src/metaseg.py:47,52)
LABEL_COLORS = np.array([[0x38, 0x6c, 0xb0, 255], [0xff, 0xff, 0x99, 255], [0x7f, 0xc9, 0x7f, 255],
                         [0xf0, 0x02, 0x7f, 255]], np.uint8)


class TiffError(ValueError):
    pass


def _lzw_decode(data, expected):
    lib = load_library()
    src = np.frombuffer(data, np.uint8)
    dst = np.empty(expected, np.uint8)
    n = lib.ecseg_lzw_decode(src.ctypes.data_as(C.c_void_p), len(src), dst.ctypes.data_as(C.c_void_p), expected)
    if n < 0:
        raise TiffError('corrupt LZW stream')
    if n < expected:
        dst[n:] = 0
    return dst


def _lzw_encode(raw):
    lib = load_library()
    src = np.ascontiguousarray(np.frombuffer(raw, np.uint8))
    cap = 2 * len(src) + 64
    dst = np.empty(cap, np.uint8)
    n = lib.ecseg_lzw_encode(src.ctypes.data_as(C.c_void_p), len(src), dst.ctypes.data_as(C.c_void_p), cap)
    if n < 0:
        raise TiffError('LZW encode failed')
    return dst[:n].tobytes()


def _packbits_decode(data, expected):
    out = bytearray()
    i, n = 0, len(data)
    while i < n and len(out) < expected:
        h = data[i]
        i += 1
        if h < 128:
            out += data[i:i + h + 1]
            i += h + 1
        elif h > 128:
            out += data[i:i + 1] * (257 - h)
            i += 1
    out = bytes(out[:expected])
    return np.frombuffer(out + b'\0' * (expected - len(out)), np.uint8)


_TYPE_FMT = {1: 'B', 2: 'c', 3: 'H', 4: 'I', 5: 'II', 6: 'b', 7: 'B', 8: 'h', 9: 'i', 10: 'ii', 11: 'f', 12: 'd',
             16: 'Q', 17: 'q', 18: 'Q'}


def _read_ifd(buf, off, bo, big):
    tags = {}
    if big:
        n = struct.unpack_from(bo + 'Q', buf, off)[0]
        p, esz, cnt_fmt, inl = off + 8, 20, 'Q', 8
    else:
        n = struct.unpack_from(bo + 'H', buf, off)[0]
        p, esz, cnt_fmt, inl = off + 2, 12, 'I', 4
    for i in range(n):
        e = p + i * esz
        tag, typ = struct.unpack_from(bo + 'HH', buf, e)
        cnt = struct.unpack_from(bo + cnt_fmt, buf, e + 4)[0]
        fmt = _TYPE_FMT.get(typ)
        if fmt is None:
            continue
        size = struct.calcsize('=' + fmt) * cnt
        voff = e + 4 + struct.calcsize(cnt_fmt)
        if size > inl:
            voff = struct.unpack_from(bo + cnt_fmt, buf, voff)[0]
        if typ == 2:
            tags[tag] = bytes(buf[voff:voff + cnt])
        else:
            vals = struct.unpack_from(bo + fmt * cnt, buf, voff)
            tags[tag] = vals
    return tags


def _native_tiff(path):
    """Whole-file native reader (csrc/host_io.cpp, runs without the GIL): baseline strips, 8 / 16-bit unsigned, none / LZW /
    Deflate, predictor 1 / 2.  -> array, or None when the layout is left to the Python reader below."""
    lib = load_library()
    H, W, spp, bits = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    enc = os.fsencode(path)
    rc = lib.ecseg_tiff_info(enc, C.byref(H), C.byref(W), C.byref(spp), C.byref(bits))
    if rc == E_UNSUPPORTED:
        return None
    if rc == E_IO:
        raise OSError('cannot read %s' % path)
    if rc != 0:
        raise TiffError('%s is not a readable TIFF file' % path)
    out = np.empty((H.value, W.value, spp.value), np.uint8 if bits.value == 8 else np.uint16)
    rc = lib.ecseg_tiff_read(enc, out.ctypes.data_as(C.c_void_p), out.nbytes)
    if rc == E_UNSUPPORTED:
        return None
    if rc != 0:
        raise TiffError('corrupt TIFF file %s' % path)
    return out[..., 0] if spp.value == 1 else out


def read_tiff(path, native=True):
    """First image of a TIFF file -> numpy array (H, W) or (H, W, S), dtype uint8 / uint16 (or what the file holds)."""
    if native:
        a = _native_tiff(path)
        if a is not None:
            return a
    with open(path, 'rb') as f:
        buf = f.read()
    if buf[:2] == b'II':
        bo = '<'
    elif buf[:2] == b'MM':
        bo = '>'
    else:
        raise TiffError('%s is not a TIFF file' % path)
    magic = struct.unpack_from(bo + 'H', buf, 2)[0]
    if magic == 42:
        big = False
        ifd = struct.unpack_from(bo + 'I', buf, 4)[0]
    elif magic == 43:
        big = True
        ifd = struct.unpack_from(bo + 'Q', buf, 8)[0]
    else:
        raise TiffError('bad TIFF magic %d' % magic)
    t = _read_ifd(buf, ifd, bo, big)
    W, H = t[256][0], t[257][0]
    bps = t.get(258, (1,))
    spp = t.get(277, (1,))[0]
    comp = t.get(259, (1,))[0]
    planar = t.get(284, (1,))[0]
    pred = t.get(317, (1,))[0]
    fmt = t.get(339, (1,))[0]
    if len(set(bps)) != 1:
        raise TiffError('mixed bits per sample are not supported')
    bits = bps[0]
    if bits not in (8, 16, 32) or fmt not in (1, 2, 3):
        raise TiffError('unsupported sample format (%d bits, format %d)' % (bits, fmt))
    kind = {1: 'u', 2: 'i', 3: 'f'}[fmt]
    dt = np.dtype(bo + kind + str(bits // 8))
    bpp = bits // 8
    if planar != 1 and spp > 1:
        raise TiffError('planar TIFF is not supported')

    def decode(chunk, nbytes):
        if comp == 1:
            a = np.frombuffer(chunk, np.uint8, count=min(len(chunk), nbytes))
            if len(a) < nbytes:
                a = np.concatenate([a, np.zeros(nbytes - len(a), np.uint8)])
            return a
        if comp == 5:
            return _lzw_decode(chunk, nbytes)
        if comp in (8, 32946):
            a = np.frombuffer(zlib.decompress(chunk), np.uint8)
            return a[:nbytes] if len(a) >= nbytes else np.concatenate([a, np.zeros(nbytes - len(a), np.uint8)])
        if comp == 32773:
            return _packbits_decode(chunk, nbytes)
        raise TiffError('TIFF compression %d is not supported' % comp)

    def unpredict(block):           # block: (rows, cols, spp) of dt
        if pred == 2:
            if kind == 'f':
                raise TiffError('predictor 2 on float samples')
            return np.cumsum(block, axis=1, dtype=block.dtype.newbyteorder('=')).astype(block.dtype.newbyteorder('='))
        if pred != 1:
            raise TiffError('TIFF predictor %d is not supported' % pred)
        return block

    out = np.zeros((H, W, spp), dt.newbyteorder('='))
    if 322 in t:       # tiles
        tw, th = t[322][0], t[323][0]
        offs, cnts = t[324], t[325]
        k = 0
        for ty in range(0, H, th):
            for tx in range(0, W, tw):
                raw = decode(buf[offs[k]:offs[k] + cnts[k]], tw * th * spp * bpp)
                blk = unpredict(raw.view(dt).reshape(th, tw, spp))
                out[ty:ty + th, tx:tx + tw] = blk[:min(th, H - ty), :min(tw, W - tx)]
                k += 1
    else:
        rps = min(t.get(278, (H,))[0] or H, H)      # RowsPerStrip 0: one strip, as libtiff and the native reader read it
        offs, cnts = t[273], t.get(279)
        if cnts is None:
            cnts = [len(buf) - offs[0]] if len(offs) == 1 else None
        if cnts is None:
            raise TiffError('missing StripByteCounts')
        for k, (o, c) in enumerate(zip(offs, cnts)):
            r0 = k * rps
            rows = min(rps, H - r0)
            if rows <= 0:
                break
            raw = decode(buf[o:o + c], rows * W * spp * bpp)
            out[r0:r0 + rows] = unpredict(raw.view(dt).reshape(rows, W, spp))
    return out[..., 0] if spp == 1 else out


def image_shape(path):
    """Shape of what ``imread(path)`` returns, from the file header where possible (resume check of `make metaseg`)."""
    if str(path).lower().endswith('.npy'):
        return np.load(path, mmap_mode='r').shape
    lib = load_library()
    H, W, spp, bits = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    if lib.ecseg_tiff_info(os.fsencode(path), C.byref(H), C.byref(W), C.byref(spp), C.byref(bits)) == 0:
        return (H.value, W.value) if spp.value == 1 else (H.value, W.value, spp.value)
    return read_tiff(path).shape


def imread(path):
    """``skimage.io.imread`` stand-in for the inputs ``get_imgs`` globs (``*.tif`` and ``*.npy``, src/utils.py:105-110)."""
    if str(path).lower().endswith('.npy'):
        return np.load(path)
    return read_tiff(path)


def _ifd_entry(tag, typ, count, value):
    return struct.pack('<HHII', tag, typ, count, value)


def _check_write(rc, path):
    if rc == E_IO:
        raise OSError('cannot write %s' % path)
    if rc != 0:
        raise ValueError('could not encode %s (status %d)' % (path, rc))


def write_tiff_gray8(path, img, invert=False, native=True):
    """8-bit grayscale TIFF, LZW + horizontal predictor, strips of 8192 // width rows (OpenCV imwrite defaults).
    ``invert``: store 255 - img (``cv2.bitwise_not``, src/utils.py:112).  The native writer (csrc/host_io.cpp) and the
    Python one below produce the same bytes (tests/test_io_fixtures.py)."""
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim != 2:
        raise ValueError('write_tiff_gray8 takes a 2-D uint8 image')
    if native and img.size:
        _check_write(load_library().ecseg_tiff_write_gray8(os.fsencode(path), img.ctypes.data_as(C.c_void_p), img.shape[0],
                                                           img.shape[1], int(bool(invert))), path)
        return
    if invert:
        img = ~img
    H, W = img.shape
    rps = max(1, min(H, 8192 // max(W, 1)))
    diff = img.copy()
    diff[:, 1:] = img[:, 1:] - img[:, :-1]            # horizontal differencing (wraps modulo 256)
    strips = [_lzw_encode(diff[r:r + rps].tobytes()) for r in range(0, H, rps)]
    n = len(strips)
    data_off = 8
    offsets, pos = [], data_off
    for s in strips:
        offsets.append(pos)
        pos += len(s) + (len(s) & 1)
    extra = b''
    extra_off = pos

    def arr(vals):
        nonlocal extra
        o = extra_off + len(extra)
        extra += struct.pack('<%dI' % len(vals), *vals)
        return o

    so = offsets[0] if n == 1 else arr(offsets)
    sc = len(strips[0]) if n == 1 else arr([len(s) for s in strips])
    entries = [
        _ifd_entry(256, 4, 1, W), _ifd_entry(257, 4, 1, H), _ifd_entry(258, 3, 1, 8), _ifd_entry(259, 3, 1, 5),
        _ifd_entry(262, 3, 1, 1), _ifd_entry(273, 4, n, so), _ifd_entry(277, 3, 1, 1), _ifd_entry(278, 4, 1, rps),
        _ifd_entry(279, 4, n, sc), _ifd_entry(284, 3, 1, 1), _ifd_entry(317, 3, 1, 2), _ifd_entry(339, 3, 1, 1),
    ]
    ifd_off = extra_off + len(extra)
    ifd_off += ifd_off & 1
    with open(path, 'wb') as f:
        f.write(b'II' + struct.pack('<HI', 42, ifd_off))
        for s in strips:
            f.write(s)
            if len(s) & 1:
                f.write(b'\0')
        f.write(extra)
        if (extra_off + len(extra)) & 1:
            f.write(b'\0')
        f.write(struct.pack('<H', len(entries)) + b''.join(entries) + struct.pack('<I', 0))


def write_png(path, img, level=6, native=True):
    """8-bit PNG: (H, W) gray, (H, W, 3) RGB or (H, W, 4) RGBA."""
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim == 2:
        ctype, ch = 0, 1
    elif img.shape[2] == 3:
        ctype, ch = 2, 3
    elif img.shape[2] == 4:
        ctype, ch = 6, 4
    else:
        raise ValueError('unsupported PNG shape %s' % (img.shape,))
    H, W = img.shape[:2]
    if native and img.size:
        _check_write(load_library().ecseg_png_write(os.fsencode(path), img.ctypes.data_as(C.c_void_p), H, W, ch, int(level)), path)
        return
    rows = np.empty((H, 1 + W * ch), np.uint8)           # filter type 0 on every scan line
    rows[:, 0] = 0
    rows[:, 1:] = img.reshape(H, W * ch)

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(data, zlib.crc32(tag)) & 0xffffffff)

    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n')
        f.write(chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 8, ctype, 0, 0, 0)))
        f.write(chunk(b'IDAT', zlib.compress(rows, level)))          # (zlib takes the buffer directly: no tobytes copy)
        f.write(chunk(b'IEND', b''))


_LABEL_COLORS_U32 = None


def write_label_png(path, labels, native=True):
    """``plt.imsave(path, I.astype('uint8'), cmap=ListedColormap([...4 colours...]), vmin=0, vmax=4)``
    (src/metaseg.py:47-52): class k -> colour k, RGBA.  The four colours compress to almost nothing at any zlib level;
    level 1 keeps the encoder off the critical path of `make metaseg` (pixels, not bytes, are the contract)."""
    global _LABEL_COLORS_U32
    if _LABEL_COLORS_U32 is None:
        _LABEL_COLORS_U32 = np.ascontiguousarray(LABEL_COLORS).view('<u4').ravel().copy()
    lab = np.asarray(labels)
    if lab.dtype != np.uint8:
        lab = np.clip(lab, 0, 3).astype(np.uint8)
    if native and lab.ndim == 2 and lab.size:
        lab = np.ascontiguousarray(lab)
        _check_write(load_library().ecseg_png_write_labels(os.fsencode(path), lab.ctypes.data_as(C.c_void_p), lab.shape[0],
                                                           lab.shape[1]), path)
        return
    rgba = np.take(_LABEL_COLORS_U32, lab, mode='clip')              # one 32-bit gather per pixel
    write_png(path, rgba.view(np.uint8).reshape(lab.shape[0], lab.shape[1], 4), level=1, native=False)


def write_npy_int64(path, labels):
    """``np.save(path, labels.astype('int64'))`` (src/metaseg.py:53) for a uint8 label image: the widening and the write
    happen in csrc/host_io.cpp, byte-identical to numpy's file (tests/test_io_fixtures.py).  ``path`` is taken as given (no
    ``.npy`` is appended)."""
    lab = np.asarray(labels)
    if lab.dtype == np.uint8 and lab.ndim == 2:
        lab = np.ascontiguousarray(lab)
        _check_write(load_library().ecseg_npy_write_i64(os.fsencode(path), lab.ctypes.data_as(C.c_void_p), lab.shape[0],
                                                        lab.shape[1]), path)
        return
    with open(path, 'wb') as f:
        np.save(f, lab.astype(np.int64))


def write_png_channel(path, img, channel, invert=False):
    """``cv2.imwrite(path, cv2.bitwise_not(np.uint8(I[..., channel])))`` (src/image_tools.py:143-144) for an interleaved uint8
    (H, W, C) image: the channel gather, the inversion and cv2's default PNG coder (SUB filter + run-length deflate) run in
    csrc/host_io.cpp without the interpreter lock."""
    a = np.asarray(img)
    if a.dtype != np.uint8 or a.ndim != 3 or not a.flags.c_contiguous or not a.size:
        g = np.ascontiguousarray(np.uint8(a[..., channel]))
        write_png(path, ~g if invert else g, level=-1)
        return
    _check_write(load_library().ecseg_png_write_channel(os.fsencode(path), a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1],
                                                        a.shape[2], int(channel), int(bool(invert))), path)


def read_npy_labels_u8(path):
    """``np.load(path).astype(np.uint8)`` for the labels/<stem>.npy files `make metaseg` writes (int64, (H, W); read_seg,
    src/utils.py:125-132), narrowed while the file is read (csrc/host_io.cpp); any other .npy layout goes through numpy."""
    lib = load_library()
    H, W = C.c_int(), C.c_int()
    rc = lib.ecseg_npy_label_info(os.fsencode(path), C.byref(H), C.byref(W))
    if rc == 0:
        out = np.empty((H.value, W.value), np.uint8)
        rc = lib.ecseg_npy_read_labels_u8(os.fsencode(path), out.ctypes.data_as(C.c_void_p), H.value, W.value)
        if rc == 0:
            return out
    if rc == E_IO:
        raise OSError('cannot read %s' % path)
    return np.ascontiguousarray(np.load(path).astype(np.uint8))

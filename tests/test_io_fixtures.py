"""The file I/O layer around the hot path (A2 imread, A5 dapi/<name>.tif, A18 labels/<stem>.png) pinned on
fixtures produced by the libraries the reference itself uses (tools/make_golden.py gen_io, run under the container's
conda Python): TIFF files written by tifffile / libtiff with the pixels ``skimage.io.imread`` returned for them, and the
RGBA that ``plt.imsave(cmap=4 colours, vmin=0, vmax=4)`` writes.  Files the product writes are read back with
independent decoders (PIL/libtiff always; tifffile when the conda Python is present).  No GPU needed: the LZW codec is
host code in libecseg_hip.so."""
import base64
import json
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

from ecseg_amd import image_io

CONDA = '/opt/conda/bin/python3.9'
EXAMPLE = '/root/reference/example_ecSeg/dapi.jpeg'


def _fixtures(golden_dir):
    files = json.load(open(os.path.join(golden_dir, 'io_tiff_files.json')))['files']
    px = np.load(os.path.join(golden_dir, 'io_tiff_pixels.npz'))
    return files, px


def test_imread_matches_skimage_on_tifffile_written_files(golden_dir, tmp_path):
    files, px = _fixtures(golden_dir)
    assert len(files) >= 12
    kinds = set()
    for name, b64 in files.items():
        path = str(tmp_path / (name + '.tif'))
        open(path, 'wb').write(base64.b64decode(b64))
        got = image_io.imread(path)
        want = px[name]
        assert got.shape == want.shape and got.dtype == want.dtype, name
        assert np.array_equal(got, want), name
        kinds.add((want.dtype.itemsize, want.ndim))
    assert kinds == {(1, 2), (1, 3), (2, 2), (2, 3)}            # 8/16-bit, gray/RGB all covered


def test_label_png_matches_matplotlib_imsave(golden_dir, tmp_path):
    z = np.load(os.path.join(golden_dir, 'io_label_png.npz'))
    assert np.array_equal(image_io.LABEL_COLORS, z['class_rgba'])
    path = str(tmp_path / 'lab.png')
    image_io.write_label_png(path, z['labels'].astype(np.int64))
    img = Image.open(path)
    assert img.mode == 'RGBA'
    assert np.array_equal(np.array(img), z['rgba'])


@pytest.mark.parametrize('hw', [(1040, 1392), (37, 52), (5, 9000), (64, 1)])
def test_written_dapi_tiff_decodes_with_libtiff_and_has_opencv_tags(tmp_path, hw):
    rng = np.random.default_rng(hw[0])
    H, W = hw
    yy, xx = np.mgrid[:H, :W]
    img = ((yy * 3 + xx * 7) % 200 + rng.integers(0, 40, (H, W))).astype(np.uint8)
    path = str(tmp_path / 'x.tif')
    image_io.write_tiff_gray8(path, img)
    im = Image.open(path)
    assert np.array_equal(np.array(im), img)
    tags = im.tag_v2
    assert tags[259] == 5 and tags[317] == 2 and tags[258] == (8,) and tags[262] == 1       # LZW, predictor 2, 8-bit gray
    assert tags[278] == max(1, min(H, 8192 // W))                 # rows per strip of cv2.imwrite (5 at width 1392)
    assert np.array_equal(image_io.imread(path), img)


@pytest.mark.skipif(not os.path.exists(CONDA), reason='conda Python with tifffile not present (GPU box)')
def test_written_files_decode_with_tifffile(tmp_path):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (300, 462)).astype(np.uint8)
    img[:, :200] = 7                                              # long runs: exercises LZW table growth / resets
    tif, png = str(tmp_path / 'a.tif'), str(tmp_path / 'b.png')
    image_io.write_tiff_gray8(tif, img)
    lab = rng.integers(0, 4, (50, 60))
    image_io.write_label_png(png, lab)
    np.save(str(tmp_path / 'img.npy'), img)
    np.save(str(tmp_path / 'rgba.npy'), image_io.LABEL_COLORS[lab])
    code = ("import sys, numpy as np\n"
            "from skimage.io import imread\n"
            "import matplotlib.pyplot as plt\n"
            "d = sys.argv[1]\n"
            "assert np.array_equal(imread(d + '/a.tif'), np.load(d + '/img.npy'))\n"
            "p = (plt.imread(d + '/b.png') * 255 + 0.5).astype(np.uint8)\n"
            "assert np.array_equal(p, np.load(d + '/rgba.npy'))\n"
            "print('ok')\n")
    out = subprocess.run([CONDA, '-W', 'ignore', '-c', code, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-1500:]


@pytest.mark.skipif(not os.path.exists(EXAMPLE), reason='reference tree not present (GPU box)')
def test_reference_example_file_decodes_to_the_stored_pixels(golden_dir):
    """example_ecSeg/dapi.jpeg is an OpenCV-written TIFF (LZW, predictor 2, 5 rows per strip): the real thing the reader
    must handle; its pixels as decoded by skimage.io.imread are in dapi_example.npz."""
    want = np.load(os.path.join(golden_dir, 'dapi_example.npz'))['dapi']
    assert np.array_equal(image_io.imread(EXAMPLE), want)


def test_lzw_roundtrip_and_corrupt_streams_do_not_crash():
    rng = np.random.default_rng(0)
    for n in (0, 1, 2, 255, 4096, 70000):
        for kind in range(3):
            raw = (np.zeros(n, np.uint8) if kind == 0 else rng.integers(0, 256, n).astype(np.uint8) if kind == 1
                   else np.repeat(rng.integers(0, 4, n // 16 + 1).astype(np.uint8), 16)[:n])
            enc = image_io._lzw_encode(raw.tobytes())
            assert np.array_equal(image_io._lzw_decode(enc, n), raw)
    raw = np.repeat(np.arange(64, dtype=np.uint8), 200)
    enc = bytearray(image_io._lzw_encode(raw.tobytes()))
    for trial in range(200):                                      # truncated / bit-flipped streams: error or garbage, never a crash
        bad = bytearray(enc[:rng.integers(1, len(enc) + 1)])
        for _ in range(int(rng.integers(0, 4))):
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        try:
            out = image_io._lzw_decode(bytes(bad), len(raw))
            assert out.shape == (len(raw),)
        except image_io.TiffError:
            pass
    with pytest.raises(image_io.TiffError):
        image_io._lzw_decode(b'\xff\xff\xff\xff\xff\xff', 100)


def _pack_codes(codes, width=9):
    """MSB-first bit string of fixed-width LZW codes (fewer than 250 codes: the width never changes)."""
    bits = ''.join(format(c, '0%db' % width) for c in codes)
    bits += '0' * (-len(bits) % 8)
    return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))


def test_lzw_streams_without_a_leading_clear_code_and_with_two_in_a_row():
    """the fast form consumed a code and then left for the careful form, which resumed from the NEXT code - a
    strip that does not begin with ClearCode (libtiff accepts it) or holds two ClearCodes in a row lost one byte and came
    out shifted.  Literal-only streams (every code < 256 adds an entry nobody uses) decode to the literals themselves."""
    rng = np.random.default_rng(5)
    lits = [int(v) for v in rng.integers(0, 256, 60)]
    want = np.array(lits, np.uint8)
    for codes in (lits + [257],                          # no ClearCode at all
                  [256, 256] + lits + [257],             # two in a row
                  [256] + lits[:30] + [256, 256] + lits[30:] + [257],
                  [256] + lits + [257]):                 # the ordinary shape
        for pad in (0, 16):                              # with / without room for the fast form to run to the end
            got = image_io._lzw_decode(_pack_codes(codes) + bytes(pad), len(lits))
            assert np.array_equal(got, want), (codes[:3], pad)
    with pytest.raises(image_io.TiffError):              # a table code with no previous string is corrupt
        image_io._lzw_decode(_pack_codes([300] + lits + [257]) + bytes(16), 61)


def test_host_codec_under_address_sanitizer():
    """`make asan`: ASan + UBSan build of csrc/host_codec.cpp driven over a corrupt-stream corpus (tools/asan)."""
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which('g++') or not shutil.which('make'):
        pytest.skip('no g++ / make')
    out = subprocess.run(['make', '-C', root, 'asan'], capture_output=True, text=True, timeout=600)
    if 'cannot find' in out.stderr and 'asan' in out.stderr:
        pytest.skip('libasan not installed')
    assert out.returncode == 0, (out.stdout + out.stderr)[-2000:]
    assert 'codec_fuzz ok' in out.stdout


def test_truncated_and_corrupt_h5_files_raise_cleanly(golden_dir, tmp_path):
    """hdf5_min parses the weights file (untrusted bytes): damaged files must raise, not hang or return garbage silently."""
    from ecseg_amd import hdf5_min
    data = open(os.path.join(golden_dir, 'keras_tiny.h5'), 'rb').read()
    rng = np.random.default_rng(1)
    for cut in (0, 7, 8, 100, len(data) // 3, len(data) - 1000):
        p = str(tmp_path / ('cut%d.h5' % cut))
        open(p, 'wb').write(data[:cut])
        with pytest.raises(Exception):
            hdf5_min.load_keras_h5(p)
    for trial in range(20):                                # byte noise in the metadata region: raise or load, never hang
        bad = bytearray(data)
        for _ in range(8):
            bad[int(rng.integers(0, min(len(bad), 4096)))] = int(rng.integers(0, 256))
        p = str(tmp_path / ('noise%d.h5' % trial))
        open(p, 'wb').write(bytes(bad))
        try:
            hdf5_min.load_keras_h5(p)
        except Exception:
            pass


def test_native_and_python_tiff_readers_agree_on_every_fixture(golden_dir, tmp_path):
    """Round 3: imread goes through the whole-file native reader (csrc/host_io.cpp) where the layout allows; the pure-Python
    reader stays for the rest.  Both must return the stored pixels, and the native one must leave tiled / PackBits / ... files
    to the Python one instead of failing."""
    files, px = _fixtures(golden_dir)
    native = 0
    for name, b64 in files.items():
        path = str(tmp_path / (name + '.tif'))
        open(path, 'wb').write(base64.b64decode(b64))
        a = image_io._native_tiff(path)
        b = image_io.read_tiff(path, native=False)
        assert np.array_equal(b, px[name]) and b.dtype == px[name].dtype, name
        if a is not None:
            native += 1
            assert a.dtype == px[name].dtype and np.array_equal(a, px[name]), name
        assert image_io.image_shape(path) == px[name].shape, name
    assert native >= 8, native                                   # strips + none / LZW / Deflate + predictor: the common files
    bad = str(tmp_path / 'bad.tif')
    open(bad, 'wb').write(b'II*\x00' + bytes(range(200)))
    with pytest.raises(ValueError):
        image_io.imread(bad)
    with pytest.raises(OSError):
        image_io.imread(str(tmp_path / 'missing.tif'))


@pytest.mark.parametrize('hw', [(1040, 1392), (37, 52), (5, 9000), (64, 1), (1, 1)])
def test_native_writers_equal_the_python_ones_byte_for_byte(tmp_path, hw):
    rng = np.random.default_rng(hw[1])
    img = rng.integers(0, 256, hw, dtype=np.uint8)
    img[: hw[0] // 2] //= 64                                     # compressible half
    a, b = str(tmp_path / 'a.tif'), str(tmp_path / 'b.tif')
    for inv in (False, True):
        image_io.write_tiff_gray8(a, img, invert=inv)
        image_io.write_tiff_gray8(b, img, invert=inv, native=False)
        assert open(a, 'rb').read() == open(b, 'rb').read()
        assert np.array_equal(np.array(Image.open(a)), ~img if inv else img)
    lab = (img >> 6).astype(np.uint8)
    image_io.write_npy_int64(a, lab)
    import io
    ref = io.BytesIO()
    np.save(ref, lab.astype(np.int64))
    assert open(a, 'rb').read() == ref.getvalue()                # np.save(labels.astype('int64')), src/metaseg.py:53
    rgb = rng.integers(0, 256, hw + (3,), dtype=np.uint8)
    image_io.write_png(a, rgb)
    image_io.write_png(b, rgb, native=False)
    assert open(a, 'rb').read() == open(b, 'rb').read()
    assert np.array_equal(np.array(Image.open(a)), rgb)


def test_label_png_run_length_deflate_decodes_everywhere(tmp_path):
    """labels/<stem>.png is compressed by a purpose-built deflate encoder (runs of equal pixels as distance-4 matches, one
    dynamic-Huffman block, Adler-32 in closed form): libpng (PIL) and zlib must reproduce the RGBA rows exactly for speckle,
    flat images, single rows / columns, runs longer than one match (> 64 pixels) and values above 3 (clipped like vmax=4)."""
    import struct
    import zlib
    rng = np.random.default_rng(7)
    cases = [rng.integers(0, 4, (97, 211)), np.zeros((64, 80)), np.full((3, 70000), 1), rng.integers(0, 4, (50, 1)),
             rng.integers(0, 4, (1, 300)), np.full((1, 1), 2), np.tile(np.array([0, 1, 2, 3]), (37, 91)), rng.integers(0, 9, (40, 50)),
             (rng.random((300, 500)) < 0.01) * 3, rng.integers(0, 2, (5, 63)), rng.integers(0, 2, (5, 64)), rng.integers(0, 2, (5, 65)),
             np.repeat(rng.integers(0, 4, (8, 40)), 67, axis=1)]
    path = str(tmp_path / 'l.png')
    for k, lab in enumerate(cases):
        lab = lab.astype(np.uint8)
        image_io.write_label_png(path, lab)
        want = image_io.LABEL_COLORS[np.minimum(lab, 3)]
        assert np.array_equal(np.array(Image.open(path)), want), k
        raw = open(path, 'rb').read()
        pos, idat = 8, b''
        while pos < len(raw):                                    # chunk CRCs + the zlib stream itself (Adler-32 is checked by zlib)
            n, tag = struct.unpack('>I4s', raw[pos:pos + 8])
            body = raw[pos + 8:pos + 8 + n]
            assert struct.unpack('>I', raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xffffffff
            if tag == b'IDAT':
                idat += body
            pos += 12 + n
        rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(lab.shape[0], 1 + 4 * lab.shape[1])
        assert not rows[:, 0].any() and np.array_equal(rows[:, 1:].reshape(want.shape), want), k


def test_channel_png_and_native_label_reader(tmp_path):
    """Round 5 host pipeline of `make meta_overlay`: red/ green/ channel PNGs written by the library's own SUB-filter + run-length
    deflate (cv2.imwrite's default settings; src/image_tools.py:143-144) decode with libpng (PIL) and zlib to the inverted channel,
    for noise, flat images, runs around the 258-byte match limit and degenerate shapes; labels/<stem>.npy is read back narrowed to
    uint8 without numpy for every integer dtype metaseg or a user may have written (src/utils.py:125-132), other layouts fall back."""
    import struct
    import zlib
    from ecseg_amd import synth
    rng = np.random.default_rng(0)
    cases = [synth.dapi_image(5, 200, 333, rgb=True), np.zeros((3, 700, 1), np.uint8), np.full((2, 9, 4), 200, np.uint8),
             rng.integers(0, 256, (1, 1, 3), dtype=np.uint8), rng.integers(0, 2, (40, 1031, 2), dtype=np.uint8) * 255]
    runs = np.zeros((6, 1500, 1), np.uint8)
    for r, n in enumerate((257, 258, 259, 260, 261, 519)):
        runs[r, 10:10 + n + 1, 0] = 77                              # SUB-filtered: one 77 followed by n zeros
    cases.append(runs)
    for a in cases:
        for c in range(a.shape[2]):
            for inv in (False, True):
                p = str(tmp_path / 'c.png')
                image_io.write_png_channel(p, a, c, invert=inv)
                want = ~a[..., c] if inv else a[..., c]
                assert np.array_equal(np.array(Image.open(p)), want), (a.shape, c, inv)
                raw = open(p, 'rb').read()
                pos, idat = 8, b''
                while pos < len(raw):
                    n, tag = struct.unpack('>I', raw[pos:pos + 4])[0], raw[pos + 4:pos + 8]
                    body = raw[pos + 8:pos + 8 + n]
                    assert struct.unpack('>I', raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xffffffff
                    idat += body if tag == b'IDAT' else b''
                    pos += 12 + n
                rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(a.shape[0], 1 + a.shape[1])
                assert (rows[:, 0] == 1).all() and np.array_equal(np.cumsum(rows[:, 1:], axis=1, dtype=np.uint8), want)
    # cv2's default settings through the generic writer too (level -1)
    g = synth.dapi_image(6, 120, 97)
    image_io.write_png(str(tmp_path / 'g.png'), g, level=-1)
    assert np.array_equal(np.array(Image.open(str(tmp_path / 'g.png'))), g)
    lab = synth.label_map(4, 130, 211)
    p = str(tmp_path / 'l.npy')
    image_io.write_npy_int64(p, lab)
    assert np.array_equal(image_io.read_npy_labels_u8(p), lab) and image_io.read_npy_labels_u8(p).dtype == np.uint8
    for dt, layout in ((np.int32, None), (np.int16, None), (np.uint8, None), (np.bool_, None), (np.int64, 'F'), (np.float32, None), ('>i4', None)):
        arr = lab.astype(dt)
        np.save(p, np.asfortranarray(arr) if layout == 'F' else arr)
        assert np.array_equal(image_io.read_npy_labels_u8(p), arr.astype(np.uint8)), dt
    open(p, 'wb').write(open(p, 'rb').read()[:300])                   # truncated: numpy's own error, not garbage
    with pytest.raises(Exception):
        image_io.read_npy_labels_u8(p)

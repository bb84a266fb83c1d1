"""Minimal pure-Python reader for the HDF5 subset that Keras ``model.save('x.h5')`` (h5py, default
``libver='earliest'``) produces.  Replaces the h5py dependency of ``tf.keras.models.load_model``
(reference src/utils.py:27-33) on machines where neither TensorFlow nor h5py exists.

Supported: superblock v0/v1 (v2/v3 root object header as well), v1 and v2 object headers, old-style groups
(symbol table: B-tree v1 ``TREE`` + ``SNOD`` + local ``HEAP``) and compact new-style groups (link messages),
attributes (message versions 1-3) holding fixed-length strings, variable-length strings (global heap
``GCOL``), and little/big-endian integer / float scalars and arrays; datasets with contiguous or compact
layout and chunked layout without filters (B-tree v1 chunk index).  Anything else raises ``Hdf5Error``.
"""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5Error(Exception):
    pass


class _Dtype:
    def __init__(self, kind, size, np_dtype=None, base=None, pad=0):
        self.kind, self.size, self.np_dtype, self.base, self.pad = kind, size, np_dtype, base, pad


class Node:
    """A group or dataset."""

    def __init__(self, f, addr, name='/'):
        self._f, self._addr, self.name = f, addr, name
        self._msgs = f._object_messages(addr)
        self._links = None
        self._attrs = None

    # ---- attributes -------------------------------------------------------------------------------
    @property
    def attrs(self):
        if self._attrs is None:
            a = {}
            for t, body in self._msgs:
                if t == 0x000C:
                    k, v = self._f._parse_attribute(body)
                    a[k] = v
                elif t == 0x0015:  # attribute info: dense storage unsupported unless empty
                    flags = body[1]
                    off = 2 + (2 if flags & 1 else 0)
                    fheap = struct.unpack_from('<Q', body, off)[0]
                    if fheap != UNDEF:
                        raise Hdf5Error('dense attribute storage is not supported')
            self._attrs = a
        return self._attrs

    # ---- group behaviour --------------------------------------------------------------------------
    def _load_links(self):
        if self._links is not None:
            return
        links = {}
        for t, body in self._msgs:
            if t == 0x0011:      # symbol table message
                btree, heap = struct.unpack_from('<QQ', body, 0)
                links.update(self._f._read_symbol_table(btree, heap))
            elif t == 0x0006:    # link message
                name, addr = self._f._parse_link(body)
                if addr is not None:
                    links[name] = addr
            elif t == 0x0002:    # link info
                flags = body[1]
                off = 2 + (8 if flags & 1 else 0)
                fheap = struct.unpack_from('<Q', body, off)[0]
                if fheap != UNDEF:
                    raise Hdf5Error('dense link storage (fractal heap) is not supported')
        self._links = links

    def keys(self):
        self._load_links()
        return list(self._links.keys())

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split('/') if p]:
            node._load_links()
            if part not in node._links:
                raise KeyError(path)
            node = Node(node._f, node._links[part], part)
        return node

    @property
    def is_dataset(self):
        return any(t == 0x0008 for t, _ in self._msgs)

    def visit_datasets(self, prefix=''):
        """Yield (path, Node) for every dataset below this group."""
        for k in self.keys():
            child = self[k]
            p = prefix + '/' + k if prefix else k
            if child.is_dataset:
                yield p, child
            else:
                yield from child.visit_datasets(p)

    # ---- dataset behaviour ------------------------------------------------------------------------
    def read(self):
        dt = shape = layout = None
        for t, body in self._msgs:
            if t == 0x0003:
                dt = self._f._parse_datatype(body, 0)[0]
            elif t == 0x0001:
                shape = self._f._parse_dataspace(body)
            elif t == 0x0008:
                layout = body
            elif t == 0x000B:
                raise Hdf5Error('filtered (compressed) datasets are not supported')
        if dt is None or shape is None or layout is None:
            raise Hdf5Error('%s is not a dataset' % self.name)
        if dt.np_dtype is None:
            raise Hdf5Error('unsupported dataset datatype in %s' % self.name)
        count = int(np.prod(shape)) if len(shape) else 1
        nbytes = count * dt.size
        ver = layout[0]
        if ver != 3:
            raise Hdf5Error('data layout message version %d is not supported' % ver)
        cls = layout[1]
        if cls == 0:
            size = struct.unpack_from('<H', layout, 2)[0]
            raw = bytes(layout[4:4 + size])
        elif cls == 1:
            addr, size = struct.unpack_from('<QQ', layout, 2)
            raw = b'\0' * nbytes if addr == UNDEF else self._f._read(addr, nbytes)
        elif cls == 2:
            raw = self._f._read_chunked(layout, shape, dt)
        else:
            raise Hdf5Error('unknown layout class %d' % cls)
        return np.frombuffer(raw[:nbytes], dtype=dt.np_dtype).reshape(shape).copy()


class File(Node):
    def __init__(self, path):
        with open(path, 'rb') as fh:
            self._buf = fh.read()
        self._base = 0
        root = self._parse_superblock()
        Node.__init__(self, self, root, '/')

    # ---- low level ----------------------------------------------------------------------------------
    def _read(self, addr, n):
        a = self._base + addr
        if a < 0 or a + n > len(self._buf):
            raise Hdf5Error('read past end of file (addr %d, %d bytes)' % (addr, n))
        return self._buf[a:a + n]

    def _parse_superblock(self):
        sig = b'\x89HDF\r\n\x1a\n'
        off = 0
        while True:
            if self._buf[off:off + 8] == sig:
                break
            off = 512 if off == 0 else off * 2
            if off >= len(self._buf):
                raise Hdf5Error('not an HDF5 file')
        b = self._buf
        ver = b[off + 8]
        if ver in (0, 1):
            size_off, size_len = b[off + 13], b[off + 14]
            if size_off != 8 or size_len != 8:
                raise Hdf5Error('only 8-byte offsets/lengths are supported')
            p = off + 24 + (4 if ver == 1 else 0)
            base, _free, _eof, _drv = struct.unpack_from('<QQQQ', b, p)
            self._base = base if base != UNDEF else 0
            p += 32
            # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
            _lno, ohdr = struct.unpack_from('<QQ', b, p)
            return ohdr
        if ver in (2, 3):
            if b[off + 9] != 8 or b[off + 10] != 8:
                raise Hdf5Error('only 8-byte offsets/lengths are supported')
            base, _ext, _eof, root = struct.unpack_from('<QQQQ', b, off + 12)
            self._base = base if base != UNDEF else 0
            return root
        raise Hdf5Error('unsupported superblock version %d' % ver)

    # ---- object headers -----------------------------------------------------------------------------
    def _object_messages(self, addr):
        head = self._read(addr, 16)
        if head[:4] == b'OHDR':
            return self._object_messages_v2(addr)
        ver = head[0]
        if ver != 1:
            raise Hdf5Error('unsupported object header version %d' % ver)
        nmsg = struct.unpack_from('<H', head, 2)[0]
        hsize = struct.unpack_from('<I', head, 8)[0]
        msgs = []
        blocks = [(addr + 16, hsize)]
        while blocks and len(msgs) < nmsg + 64:
            baddr, blen = blocks.pop(0)
            data = self._read(baddr, blen)
            p = 0
            while p + 8 <= blen:
                mtype, msize, _flags = struct.unpack_from('<HHB', data, p)
                body = data[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x0010:
                    coff, clen = struct.unpack_from('<QQ', body, 0)
                    blocks.append((coff, clen))
                elif mtype != 0:
                    msgs.append((mtype, body))
        return msgs

    def _object_messages_v2(self, addr):
        head = self._read(addr, 6)
        flags = head[5]
        p = addr + 6
        if flags & 0x20:
            p += 16
        if flags & 0x10:
            p += 4
        nsz = 1 << (flags & 3)
        chunk0 = int.from_bytes(self._read(p, nsz), 'little')
        p += nsz
        msgs = []
        blocks = [(p, chunk0, True)]
        track = bool(flags & 0x04)
        while blocks:
            baddr, blen, first = blocks.pop(0)
            data = self._read(baddr, blen)
            q = 0
            if not first:
                if data[:4] != b'OCHK':
                    raise Hdf5Error('bad object header continuation')
                q = 4
            end = blen - 4  # checksum
            while q + 4 <= end:
                mtype = data[q]
                msize = struct.unpack_from('<H', data, q + 1)[0]
                q += 4 + (2 if track else 0)
                body = data[q:q + msize]
                q += msize
                if mtype == 0x10:
                    coff, clen = struct.unpack_from('<QQ', body, 0)
                    blocks.append((coff, clen, False))
                elif mtype != 0:
                    msgs.append((mtype, body))
        return msgs

    # ---- groups -------------------------------------------------------------------------------------
    def _heap_string(self, heap_data_addr, off):
        a = self._base + heap_data_addr + off
        e = self._buf.index(b'\0', a)
        return self._buf[a:e].decode('utf-8')

    def _read_symbol_table(self, btree, heap):
        h = self._read(heap, 32)
        if h[:4] != b'HEAP':
            raise Hdf5Error('bad local heap')
        data_addr = struct.unpack_from('<Q', h, 24)[0]
        out = {}

        def walk(addr):
            n = self._read(addr, 24)
            if n[:4] != b'TREE':
                raise Hdf5Error('bad B-tree node')
            ntype, level, used = n[4], n[5], struct.unpack_from('<H', n, 6)[0]
            if ntype != 0:
                raise Hdf5Error('unexpected B-tree node type')
            body = self._read(addr + 24, (2 * used + 1) * 8)
            for i in range(used):
                child = struct.unpack_from('<Q', body, 8 + 16 * i)[0]
                if level > 0:
                    walk(child)
                else:
                    s = self._read(child, 8)
                    if s[:4] != b'SNOD':
                        raise Hdf5Error('bad symbol table node')
                    nsym = struct.unpack_from('<H', s, 6)[0]
                    ents = self._read(child + 8, nsym * 40)
                    for k in range(nsym):
                        lno, ohdr = struct.unpack_from('<QQ', ents, 40 * k)
                        out[self._heap_string(data_addr, lno)] = ohdr

        walk(btree)
        return out

    def _parse_link(self, body):
        ver, flags = body[0], body[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = body[p]; p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        lsz = 1 << (flags & 3)
        nlen = int.from_bytes(body[p:p + lsz], 'little'); p += lsz
        name = bytes(body[p:p + nlen]).decode('utf-8'); p += nlen
        if ltype != 0:
            return name, None
        return name, struct.unpack_from('<Q', body, p)[0]

    # ---- datatypes / dataspaces / attributes -----------------------------------------------------------
    def _parse_datatype(self, b, p):
        cv = b[p]
        cls, ver = cv & 0x0F, cv >> 4
        bits = b[p + 1] | (b[p + 2] << 8) | (b[p + 3] << 16)
        size = struct.unpack_from('<I', b, p + 4)[0]
        q = p + 8
        if cls == 0:      # fixed point
            order = '>' if bits & 1 else '<'
            signed = bool(bits & 8)
            q += 4
            if size not in (1, 2, 4, 8):
                return _Dtype('other', size), q
            return _Dtype('int', size, np.dtype(order + ('i' if signed else 'u') + str(size))), q
        if cls == 1:      # floating point
            order = '>' if bits & 1 else '<'
            q += 12
            if size not in (2, 4, 8):
                return _Dtype('other', size), q
            return _Dtype('float', size, np.dtype(order + 'f' + str(size))), q
        if cls == 3:      # string
            return _Dtype('string', size, np.dtype('S%d' % size), pad=bits & 0x0F), q
        if cls == 9:      # variable length
            base, q2 = self._parse_datatype(b, q)
            is_str = (bits & 0x0F) == 1
            return _Dtype('vlen_str' if is_str else 'vlen', size, None, base=base), q2
        return _Dtype('other', size), q

    def _parse_dataspace(self, b):
        ver = b[0]
        rank = b[1]
        if ver == 1:
            p = 8
        elif ver == 2:
            p = 4
            if b[3] == 2:   # null dataspace
                return (0,)
        else:
            raise Hdf5Error('unsupported dataspace version %d' % ver)
        return tuple(struct.unpack_from('<Q', b, p + 8 * i)[0] for i in range(rank))

    def _global_heap_object(self, coll_addr, index):
        h = self._read(coll_addr, 16)
        if h[:4] != b'GCOL':
            raise Hdf5Error('bad global heap collection')
        csize = struct.unpack_from('<Q', h, 8)[0]
        data = self._read(coll_addr, csize)
        p = 16
        while p + 16 <= csize:
            idx, _ref, _res, osize = struct.unpack_from('<HHIQ', data, p)
            if idx == 0:
                break
            if idx == index:
                return bytes(data[p + 16:p + 16 + osize])
            p += 16 + ((osize + 7) & ~7)
        raise Hdf5Error('global heap object %d not found' % index)

    def _parse_attribute(self, b):
        ver = b[0]
        nsz, tsz, ssz = struct.unpack_from('<HHH', b, 2)
        p = 8
        if ver == 3:
            p = 9
        pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
        name = bytes(b[p:p + nsz]).split(b'\0')[0].decode('utf-8'); p += pad(nsz)
        dt, _ = self._parse_datatype(b, p); p += pad(tsz)
        shape = self._parse_dataspace(b[p:p + ssz]) if ssz else (); p += pad(ssz)
        count = int(np.prod(shape)) if len(shape) else 1
        data = b[p:]
        if dt.kind == 'vlen_str':
            vals = []
            for i in range(count):
                ln, addr, idx = struct.unpack_from('<IQI', data, 16 * i)
                vals.append(self._global_heap_object(addr, idx)[:ln] if addr not in (0, UNDEF) and ln else b'')
            vals = [v.decode('utf-8') for v in vals]
            return name, (vals[0] if not shape else np.array(vals, dtype=object).reshape(shape))
        if dt.kind == 'string':
            arr = np.frombuffer(bytes(data[:count * dt.size]), dtype=dt.np_dtype, count=count)
            vals = [bytes(v).split(b'\0')[0] if dt.pad in (0, 1) else bytes(v).rstrip(b' ') for v in arr]
            if not shape:
                return name, vals[0].decode('utf-8')
            return name, np.array(vals, dtype=object).reshape(shape)
        if dt.np_dtype is not None:
            arr = np.frombuffer(bytes(data[:count * dt.size]), dtype=dt.np_dtype, count=count)
            return name, (arr[0] if not shape else arr.reshape(shape).copy())
        return name, None

    # ---- chunked datasets without filters -----------------------------------------------------------
    def _read_chunked(self, layout, shape, dt):
        rank1 = layout[2]                      # dimensionality + 1
        btree = struct.unpack_from('<Q', layout, 3)[0]
        cdims = struct.unpack_from('<%dI' % rank1, layout, 11)
        chunk = cdims[:-1]
        out = np.zeros(shape, dtype=dt.np_dtype)
        if btree == UNDEF:
            return out.tobytes()
        cbytes = int(np.prod(chunk)) * dt.size

        def walk(addr):
            n = self._read(addr, 24)
            if n[:4] != b'TREE' or n[4] != 1:
                raise Hdf5Error('bad chunk B-tree node')
            level, used = n[5], struct.unpack_from('<H', n, 6)[0]
            ksz = 8 + 8 * rank1
            body = self._read(addr + 24, used * (ksz + 8) + ksz)
            for i in range(used):
                kp = i * (ksz + 8)
                csize, fmask = struct.unpack_from('<II', body, kp)
                offs = struct.unpack_from('<%dQ' % rank1, body, kp + 8)[:-1]
                child = struct.unpack_from('<Q', body, kp + ksz)[0]
                if level > 0:
                    walk(child)
                    continue
                if fmask != 0 or csize != cbytes:
                    raise Hdf5Error('filtered chunks are not supported')
                blk = np.frombuffer(self._read(child, cbytes), dtype=dt.np_dtype).reshape(chunk)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, shape))
                out[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]

        walk(btree)
        return out.tobytes()


def _attr_list(attrs, name):
    """Keras splits long name lists over ``name0``, ``name1`` ... (HDF5 64 KiB header-message limit)."""
    if name in attrs:
        v = attrs[name]
    else:
        parts, i = [], 0
        while '%s%d' % (name, i) in attrs:
            parts.extend(list(np.asarray(attrs['%s%d' % (name, i)]).ravel()))
            i += 1
        v = parts
    out = []
    for x in (np.asarray(v, dtype=object).ravel() if not isinstance(v, list) else v):
        out.append(x.decode('utf-8') if isinstance(x, bytes) else str(x))
    return out


def load_keras_h5(path):
    """-> (model_config_json_text, {layer_name: [np.ndarray, ...] in Keras weight order})."""
    from .keras_plan import NamedWeights
    f = File(path)
    attrs = f.attrs
    if 'model_config' not in attrs:
        raise Hdf5Error('%s has no model_config attribute (weights-only file?)' % path)
    cfg = attrs['model_config']
    if isinstance(cfg, bytes):
        cfg = cfg.decode('utf-8')
    g = f['model_weights'] if 'model_weights' in f else f
    weights = {}
    for lname in _attr_list(g.attrs, 'layer_names'):
        lg = g[lname]
        names = _attr_list(lg.attrs, 'weight_names')
        # a list of arrays that also carries the weight names: a nested model's list mixes the variables of all its layers
        # (keras_plan.inline_nested sorts them out by name)
        weights[lname] = NamedWeights([lg[wname].read() for wname in names], names)
    return cfg, weights

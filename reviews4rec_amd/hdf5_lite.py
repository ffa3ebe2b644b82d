"""Read-only HDF5 access for the reference's quick-data files, without h5py / libhdf5.

The reference's fast loader opens its epoch files with h5py (data_fast.py:10,31-45,79-87: ``h5py.File(path, 'r')``,
``len(f['a'])``, ``f['a'][:]``, ``f['a'][start:end]``) and data_scripts/make_quick_data.py:21-44 writes them: eight
datasets ``a`` .. ``h`` in the root group, ``i8`` / ``f8``, ``compression="gzip"``, filled one batch slice at a
time.  Neither h5py nor libhdf5 exists in the MI355X image, so this module reads that container format directly
(numpy + zlib + mmap), with the small part of h5py's surface the loader uses: ``File(path)`` as a context manager,
``f[name]`` (``/``-separated paths through old-style groups), ``len(ds)``, ``ds.shape``, ``ds.dtype``,
``ds[:]`` / ``ds[a:b]`` (first-axis slabs; anything further is applied to the slab by numpy).

What is read (HDF5 File Format Specification, version 1.x structures -- what libhdf5 writes under its default
``libver='earliest'`` bound, i.e. what h5py writes unless asked otherwise):
  * superblock versions 0 / 1 (2 / 3 are located, their root group must still be an old-style one);
  * old-style groups: symbol-table message -> version-1 B-tree of symbol-table nodes + local heap;
  * version-1 object headers with continuation blocks;
  * dataspace messages v1 / v2 (simple), fixed-point and IEEE floating-point datatypes of either byte order,
    fill-value messages (old and new), data-layout message v3 -- compact, contiguous, chunked (version-1 chunk
    B-tree of any depth; chunks that were never written read as the fill value; edge chunks are clipped) --
    and the filter pipeline v1 / v2 with deflate, shuffle and fletcher32 (the checksum is stripped, not checked).
Everything else (new-style groups, version-2 object headers, layout v4's chunk indices, szip / lzf / ...,
strings and compound types) raises ``Hdf5Error`` naming the feature, rather than returning wrong data.

Pinned by files written by the real h5py 3.3.0 / HDF5 1.10.6 (tests/golden/hdf5/, generator committed beside
them: make_golden_hdf5.py) -- tests/test_hdf5_lite.py.
"""
import mmap
import os
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
_FILTER_NAMES = {1: 'deflate', 2: 'shuffle', 3: 'fletcher32', 4: 'szip', 5: 'nbit', 6: 'scaleoffset', 32000: 'lzf'}


class Hdf5Error(RuntimeError):
    pass


class File:
    def __init__(self, path, mode='r'):
        if mode != 'r':
            raise Hdf5Error('hdf5_lite reads only (mode %r)' % (mode,))
        self.filename = path
        self._fh = open(path, 'rb')
        try:
            self._m = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ) if os.path.getsize(path) else b''
            self._superblock()
        except Exception:
            self.close()
            raise
        self._root = Group(self, self._root_header)

    # ---- little-endian field readers
    def _u(self, pos, n):
        if pos < 0 or pos + n > len(self._m):
            raise Hdf5Error('%s: structure at byte %d runs past the end of the file (truncated?)' % (self.filename, pos))
        return int.from_bytes(self._m[pos:pos + n], 'little')

    def _addr(self, pos):
        """An address field: file offset (base address applied), or None for the undefined address."""
        v = self._u(pos, self._so)
        return None if v == (1 << (8 * self._so)) - 1 else v + self._base

    def _superblock(self):
        m, at = self._m, 0
        while True:                                          # byte 0, 512, 1024, 2048, ... (a user block may precede it)
            if m[at:at + 8] == SIGNATURE:
                break
            at = 512 if at == 0 else at * 2
            if at + 8 > len(m):
                raise Hdf5Error('%s: not an HDF5 file (no superblock signature)' % self.filename)
        ver = m[at + 8]
        self._base = 0
        if ver in (0, 1):
            self._so, self._sl = m[at + 13], m[at + 14]
            pos = at + 24 + (4 if ver == 1 else 0)
            base = self._u(pos, self._so)
            pos += 4 * self._so                              # base, free-space info, end of file, driver info
            self._base = base
            self._root_header = self._addr(pos + self._so)   # root symbol-table entry: link-name offset, header address
        elif ver in (2, 3):
            self._so, self._sl = m[at + 9], m[at + 10]
            self._base = self._u(at + 12, self._so)
            self._root_header = self._addr(at + 12 + 3 * self._so)
        else:
            raise Hdf5Error('%s: superblock version %d' % (self.filename, ver))
        if self._so not in (4, 8) or self._sl not in (4, 8):
            raise Hdf5Error('%s: offsets / lengths of %d / %d bytes' % (self.filename, self._so, self._sl))
        if self._root_header is None:
            raise Hdf5Error('%s: no root group' % self.filename)

    def _messages(self, addr):
        """[(type, flags, data offset, data size)] of the version-1 object header at `addr`, continuation blocks included."""
        m = self._m
        if m[addr:addr + 4] == b'OHDR':
            raise Hdf5Error('%s: version-2 object header (file written with libver="latest"?): re-save it with the '
                            'default libver, or convert it with tools/hdf5_to_npz.py where h5py exists' % self.filename)
        if m[addr] != 1:
            raise Hdf5Error('%s: object header version %d at byte %d' % (self.filename, m[addr], addr))
        nmsg = self._u(addr + 2, 2)
        blocks, out = [(addr + 16, self._u(addr + 8, 4))], []
        while blocks and len(out) < nmsg:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, mflags = self._u(pos, 2), self._u(pos + 2, 2), m[pos + 4]
                data = pos + 8
                if mtype == 0x10:                            # continuation: (address, length) of the next block
                    blocks.append((self._addr(data), self._u(data + self._so, self._sl)))
                out.append((mtype, mflags, data, msize))
                pos = data + msize
        return out

    def __getitem__(self, name):
        return self._root[name]

    def __contains__(self, name):
        return name in self._root

    def keys(self):
        return self._root.keys()

    def __iter__(self):
        return iter(self._root.keys())

    def close(self):
        m = getattr(self, '_m', None)
        if m is not None and not isinstance(m, bytes):
            m.close()
        self._m = b''
        if self._fh is not None:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Group:
    def __init__(self, f, header):
        self._f, self._header = f, header
        self._links = None

    def _load(self):
        if self._links is not None:
            return self._links
        f = self._f
        sym = [x for x in f._messages(self._header) if x[0] == 0x11]
        if not sym:
            kinds = sorted({x[0] for x in f._messages(self._header)})
            if 0x02 in kinds or 0x06 in kinds:
                raise Hdf5Error('%s: new-style group (link messages): written with libver="latest"' % f.filename)
            raise Hdf5Error('%s: object at byte %d is not a group' % (f.filename, self._header))
        data = sym[0][2]
        btree, heap = f._addr(data), f._addr(data + f._so)
        if f._m[heap:heap + 4] != b'HEAP':
            raise Hdf5Error('%s: local heap signature missing at byte %d' % (f.filename, heap))
        names = f._addr(heap + 8 + 2 * f._sl)                # the heap's data segment
        links = {}

        def node(a):
            m = f._m
            if m[a:a + 4] != b'TREE' or m[a + 4] != 0:
                raise Hdf5Error('%s: group B-tree node expected at byte %d' % (f.filename, a))
            level, n = m[a + 5], f._u(a + 6, 2)
            p = a + 8 + 2 * f._so
            for i in range(n):
                child = f._addr(p + f._sl + i * (f._sl + f._so))
                if level:
                    node(child)
                    continue
                if m[child:child + 4] != b'SNOD':
                    raise Hdf5Error('%s: symbol-table node expected at byte %d' % (f.filename, child))
                for k in range(f._u(child + 6, 2)):
                    e = child + 8 + k * (2 * f._so + 24)
                    s = names + f._u(e, f._so)
                    links[bytes(m[s:m.find(b'\0', s)]).decode('utf-8')] = f._addr(e + f._so)
        if btree is not None:
            node(btree)
        self._links = links
        return links

    def keys(self):
        return sorted(self._load())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, name):
        obj = self
        for part in [p for p in name.split('/') if p]:
            if not isinstance(obj, Group):
                raise KeyError(name)
            links = obj._load()
            if part not in links:
                raise KeyError("%s: no object %r (has: %s)" % (self._f.filename, name, ', '.join(sorted(links))))
            header = links[part]
            kinds = {x[0] for x in self._f._messages(header)}
            obj = Dataset(self._f, header, part) if 0x08 in kinds else Group(self._f, header)
        return obj


class Dataset:
    def __init__(self, f, header, name):
        self._f, self.name = f, name
        self._filters, self._fill, self._layout = [], None, None
        self.shape = self.dtype = None
        old_fill = None
        for mtype, _, d, size in f._messages(header):
            m = f._m
            if mtype == 0x01:                                # dataspace
                ver, rank, flags = m[d], m[d + 1], m[d + 2]
                if ver not in (1, 2) or (ver == 2 and m[d + 3] == 2):
                    raise Hdf5Error('%s/%s: dataspace version %d / null dataspace' % (f.filename, name, ver))
                p = d + (8 if ver == 1 else 4)
                self.shape = tuple(f._u(p + i * f._sl, f._sl) for i in range(rank))
            elif mtype == 0x03:                              # datatype
                cls, bits, nbytes = m[d] & 15, m[d + 1], f._u(d + 4, 4)
                order = '>' if bits & 1 else '<'
                if cls == 0:
                    self.dtype = np.dtype('%s%s%d' % (order, 'i' if bits & 8 else 'u', nbytes))
                elif cls == 1 and not bits & 0x40 and nbytes in (2, 4, 8):
                    self.dtype = np.dtype('%sf%d' % (order, nbytes))
                else:
                    raise Hdf5Error('%s/%s: datatype class %d (only integers and IEEE floats are read)' % (f.filename, name, cls))
            elif mtype == 0x04:                              # fill value, old form
                n = f._u(d, 4)
                old_fill = bytes(m[d + 4:d + 4 + n]) if n else None
            elif mtype == 0x05:                              # fill value
                ver = m[d]
                if ver in (1, 2):
                    if ver == 1 or m[d + 3]:
                        n = f._u(d + 4, 4)
                        self._fill = bytes(m[d + 8:d + 8 + n]) if n else None
                elif ver == 3:
                    if m[d + 1] & 0x20:
                        n = f._u(d + 2, 4)
                        self._fill = bytes(m[d + 6:d + 6 + n]) if n else None
                else:
                    raise Hdf5Error('%s/%s: fill-value message version %d' % (f.filename, name, ver))
            elif mtype == 0x08:                              # data layout
                ver, cls = m[d], m[d + 1]
                if ver != 3:
                    raise Hdf5Error('%s/%s: data-layout message version %d (3 is read; 4 = libver="latest")' % (f.filename, name, ver))
                if cls == 0:
                    self._layout = ('compact', d + 4, f._u(d + 2, 2))
                elif cls == 1:
                    self._layout = ('contiguous', f._addr(d + 2), f._u(d + 2 + f._so, f._sl))
                elif cls == 2:
                    nd = m[d + 2]
                    dims = tuple(f._u(d + 3 + f._so + 4 * i, 4) for i in range(nd))
                    self._layout = ('chunked', f._addr(d + 3), dims)
                else:
                    raise Hdf5Error('%s/%s: layout class %d' % (f.filename, name, cls))
            elif mtype == 0x0B:                              # filter pipeline
                ver, nf = m[d], m[d + 1]
                if ver not in (1, 2):
                    raise Hdf5Error('%s/%s: filter-pipeline message version %d' % (f.filename, name, ver))
                p = d + (8 if ver == 1 else 2)
                for _ in range(nf):
                    fid = f._u(p, 2)
                    if ver == 1 or fid >= 256:
                        nlen = f._u(p + 2, 2)
                        p += 4
                    else:
                        nlen = 0
                        p += 2
                    ncd = f._u(p + 2, 2)
                    p += 4
                    p += (nlen + 7) // 8 * 8 if ver == 1 else nlen
                    cd = [f._u(p + 4 * i, 4) for i in range(ncd)]
                    p += 4 * ncd + (4 if ver == 1 and ncd & 1 else 0)
                    self._filters.append((fid, cd))
        if self.shape is None or self.dtype is None or self._layout is None:
            raise Hdf5Error('%s/%s: not a dataset (dataspace / datatype / layout message missing)' % (f.filename, name))
        if self._fill is None:
            self._fill = old_fill
        if self._fill is not None and len(self._fill) != self.dtype.itemsize:
            self._fill = None
        for fid, _ in self._filters:
            if fid not in (1, 2, 3):
                raise Hdf5Error('%s/%s: filter %d (%s) is not read: deflate, shuffle and fletcher32 are' %
                                (f.filename, name, fid, _FILTER_NAMES.get(fid, 'third-party')))
        if self._layout[0] == 'chunked' and (len(self._layout[2]) != len(self.shape) + 1 or self._layout[2][-1] != self.dtype.itemsize):
            raise Hdf5Error('%s/%s: chunk dimensionality %r does not match shape %r' % (f.filename, name, self._layout[2], self.shape))

    def __len__(self):
        if not self.shape:
            raise TypeError('len() of a scalar dataset')
        return self.shape[0]

    @property
    def chunks(self):
        return self._layout[2][:-1] if self._layout[0] == 'chunked' else None

    # ---- reading
    def _native(self):
        return self.dtype.newbyteorder('=')

    def _blank(self, shape):
        out = np.empty(shape, self._native())
        out[...] = np.frombuffer(self._fill, self.dtype)[0] if self._fill is not None else 0
        return out

    def _decode(self, raw, mask):
        for i in range(len(self._filters) - 1, -1, -1):
            if mask >> i & 1:
                continue
            fid, cd = self._filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:
                width = cd[0] if cd else self.dtype.itemsize
                n = len(raw) // width
                if width > 1 and n > 1:
                    body = np.frombuffer(raw, np.uint8, n * width).reshape(width, n).T.tobytes()
                    raw = body + bytes(raw[n * width:])
            elif fid == 3:
                raw = raw[:-4]
        return raw

    def _chunk_list(self, lo, hi):
        """[(file offset, stored bytes, filter mask, chunk origin)] of the chunks that overlap rows [lo, hi)."""
        f, (_, root, cdims) = self._f, self._layout
        nd, out = len(cdims), []
        ksz = 8 + 8 * nd

        def node(a):
            m = f._m
            if m[a:a + 4] != b'TREE' or m[a + 4] != 1:
                raise Hdf5Error('%s/%s: chunk B-tree node expected at byte %d' % (f.filename, self.name, a))
            level, n = m[a + 5], f._u(a + 6, 2)
            p = a + 8 + 2 * f._so
            for i in range(n):
                key = p + i * (ksz + f._so)                  # key i, child i, key i + 1, ...: keys ascend, slowest axis first
                first, nxt = f._u(key + 8, 8), f._u(key + ksz + f._so + 8, 8)
                if first >= hi:
                    break                                    # this child and every later one start past the slab
                if level:
                    if nxt + cdims[0] > lo:                  # (child i holds chunks in [key i, key i + 1])
                        node(f._addr(key + ksz))
                elif first + cdims[0] > lo:
                    origin = tuple(f._u(key + 8 + 8 * k, 8) for k in range(nd - 1))
                    out.append((f._addr(key + ksz), f._u(key, 4), f._u(key + 4, 4), origin))
        if root is not None:
            node(root)
        return out

    def _read_rows(self, lo, hi):
        f, kind = self._f, self._layout[0]
        shape = (hi - lo,) + self.shape[1:]
        count = int(np.prod(shape))
        if count == 0:
            return np.empty(shape, self._native())
        row = int(np.prod(self.shape[1:])) * self.dtype.itemsize
        if kind in ('compact', 'contiguous'):
            start = self._layout[1]
            if start is None:                                # storage never allocated
                return self._blank(shape)
            raw = f._m[start + lo * row:start + hi * row]
            if len(raw) != count * self.dtype.itemsize:
                raise Hdf5Error('%s/%s: data runs past the end of the file (truncated?)' % (f.filename, self.name))
            return np.frombuffer(raw, self.dtype).reshape(shape).astype(self._native())
        cdims = self._layout[2][:-1]
        out = self._blank(shape)
        chunks = self._chunk_list(lo, hi)

        def place(c):
            addr, nbytes, mask, origin = c
            raw = self._decode(f._m[addr:addr + nbytes], mask)
            if len(raw) < int(np.prod(cdims)) * self.dtype.itemsize:
                raise Hdf5Error('%s/%s: chunk at %r decodes to %d bytes' % (f.filename, self.name, origin, len(raw)))
            block = np.frombuffer(raw, self.dtype, int(np.prod(cdims))).reshape(cdims)
            src, dst = [], []
            for ax, (o, c_, s) in enumerate(zip(origin, cdims, self.shape)):
                a0, a1 = (max(o, lo), min(o + c_, hi)) if ax == 0 else (o, min(o + c_, s))
                if a1 <= a0:
                    return
                src.append(slice(a0 - o, a1 - o))
                dst.append(slice(a0 - (lo if ax == 0 else 0), a1 - (lo if ax == 0 else 0)))
            out[tuple(dst)] = block[tuple(src)]
        if len(chunks) > 4 and any(fid == 1 for fid, _ in self._filters):
            with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as pool:     # zlib releases the GIL
                list(pool.map(place, chunks))
        else:
            for c in chunks:
                place(c)
        return out

    def __getitem__(self, key):
        if not self.shape:                                   # scalar dataset
            if key not in ((), Ellipsis):
                raise IndexError('scalar dataset')
            kind = self._layout[0]
            if kind == 'chunked':
                raise Hdf5Error('%s/%s: chunked scalar dataset' % (self._f.filename, self.name))
            start = self._layout[1]
            if start is None:
                return self._blank(())[()]
            return np.frombuffer(self._f._m[start:start + self.dtype.itemsize], self.dtype)[0].astype(self._native())
        rest = ()
        if isinstance(key, tuple):
            key, rest = (key[0], key[1:]) if key else (slice(None), ())
        if key is Ellipsis:
            key = slice(None)
        n = self.shape[0]
        if isinstance(key, (int, np.integer)):
            i = int(key) + (n if key < 0 else 0)
            if not 0 <= i < n:
                raise IndexError('index %d out of range for %d rows' % (key, n))
            got = self._read_rows(i, i + 1)[0]
        elif isinstance(key, slice):
            lo, hi, step = key.indices(n)
            if step != 1:
                got = self._read_rows(0, n)[key]
            else:
                got = self._read_rows(lo, max(lo, hi))
        else:
            got = self._read_rows(0, n)[key]
        if not rest:
            return got
        return got[rest] if isinstance(key, (int, np.integer)) else got[(slice(None),) + rest]


# ----------------------------------------------------------------------------------------------------- writing
# The counterpart of data_scripts/make_quick_data.py:21-44's h5py calls: a file of root-group datasets that libhdf5 /
# h5py (and the reference's data_fast.py) read.  Same structures the reader above takes apart, laid out the way
# libhdf5 1.10 lays out a default (libver='earliest') file -- the byte patterns of the messages are those of the
# h5py-written fixtures under tests/golden/hdf5/.  Chunks are cut along the first axis only (whole rows, ~1 MB),
# which is what a loader that reads row slabs wants.
_UNDEF = b'\xff' * 8
_GROUP_INTERNAL_K, _CHUNK_K = 16, 32                         # libhdf5's defaults (superblock v0 stores only the group ones)


def _u64(*v):
    return b''.join(int(x).to_bytes(8, 'little') for x in v)


def _datatype_message(dt):
    dt = np.dtype(dt)
    if dt.byteorder == '>':
        raise Hdf5Error('write: big-endian arrays are not written (%s)' % dt)
    n = dt.itemsize
    if dt.kind in 'iu':
        body = bytes([0x10, 0x08 if dt.kind == 'i' else 0x00, 0, 0]) + n.to_bytes(4, 'little') + (0).to_bytes(2, 'little') + (8 * n).to_bytes(2, 'little')
    elif dt.kind == 'f' and n in (4, 8):
        sign, eloc, esz, msz, bias = (63, 52, 11, 52, 1023) if n == 8 else (31, 23, 8, 23, 127)
        body = bytes([0x11, 0x20, sign, 0]) + n.to_bytes(4, 'little') + (0).to_bytes(2, 'little') + (8 * n).to_bytes(2, 'little') + \
            bytes([eloc, esz, 0, msz]) + bias.to_bytes(4, 'little')
    else:
        raise Hdf5Error('write: dtype %s (integers, float32 and float64 are written)' % dt)
    return body


def _message(mtype, flags, body):
    body = body + b'\0' * (-len(body) % 8)
    return mtype.to_bytes(2, 'little') + len(body).to_bytes(2, 'little') + bytes([flags, 0, 0, 0]) + body


def _object_header(messages):
    body = b''.join(messages)
    return bytes([1, 0]) + len(messages).to_bytes(2, 'little') + (1).to_bytes(4, 'little') + len(body).to_bytes(4, 'little') + b'\0' * 4 + body


class _Out:
    def __init__(self, fh):
        self.fh = fh

    def put(self, data):
        """Append `data` at the next 8-byte boundary; -> its address."""
        at = self.fh.tell()
        pad = -at % 8
        if pad:
            self.fh.write(b'\0' * pad)
        self.fh.write(data)
        return at + pad


def _chunk_btree(out, entries, rank, itemsize):
    """Version-1 B-tree (node type 1) over `entries` = [(address, stored bytes, origin)] in ascending origin order;
    -> root address.  Nodes are written at full size (2K children, 2K + 1 keys), as libhdf5 reads them."""
    nd = rank + 1
    ksz, cap = 8 + 8 * nd, 2 * _CHUNK_K

    def key(nbytes, origin, last=0):
        return int(nbytes).to_bytes(4, 'little') + (0).to_bytes(4, 'little') + _u64(*origin, last)
    # level 0: (first key, final key, address) per node, then levels of nodes over nodes until one is left
    final = key(0, entries[-1][2], itemsize)                 # (what libhdf5 leaves there: the last origin, element size in the extra slot)
    level, nodes = 0, [(key(n, o), a) for a, n, o in entries]
    while True:
        groups = [nodes[i:i + cap] for i in range(0, len(nodes), cap)]
        size = 24 + cap * 8 + (cap + 1) * ksz
        base = out.put(b'')                                  # the level's nodes are consecutive: siblings are known up front
        made = []
        for g, grp in enumerate(groups):
            right_key = groups[g + 1][0][0] if g + 1 < len(groups) else final
            body = b'TREE' + bytes([1, level]) + len(grp).to_bytes(2, 'little')
            body += (_u64(base + (g - 1) * size) if g else _UNDEF) + (_u64(base + (g + 1) * size) if g + 1 < len(groups) else _UNDEF)
            body += b''.join(k + _u64(a) for k, a in grp) + right_key
            body += b'\0' * (size - len(body))
            assert size % 8 == 0
            addr = out.put(body)
            assert addr == base + g * size
            made.append((grp[0][0], addr))
        if len(made) == 1:
            return made[0][1]
        nodes, level = made, level + 1


def write_file(path, datasets, compression='gzip', level=4, chunk_bytes=1 << 20):
    """Write `datasets` (name -> array; names without '/') as the root-group datasets of a new HDF5 file:
    ``compression='gzip'`` -- chunked + deflate like make_quick_data.py:23-32 -- or ``None`` (contiguous)."""
    if compression not in ('gzip', None):
        raise Hdf5Error('write: compression %r (gzip or None)' % (compression,))
    names = sorted(datasets, key=lambda s: s.encode('utf-8'))
    for nm in names:
        if not nm or '/' in nm or '\0' in nm:
            raise Hdf5Error('write: dataset name %r' % nm)
    tmp = path + '.part'
    with open(tmp, 'wb') as fh:
        out = _Out(fh)
        fh.write(b'\0' * 96)                                 # the superblock, filled in last
        headers = {}
        for nm in names:
            a = np.ascontiguousarray(datasets[nm])
            if a.dtype.byteorder == '>':
                a = a.astype(a.dtype.newbyteorder('<'))
            if a.ndim < 1:
                raise Hdf5Error('write: %r is a scalar (arrays of one or more axes are written)' % nm)
            dims = a.shape
            space = bytes([1, a.ndim, 1, 0, 0, 0, 0, 0]) + _u64(*dims) + _u64(*dims)      # maxshape = shape
            msgs = [_message(0x01, 0, space), _message(0x03, 1, _datatype_message(a.dtype))]
            rowbytes = int(np.prod(dims[1:])) * a.itemsize
            if compression is None:
                raw = a.tobytes()
                addr = out.put(raw) if raw else None
                msgs.append(_message(0x05, 1, bytes([2, 2, 2, 1, 0, 0, 0, 0])))
                msgs.append(_message(0x08, 0, bytes([3, 1]) + (_u64(addr) if raw else _UNDEF) + _u64(len(raw))))
            else:
                rows = max(1, min(max(dims[0], 1), chunk_bytes // max(rowbytes, 1)))
                cdims = (rows,) + tuple(max(d, 1) for d in dims[1:])
                entries = []
                if a.size:
                    def pack(r0):
                        blk = a[r0:r0 + rows]
                        if len(blk) < rows:                  # an edge chunk is stored whole
                            blk = np.concatenate([blk, np.zeros((rows - len(blk),) + dims[1:], a.dtype)])
                        return zlib.compress(blk.tobytes(), level)
                    starts = list(range(0, dims[0], rows))
                    with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as pool:
                        for g0 in range(0, len(starts), 64):                 # bounded: 64 chunks in flight
                            for r0, z in zip(starts[g0:g0 + 64], pool.map(pack, starts[g0:g0 + 64])):
                                entries.append((out.put(z), len(z), (r0,) + (0,) * (a.ndim - 1)))
                root = _chunk_btree(out, entries, a.ndim, a.itemsize) if entries else None
                msgs.append(_message(0x05, 1, bytes([2, 3, 0, 1, 0, 0, 0, 0])))
                msgs.append(_message(0x0B, 1, bytes([1, 1, 0, 0, 0, 0, 0, 0]) + bytes([1, 0, 8, 0, 1, 0, 1, 0]) + b'deflate\0' +
                                     int(level).to_bytes(4, 'little') + b'\0' * 4))
                msgs.append(_message(0x08, 0, bytes([3, 2, a.ndim + 1]) + (_u64(root) if root is not None else _UNDEF) +
                                     b''.join(int(c).to_bytes(4, 'little') for c in cdims + (a.itemsize,))))
            headers[nm] = out.put(_object_header(msgs))
        # root group: local heap (names), one symbol-table node (its K sized for the links), a one-entry B-tree
        heap_data, offs = bytearray(8), {}
        for nm in names:
            offs[nm] = len(heap_data)
            b = nm.encode('utf-8') + b'\0'
            heap_data += b + b'\0' * (-len(b) % 8)
        free_at = len(heap_data)
        heap_data += _u64(1, 16)                             # one free block, end of list: how libhdf5 closes a heap
        data_addr = out.put(bytes(heap_data))
        heap = out.put(b'HEAP' + b'\0' * 4 + _u64(len(heap_data), free_at, data_addr))
        leaf_k = max(4, (len(names) + 1) // 2)
        snod = b'SNOD' + bytes([1, 0]) + len(names).to_bytes(2, 'little')
        snod += b''.join(_u64(offs[nm], headers[nm]) + b'\0' * 24 for nm in names)
        snod += b'\0' * (8 + 2 * leaf_k * 40 - len(snod))
        snod_addr = out.put(snod)
        tree = b'TREE' + bytes([0, 0]) + (1 if names else 0).to_bytes(2, 'little') + _UNDEF + _UNDEF
        tree += (_u64(0, snod_addr, offs[names[-1]]) if names else b'')
        tree += b'\0' * (24 + 2 * _GROUP_INTERNAL_K * 8 + (2 * _GROUP_INTERNAL_K + 1) * 8 - len(tree))
        tree_addr = out.put(tree)
        root_header = out.put(_object_header([_message(0x11, 0, _u64(tree_addr, heap))]))
        eof = out.put(b'')
        fh.truncate(eof)
        sb = SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + leaf_k.to_bytes(2, 'little') + _GROUP_INTERNAL_K.to_bytes(2, 'little') + b'\0' * 4
        sb += _u64(0) + _UNDEF + _u64(eof) + _UNDEF
        sb += _u64(0, root_header) + (1).to_bytes(4, 'little') + b'\0' * 4 + _u64(tree_addr, heap)
        assert len(sb) == 96
        fh.seek(0)
        fh.write(sb)
    os.replace(tmp, path)

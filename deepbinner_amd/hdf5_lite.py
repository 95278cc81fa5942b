"""
hdf5_lite — a small, dependency-free reader for the two HDF5 dialects on Deepbinner's classify
path: Keras-2.1.4 model files (weights + ``model_config``) and Oxford Nanopore fast5 files
(single-read old/new layout and multi-read).

The reference reads both through h5py (``deepbinner/load_fast5s.py:19,27-46``,
``deepbinner/classify.py:22,90,249``); h5py is not part of this stack, so this module implements
just the slice of the HDF5 file format those files use:

* superblock v0/v1 (and v2/v3 for newer writers), 8-byte offsets/lengths;
* version-1 and version-2 object headers with continuation blocks;
* old-style groups (symbol-table message -> v1 B-tree -> SNOD nodes + local heap);
* new-style groups: compact (link messages in the header) and dense (fractal heap indexed by a
  version-2 B-tree on link names);
* datasets: compact / contiguous / chunked with deflate, shuffle and fletcher32 filters; the chunk
  index is a v1 B-tree (layout message v1-v3) or, in files written with ``libver='latest'`` by
  HDF5 >= 1.10 (layout message v4), the single chunk itself, an implicit run, a fixed array or an
  extensible array (with their paged and super-block forms);
* attributes (v1-v3) of fixed-point, float, fixed-length string and variable-length string type
  (global heap).

It mirrors the small part of the h5py API the reference uses: ``File(path)``, ``.keys()``,
``.values()``, ``[name]`` with ``/``-separated paths, ``.attrs[...]`` and ``dataset[:]``.
The reader is pinned to the real library: ``oracle/make_h5py_fixtures.py`` writes the variants
with h5py (the image has one interpreter with it), ``tests/test_fast5_native.py`` reads them back.
Anything outside that slice raises ``OSError`` — the same exception class h5py raises for an
unreadable file, which ``load_fast5s.get_read_id_and_signal`` turns into ``(None, None)``.
"""

import mmap
import zlib

import numpy as np

_SIGNATURE = b'\x89HDF\r\n\x1a\n'
_UNDEF = 0xFFFFFFFFFFFFFFFF


class Hdf5FormatError(OSError):
    pass


def _u(buf, off, size):
    return int.from_bytes(buf[off:off + size], 'little')


def _pad8(n):
    return (n + 7) & ~7


class _Datatype:
    """Decoded datatype message (only the classes the model/fast5 files use)."""

    def __init__(self, buf, off):
        b0 = buf[off]
        self.cls = b0 & 0x0F
        self.version = b0 >> 4
        bits = _u(buf, off + 1, 3)
        self.size = _u(buf, off + 4, 4)
        self.numpy = None
        self.vlen_string = False
        self.string = False
        if self.cls == 0:  # fixed point
            order = '>' if bits & 1 else '<'
            signed = bool(bits & 0x08)
            self.numpy = np.dtype('%s%s%d' % (order, 'i' if signed else 'u', self.size))
        elif self.cls == 1:  # floating point
            order = '>' if bits & 1 else '<'
            self.numpy = np.dtype('%sf%d' % (order, self.size))
        elif self.cls == 3:  # fixed-length string
            self.string = True
            self.numpy = np.dtype('S%d' % self.size)
        elif self.cls == 9:  # variable length
            if (bits & 0x0F) == 1:
                self.vlen_string = True
            else:
                raise Hdf5FormatError('variable-length sequences are not supported')
        else:
            raise Hdf5FormatError('unsupported datatype class %d' % self.cls)


def _parse_dataspace(buf, off, len_size):
    version = buf[off]
    rank = buf[off + 1]
    flags = buf[off + 2]
    if version == 1:
        p = off + 8
    elif version == 2:
        p = off + 4
    else:
        raise Hdf5FormatError('unsupported dataspace version %d' % version)
    dims = tuple(_u(buf, p + i * len_size, len_size) for i in range(rank))
    p += rank * len_size
    maxdims = None
    if flags & 1:
        maxdims = tuple(_u(buf, p + i * len_size, len_size) for i in range(rank))
    return dims, maxdims


class _Message:
    __slots__ = ('type', 'off', 'size', 'flags')

    def __init__(self, mtype, off, size, flags):
        self.type, self.off, self.size, self.flags = mtype, off, size, flags


class AttributeManager:
    def __init__(self, obj):
        self._obj = obj
        self._cache = None

    def _load(self):
        if self._cache is None:
            self._cache = {}
            f = self._obj._file
            for m in self._obj._messages:
                if m.type == 0x000C:
                    name, value = f._parse_attribute(m.off)
                    self._cache[name] = value
                elif m.type == 0x0015:
                    # Dense attribute storage: attributes live in a fractal heap.
                    flags = f._buf[m.off + 1]
                    p = m.off + 2 + (2 if flags & 1 else 0)
                    heap_addr = _u(f._buf, p, f._O)
                    index_addr = _u(f._buf, p + f._O, f._O)
                    if heap_addr != _UNDEF and index_addr != _UNDEF:
                        for obj_off in f._dense_objects(heap_addr, index_addr):
                            name, value = f._parse_attribute(obj_off)
                            self._cache[name] = value
        return self._cache

    def __getitem__(self, name):
        return self._load()[name]

    def __contains__(self, name):
        return name in self._load()

    def keys(self):
        return self._load().keys()

    def items(self):
        return self._load().items()

    def get(self, name, default=None):
        return self._load().get(name, default)


class _Object:
    def __init__(self, hfile, addr, name):
        self._file = hfile
        self._addr = addr
        self.name = name
        self._messages = hfile._read_object_header(addr)
        self.attrs = AttributeManager(self)

    def _first(self, mtype):
        for m in self._messages:
            if m.type == mtype:
                return m
        return None


class Group(_Object):
    def __init__(self, hfile, addr, name):
        super().__init__(hfile, addr, name)
        self._links = None

    def _load_links(self):
        if self._links is not None:
            return self._links
        f = self._file
        buf = f._buf
        links = {}
        stab = self._first(0x0011)
        if stab is not None:
            btree = _u(buf, stab.off, f._O)
            heap = _u(buf, stab.off + f._O, f._O)
            f._walk_group_btree(btree, f._local_heap_data(heap), links)
        for m in self._messages:
            if m.type == 0x0006:
                name, addr = f._parse_link(m.off)
                if addr is not None:
                    links[name] = addr
        linfo = self._first(0x0002)
        if linfo is not None:
            flags = buf[linfo.off + 1]
            p = linfo.off + 2 + (8 if flags & 1 else 0)
            heap_addr = _u(buf, p, f._O)
            index_addr = _u(buf, p + f._O, f._O)
            if heap_addr != _UNDEF and index_addr != _UNDEF:
                for obj_off in f._dense_objects(heap_addr, index_addr):
                    name, addr = f._parse_link(obj_off)
                    if addr is not None:
                        links[name] = addr
        self._links = dict(sorted(links.items()))
        return self._links

    def keys(self):
        return self._load_links().keys()

    def __iter__(self):
        return iter(self._load_links())

    def __len__(self):
        return len(self._load_links())

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def values(self):
        return [self[k] for k in self._load_links()]

    def items(self):
        return [(k, self[k]) for k in self._load_links()]

    def __getitem__(self, path):
        parts = [p for p in path.split('/') if p]
        node = self._file.root if path.startswith('/') else self
        for part in parts:
            if not isinstance(node, Group):
                raise KeyError(path)
            links = node._load_links()
            if part not in links:
                raise KeyError("Unable to open object (component not found: %r)" % part)
            base = node.name.rstrip('/')
            node = node._file._open(links[part], base + '/' + part)
        return node


class Dataset(_Object):
    def __init__(self, hfile, addr, name):
        super().__init__(hfile, addr, name)
        f = hfile
        dt = self._first(0x0003)
        ds = self._first(0x0001)
        if dt is None or ds is None:
            raise Hdf5FormatError('dataset without datatype/dataspace')
        self._dtype = _Datatype(f._buf, dt.off)
        self.shape, self.maxshape = _parse_dataspace(f._buf, ds.off, f._L)

    @property
    def dtype(self):
        return self._dtype.numpy

    def __len__(self):
        return self.shape[0]

    def _filters(self):
        m = self._first(0x000B)
        if m is None:
            return []
        buf = self._file._buf
        version = buf[m.off]
        n = buf[m.off + 1]
        p = m.off + (8 if version == 1 else 2)
        out = []
        for _ in range(n):
            fid = _u(buf, p, 2)
            p += 2
            if version == 1 or fid >= 256:
                name_len = _u(buf, p, 2)
                p += 2
            else:
                name_len = 0
            p += 2  # flags
            ncd = _u(buf, p, 2)
            p += 2
            p += _pad8(name_len) if version == 1 else name_len
            cd = [_u(buf, p + 4 * i, 4) for i in range(ncd)]
            p += 4 * ncd
            if version == 1 and ncd % 2 == 1:
                p += 4
            out.append((fid, cd))
        return out

    def read(self):
        f = self._file
        buf = f._buf
        if self._dtype.numpy is None:
            raise Hdf5FormatError('unsupported dataset datatype')
        dt = self._dtype.numpy
        count = int(np.prod(self.shape)) if self.shape else 1
        # deflate expands at most 1032-fold: a dataset cannot be much larger than that many times
        # the file (a damaged dataspace would otherwise ask numpy for terabytes)
        if count * dt.itemsize // 1100 > len(buf) + 4096:
            raise Hdf5FormatError('implausible dataset size')
        lay = self._first(0x0008)
        if lay is None:
            raise Hdf5FormatError('dataset without layout message')
        version = buf[lay.off]
        if version in (1, 2):
            ndim = buf[lay.off + 1]
            cls = buf[lay.off + 2]
            p = lay.off + 8
            addr = None
            if cls != 0:
                addr = _u(buf, p, f._O)
                p += f._O
            dims = [_u(buf, p + 4 * i, 4) for i in range(ndim)]
            p += 4 * ndim
            if cls == 0:
                size = _u(buf, p, 4)
                raw = bytes(buf[p + 4:p + 4 + size])
                return np.frombuffer(raw, dtype=dt, count=count).reshape(self.shape).copy()
            if cls == 1:
                return self._read_contiguous(addr, count, dt)
            return self._read_chunked(addr, dims[:-1], dt)
        if version == 3:
            cls = buf[lay.off + 1]
            p = lay.off + 2
            if cls == 0:
                size = _u(buf, p, 2)
                raw = bytes(buf[p + 2:p + 2 + size])
                return np.frombuffer(raw, dtype=dt, count=count).reshape(self.shape).copy()
            if cls == 1:
                addr = _u(buf, p, f._O)
                return self._read_contiguous(addr, count, dt)
            if cls == 2:
                ndim = buf[p]
                addr = _u(buf, p + 1, f._O)
                q = p + 1 + f._O
                dims = [_u(buf, q + 4 * i, 4) for i in range(ndim)]
                return self._read_chunked(addr, dims[:-1], dt)
        if version == 4:
            cls = buf[lay.off + 1]
            p = lay.off + 2
            if cls == 0:
                size = _u(buf, p, 2)
                raw = bytes(buf[p + 2:p + 2 + size])
                return np.frombuffer(raw, dtype=dt, count=count).reshape(self.shape).copy()
            if cls == 1:
                addr = _u(buf, p, f._O)
                return self._read_contiguous(addr, count, dt)
            if cls == 2:
                return self._read_chunked_v4(p, dt)
        raise Hdf5FormatError('unsupported data layout (version %d)' % version)

    def _read_chunked_v4(self, p, dt):
        """Layout message version 4 (HDF5 1.10, libver='latest'): the chunk index is no longer a
        v1 B-tree but one of: the single chunk itself, an implicit run of equal chunks, a fixed
        array, an extensible array (one unlimited dimension) or a v2 B-tree (several; not
        handled - raw signals are one-dimensional)."""
        f = self._file
        buf = f._buf
        flags = buf[p]
        ndim = buf[p + 1]
        enc = buf[p + 2]
        p += 3
        dims = [_u(buf, p + enc * i, enc) for i in range(ndim)]
        p += enc * ndim
        chunk_dims = dims[:-1]
        if len(chunk_dims) != len(self.shape) or any(c < 1 for c in chunk_dims):
            raise Hdf5FormatError('chunk rank does not match the dataset')
        index_type = buf[p]
        p += 1
        chunk_bytes = int(np.prod(chunk_dims)) * dt.itemsize
        grid = [-(-s // c) for s, c in zip(self.shape, chunk_dims)]
        n_chunks = int(np.prod(grid)) if grid else 1

        def offsets_of(k):
            out = []
            for g, c in zip(reversed(grid), reversed(chunk_dims)):
                out.append((k % g) * c)
                k //= g
            return out[::-1]

        if index_type == 1:                      # single chunk
            size, mask = chunk_bytes, 0
            if flags & 2:                        # ... with its filtered size and filter mask
                size = _u(buf, p, f._L)
                mask = _u(buf, p + f._L, 4)
                p += f._L + 4
            addr = _u(buf, p, f._O)
            entries = [] if addr == _UNDEF else [(offsets_of(0), mask, addr, size)]
        elif index_type == 2:                    # implicit: all chunks allocated, back to back
            addr = _u(buf, p, f._O)
            entries = [] if addr == _UNDEF else \
                [(offsets_of(k), 0, addr + k * chunk_bytes, chunk_bytes) for k in range(n_chunks)]
        elif index_type == 3:                    # fixed array
            addr = _u(buf, p + 1, f._O)
            entries = [] if addr == _UNDEF else \
                [(offsets_of(k),) + e for k, e in enumerate(f._fixed_array(addr, chunk_bytes))
                 if k < n_chunks and e is not None]
        elif index_type == 4:                    # extensible array
            if len(self.shape) != 1:
                raise Hdf5FormatError('extensible-array chunk index with rank > 1')
            addr = _u(buf, p + 5, f._O)
            entries = [] if addr == _UNDEF else \
                [(offsets_of(k),) + e
                 for k, e in enumerate(f._extensible_array(addr, chunk_bytes, n_chunks))
                 if e is not None]
        else:
            raise Hdf5FormatError('unsupported chunk index type %d' % index_type)
        # flag bit 0: partial chunks on the boundary are stored unfiltered
        unfiltered_edges = bool(flags & 1)
        return self._assemble_chunks(entries, chunk_dims, dt, unfiltered_edges)

    def _read_contiguous(self, addr, count, dt):
        if addr == _UNDEF or count == 0:
            return np.zeros(self.shape, dtype=dt)
        f = self._file
        start = f._base + addr
        nbytes = count * dt.itemsize
        if start + nbytes > len(f._buf):
            raise Hdf5FormatError('dataset extends past end of file')
        return np.frombuffer(bytes(f._buf[start:start + nbytes]), dtype=dt,
                             count=count).reshape(self.shape).copy()

    def _read_chunked(self, btree_addr, chunk_dims, dt):
        if btree_addr == _UNDEF:
            return np.zeros(self.shape, dtype=dt)
        return self._assemble_chunks(self._file._walk_chunk_btree(btree_addr, len(self.shape)),
                                     chunk_dims, dt, False)

    def _assemble_chunks(self, entries, chunk_dims, dt, unfiltered_edges):
        f = self._file
        out = np.zeros(self.shape, dtype=dt)
        if out.size == 0:
            return out
        filters = self._filters()
        rank = len(self.shape)
        chunk_elems = int(np.prod(chunk_dims))
        for offsets, filter_mask, addr, nbytes in entries:
            start = f._base + addr
            if start + nbytes > len(f._buf):
                raise Hdf5FormatError('chunk extends past end of file')
            raw = bytes(f._buf[start:start + nbytes])
            if unfiltered_edges and any(o + c > s for o, c, s in
                                        zip(offsets, chunk_dims, self.shape)):
                filter_mask = ~0
            for i, (fid, cd) in reversed(list(enumerate(filters))):
                if filter_mask & (1 << i):
                    continue
                if fid == 1:
                    raw = zlib.decompress(raw)
                elif fid == 2:
                    esize = cd[0] if cd else dt.itemsize
                    arr = np.frombuffer(raw, dtype=np.uint8)
                    n = len(arr) // esize
                    raw = arr[:n * esize].reshape(esize, n).T.tobytes() + arr[n * esize:].tobytes()
                elif fid == 3:  # fletcher32: strip the trailing checksum
                    raw = raw[:-4]
                else:
                    raise Hdf5FormatError('unsupported filter id %d' % fid)
            need = chunk_elems * dt.itemsize
            if len(raw) < need:
                # Some writers (MinKNOW) store a short final chunk; libhdf5 zero-extends it.
                raw = raw + b'\x00' * (need - len(raw))
            chunk = np.frombuffer(raw, dtype=dt, count=chunk_elems).reshape(chunk_dims)
            sel_out, sel_chunk = [], []
            skip = False
            for d in range(rank):
                lo = offsets[d]
                hi = min(lo + chunk_dims[d], self.shape[d])
                if hi <= lo:
                    skip = True
                    break
                sel_out.append(slice(lo, hi))
                sel_chunk.append(slice(0, hi - lo))
            if not skip:
                out[tuple(sel_out)] = chunk[tuple(sel_chunk)]
        return out

    def __getitem__(self, key):
        data = self.read()
        if key is Ellipsis or key == ():
            return data
        return data[key]

    def __array__(self, dtype=None, copy=None):
        data = self.read()
        return data if dtype is None else data.astype(dtype)


class File(Group):
    """Read-only HDF5 file. Usable as a context manager like ``h5py.File(path, 'r')``."""

    def __init__(self, path, mode='r'):
        if mode != 'r':
            raise ValueError('hdf5_lite is read-only')
        self.filename = str(path)
        self._fh = None
        self._mm = None
        try:
            self._fh = open(self.filename, 'rb')
            try:
                self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
            except ValueError:
                raise Hdf5FormatError('file is empty')
            self._buf = self._mm
            self._parse_superblock()
            self._objects = {}
            super().__init__(self, self._root_addr, '/')
            self.root = self
        except Exception as e:
            self.close()
            if isinstance(e, OSError):
                raise
            raise Hdf5FormatError('not a readable HDF5 file (%s: %s)'
                                  % (type(e).__name__, e)) from e

    # -- lifecycle --------------------------------------------------------------------------
    def close(self):
        self._buf = None
        if self._mm is not None:
            try:
                self._mm.close()
            except (BufferError, ValueError):
                pass
            self._mm = None
        if self._fh is not None:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # -- superblock --------------------------------------------------------------------------
    def _parse_superblock(self):
        buf = self._buf
        off = 0
        n = len(buf)
        while True:
            if off + 8 > n:
                raise Hdf5FormatError('HDF5 signature not found')
            if buf[off:off + 8] == _SIGNATURE:
                break
            off = 512 if off == 0 else off * 2
        version = buf[off + 8]
        if version in (0, 1):
            self._O = buf[off + 13]
            self._L = buf[off + 14]
            p = off + 24 + (4 if version == 1 else 0)
            self._base = _u(buf, p, self._O)
            p += 4 * self._O
            self._root_addr = _u(buf, p + self._O, self._O)
        elif version in (2, 3):
            self._O = buf[off + 9]
            self._L = buf[off + 10]
            p = off + 12
            self._base = _u(buf, p, self._O)
            self._root_addr = _u(buf, p + 3 * self._O, self._O)
        else:
            raise Hdf5FormatError('unsupported superblock version %d' % version)
        if self._O != 8 or self._L != 8:
            raise Hdf5FormatError('only 8-byte offsets/lengths are supported')
        self._base += off if self._base == 0 and off else 0

    # -- object headers ------------------------------------------------------------------------
    def _read_object_header(self, addr):
        buf = self._buf
        start = self._base + addr
        if start + 16 > len(buf):
            raise Hdf5FormatError('object header address out of range')
        messages = []
        if buf[start:start + 4] == b'OHDR':
            version = buf[start + 4]
            if version != 2:
                raise Hdf5FormatError('unsupported object header version %d' % version)
            flags = buf[start + 5]
            p = start + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            csize = 1 << (flags & 3)
            chunk0 = _u(buf, p, csize)
            p += csize
            track_order = bool(flags & 0x04)
            blocks = [(p, chunk0)]
            while blocks:
                bp, blen = blocks.pop(0)
                end = bp + blen
                while bp + 4 <= end:
                    mtype = buf[bp]
                    msize = _u(buf, bp + 1, 2)
                    mflags = buf[bp + 3]
                    bp += 4 + (2 if track_order else 0)
                    if mtype == 0x10:
                        caddr = _u(buf, bp, self._O)
                        clen = _u(buf, bp + self._O, self._L)
                        # v2 continuation blocks start with 'OCHK' and end with a checksum.
                        blocks.append((self._base + caddr + 4, clen - 8))
                    elif mtype != 0:
                        messages.append(_Message(mtype, bp, msize, mflags))
                    bp += msize
            return messages
        version = buf[start]
        if version != 1:
            raise Hdf5FormatError('unsupported object header version %d' % version)
        nmsgs = _u(buf, start + 2, 2)
        hsize = _u(buf, start + 8, 4)
        blocks = [(start + 16, hsize)]
        seen = 0
        while blocks and seen < nmsgs:
            bp, blen = blocks.pop(0)
            end = bp + blen
            while bp + 8 <= end and seen < nmsgs:
                mtype = _u(buf, bp, 2)
                msize = _u(buf, bp + 2, 2)
                mflags = buf[bp + 4]
                bp += 8
                seen += 1
                if mtype == 0x0010:
                    caddr = _u(buf, bp, self._O)
                    clen = _u(buf, bp + self._O, self._L)
                    blocks.append((self._base + caddr, clen))
                elif mtype != 0:
                    messages.append(_Message(mtype, bp, msize, mflags))
                bp += msize
        return messages

    def _open(self, addr, name):
        key = addr
        obj = self._objects.get(key)
        if obj is not None and obj.name == name:
            return obj
        msgs = self._read_object_header(addr)
        is_dataset = any(m.type == 0x0008 for m in msgs)
        obj = (Dataset if is_dataset else Group)(self, addr, name)
        self._objects[key] = obj
        return obj

    # -- old-style groups ------------------------------------------------------------------------
    def _local_heap_data(self, addr):
        buf = self._buf
        p = self._base + addr
        if buf[p:p + 4] != b'HEAP':
            raise Hdf5FormatError('bad local heap signature')
        return self._base + _u(buf, p + 8 + 2 * self._L, self._O)

    def _cstring(self, off):
        buf = self._buf
        end = buf.find(b'\x00', off)
        return bytes(buf[off:end]).decode('utf-8')

    def _walk_group_btree(self, addr, heap_data, links):
        buf = self._buf
        p = self._base + addr
        if buf[p:p + 4] != b'TREE':
            raise Hdf5FormatError('bad group B-tree signature')
        level = buf[p + 5]
        used = _u(buf, p + 6, 2)
        q = p + 8 + 2 * self._O
        for i in range(used):
            child = _u(buf, q + self._L + i * (self._L + self._O), self._O)
            if level > 0:
                self._walk_group_btree(child, heap_data, links)
            else:
                s = self._base + child
                if buf[s:s + 4] != b'SNOD':
                    raise Hdf5FormatError('bad symbol table node signature')
                nsym = _u(buf, s + 6, 2)
                e = s + 8
                for _ in range(nsym):
                    name_off = _u(buf, e, self._O)
                    ohdr = _u(buf, e + self._O, self._O)
                    links[self._cstring(heap_data + name_off)] = ohdr
                    e += 2 * self._O + 24

    # -- new-style groups ------------------------------------------------------------------------
    def _parse_link(self, off):
        buf = self._buf
        if buf[off] != 1:
            raise Hdf5FormatError('bad link message version')
        flags = buf[off + 1]
        p = off + 2
        ltype = 0
        if flags & 0x08:
            ltype = buf[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        lsize = 1 << (flags & 3)
        nlen = _u(buf, p, lsize)
        p += lsize
        name = bytes(buf[p:p + nlen]).decode('utf-8')
        p += nlen
        if ltype != 0:
            return name, None  # soft/external links are not followed
        return name, _u(buf, p, self._O)

    class _FractalHeap:
        """Managed-object address lookup in a fractal heap (doubling table of direct blocks)."""

        def __init__(self, hfile, addr):
            self.f = hfile
            buf = hfile._buf
            O, L = hfile._O, hfile._L
            p = hfile._base + addr
            if buf[p:p + 4] != b'FRHP':
                raise Hdf5FormatError('bad fractal heap signature')
            self.id_len = _u(buf, p + 5, 2)
            if _u(buf, p + 7, 2):
                raise Hdf5FormatError('filtered fractal heaps are not supported')
            flags = buf[p + 9]
            self.max_managed = _u(buf, p + 10, 4)
            q = p + 14 + L + O + L + O + 8 * L
            self.width = _u(buf, q, 2)
            self.start_size = _u(buf, q + 2, L)
            self.max_direct = _u(buf, q + 2 + L, L)
            self.max_heap_bits = _u(buf, q + 2 + 2 * L, 2)
            self.root_addr = _u(buf, q + 6 + 2 * L, O)
            self.cur_rows = _u(buf, q + 6 + 2 * L + O, 2)
            self.off_bytes = (self.max_heap_bits + 7) // 8
            self.dblock_hdr = 5 + O + self.off_bytes + (4 if flags & 0x02 else 0)
            self.max_direct_rows = 2
            size = self.start_size
            while size < self.max_direct:
                size <<= 1
                self.max_direct_rows += 1
            # length field of a managed heap ID: enough bytes for min(max direct, max managed)
            lim = min(self.max_direct, self.max_managed)
            self.len_bytes = (max(lim, 1).bit_length() + 7) // 8

        def _row_geometry(self, row):
            s, w = self.start_size, self.width
            if row == 0:
                return 0, s
            return w * s << (row - 1), s << (row - 1)

        def locate(self, offset):
            """File offset of the byte at heap offset ``offset``."""
            f = self.f
            buf = f._buf
            O = f._O
            if self.root_addr == _UNDEF:
                raise Hdf5FormatError('empty fractal heap')
            if self.cur_rows == 0:
                return f._base + self.root_addr + offset
            iaddr, rel = self.root_addr, offset
            while True:
                b = f._base + iaddr
                if buf[b:b + 4] != b'FHIB':
                    raise Hdf5FormatError('bad fractal heap indirect block signature')
                first = self.width * self.start_size
                if rel < first:
                    row = 0
                else:
                    row = (rel // first).bit_length()
                row_start, bsize = self._row_geometry(row)
                col = (rel - row_start) // bsize
                entry = b + 5 + O + self.off_bytes + (row * self.width + col) * O
                child = _u(buf, entry, O)
                if child == _UNDEF:
                    raise Hdf5FormatError('heap object in an unallocated block')
                rel -= row_start + col * bsize
                if row < self.max_direct_rows:
                    return f._base + child + rel
                iaddr = child

        def object_offset(self, heap_id):
            kind = (heap_id[0] >> 4) & 3
            if kind != 0:
                raise Hdf5FormatError('only managed fractal-heap objects are supported')
            off = int.from_bytes(heap_id[1:1 + self.off_bytes], 'little')
            return self.locate(off)

    def _btree2_records(self, addr):
        """Yield the raw records of a version-2 B-tree in key order."""
        buf = self._buf
        O = self._O
        p = self._base + addr
        if buf[p:p + 4] != b'BTHD':
            raise Hdf5FormatError('bad v2 B-tree signature')
        node_size = _u(buf, p + 6, 4)
        rec_size = _u(buf, p + 10, 2)
        depth = _u(buf, p + 12, 2)
        root = _u(buf, p + 16, O)
        root_nrec = _u(buf, p + 16 + O, 2)
        if root == _UNDEF or root_nrec == 0:
            return

        def enc_size(limit):
            return (max(limit, 1).bit_length() - 1) // 8 + 1

        # Per-depth node capacities, as libhdf5 derives them from node and record size.
        max_nrec = [(node_size - 10) // rec_size]
        cum_max = [max_nrec[0]]
        nrec_size = enc_size(max_nrec[0])
        cum_size = [enc_size(cum_max[0])]
        for d in range(1, depth + 1):
            ptr = O + nrec_size + (cum_size[d - 1] if d > 1 else 0)
            m = (node_size - (10 + ptr)) // (rec_size + ptr)
            max_nrec.append(m)
            cum_max.append((m + 1) * cum_max[d - 1] + m)
            cum_size.append(enc_size(cum_max[d]))

        def walk(naddr, nrec, d):
            b = self._base + naddr
            if d == 0:
                if buf[b:b + 4] != b'BTLF':
                    raise Hdf5FormatError('bad v2 B-tree leaf signature')
                for i in range(nrec):
                    r = b + 6 + i * rec_size
                    yield bytes(buf[r:r + rec_size])
                return
            if buf[b:b + 4] != b'BTIN':
                raise Hdf5FormatError('bad v2 B-tree internal node signature')
            recs = b + 6
            ptrs = recs + nrec * rec_size
            ptr = O + nrec_size + (cum_size[d - 1] if d > 1 else 0)
            for i in range(nrec + 1):
                e = ptrs + i * ptr
                child = _u(buf, e, O)
                child_nrec = _u(buf, e + O, nrec_size)
                yield from walk(child, child_nrec, d - 1)
                if i < nrec:
                    r = recs + i * rec_size
                    yield bytes(buf[r:r + rec_size])

        yield from walk(root, root_nrec, depth)

    def _dense_objects(self, heap_addr, index_addr):
        """File offsets of the messages held in dense (fractal heap + v2 B-tree) storage."""
        heap = File._FractalHeap(self, heap_addr)
        for rec in self._btree2_records(index_addr):
            # type 5 (link name) record: hash(4) + heap id; type 8 (attribute name) record:
            # heap id + flags(1) + creation order(4) + hash(4).
            btype = self._buf[self._base + index_addr + 5]
            heap_id = rec[4:4 + heap.id_len] if btype == 5 else rec[:heap.id_len]
            yield heap.object_offset(heap_id)

    # -- chunk index --------------------------------------------------------------------------
    def _walk_chunk_btree(self, addr, rank):
        buf = self._buf
        p = self._base + addr
        if buf[p:p + 4] != b'TREE':
            raise Hdf5FormatError('bad chunk B-tree signature')
        if buf[p + 4] != 1:
            raise Hdf5FormatError('expected a raw-data chunk B-tree')
        level = buf[p + 5]
        used = _u(buf, p + 6, 2)
        key_size = 8 + 8 * (rank + 1)
        q = p + 8 + 2 * self._O
        for i in range(used):
            k = q + i * (key_size + self._O)
            nbytes = _u(buf, k, 4)
            mask = _u(buf, k + 4, 4)
            offsets = [_u(buf, k + 8 + 8 * d, 8) for d in range(rank)]
            child = _u(buf, k + key_size, self._O)
            if level > 0:
                yield from self._walk_chunk_btree(child, rank)
            else:
                yield offsets, mask, child, nbytes

    # -- chunk indexes of layout version 4 ------------------------------------------------------
    def _index_element(self, p, filtered, elmt_size, chunk_bytes):
        """One chunk record of a fixed / extensible array: (filter mask, address, bytes) or None
        for a chunk that was never written."""
        buf = self._buf
        addr = _u(buf, p, self._O)
        if addr == _UNDEF:
            return None
        if not filtered:
            return (0, addr, chunk_bytes)
        size_len = elmt_size - self._O - 4
        return (_u(buf, p + self._O + size_len, 4), addr, _u(buf, p + self._O, size_len))

    def _fixed_array(self, addr, chunk_bytes):
        """Records of a fixed array (FAHD -> FADB, optionally paged), in index order."""
        buf = self._buf
        p = self._base + addr
        if buf[p:p + 4] != b'FAHD' or buf[p + 4] != 0:
            raise Hdf5FormatError('bad fixed array header')
        filtered = buf[p + 5] == 1
        elmt_size = buf[p + 6]
        page_bits = buf[p + 7]
        n = _u(buf, p + 8, self._L)
        block = self._base + _u(buf, p + 8 + self._L, self._O)
        if buf[block:block + 4] != b'FADB':
            raise Hdf5FormatError('bad fixed array data block')
        q = block + 6 + self._O
        page = 1 << page_bits
        if n > page:                             # paged: bitmap, checksum, then checksummed pages
            n_pages = -(-n // page)
            bitmap = bytes(buf[q:q + (n_pages + 7) // 8])
            q += len(bitmap) + 4
            for k in range(n):
                pg, within = divmod(k, page)
                if not bitmap[pg // 8] & (0x80 >> (pg % 8)):
                    yield None
                    continue
                at = q + pg * (page * elmt_size + 4) + within * elmt_size
                yield self._index_element(at, filtered, elmt_size, chunk_bytes)
        else:
            for k in range(n):
                yield self._index_element(q + k * elmt_size, filtered, elmt_size, chunk_bytes)

    def _extensible_array(self, addr, chunk_bytes, n_chunks):
        """Records 0 .. n_chunks-1 of an extensible array (EAHD -> EAIB -> EADB / EASB): the first
        few live in the index block, then data blocks of doubling size, the early ones addressed
        from the index block, the later ones through super blocks; big data blocks are paged."""
        buf = self._buf
        O, L = self._O, self._L
        p = self._base + addr
        if buf[p:p + 4] != b'EAHD' or buf[p + 4] != 0:
            raise Hdf5FormatError('bad extensible array header')
        filtered = buf[p + 5] == 1
        elmt_size, max_bits, idx_elmts, dblk_min, sblk_min_ptrs, page_bits = buf[p + 6:p + 12]
        index_block = _u(buf, p + 12 + 6 * L, O)
        if index_block == _UNDEF:
            for _ in range(n_chunks):
                yield None
            return
        off_size = (max_bits + 7) // 8
        log2 = lambda v: v.bit_length() - 1                              # noqa: E731
        n_sblks = 1 + max_bits - log2(dblk_min)
        sblk = []                                 # (data blocks, elements per data block, first element, first data block)
        start_idx = start_dblk = 0
        for u in range(n_sblks):
            n_dblks, dblk_elmts = 1 << (u // 2), (1 << ((u + 1) // 2)) * dblk_min
            sblk.append((n_dblks, dblk_elmts, start_idx, start_dblk))
            start_idx += n_dblks * dblk_elmts
            start_dblk += n_dblks
        iblock_sblks = 2 * log2(sblk_min_ptrs)
        n_dblk_addrs = 2 * (sblk_min_ptrs - 1)
        q = self._base + index_block
        if buf[q:q + 4] != b'EAIB':
            raise Hdf5FormatError('bad extensible array index block')
        elements = q + 6 + O
        dblk_addrs = elements + idx_elmts * elmt_size
        sblk_addrs = dblk_addrs + n_dblk_addrs * O
        page = 1 << page_bits

        def in_data_block(block_addr, within, dblk_elmts):
            if block_addr == _UNDEF:
                return None
            b = self._base + block_addr
            if buf[b:b + 4] != b'EADB':
                raise Hdf5FormatError('bad extensible array data block')
            e = b + 6 + O + off_size
            if dblk_elmts > page:                 # prefix checksum, then checksummed pages
                pg, within = divmod(within, page)
                e += 4 + pg * (page * elmt_size + 4)
            return self._index_element(e + within * elmt_size, filtered, elmt_size, chunk_bytes)

        for k in range(n_chunks):
            if k < idx_elmts:
                yield self._index_element(elements + k * elmt_size, filtered, elmt_size,
                                          chunk_bytes)
                continue
            rel = k - idx_elmts
            s_idx = log2(rel // dblk_min + 1)
            n_dblks, dblk_elmts, first_elmt, first_dblk = sblk[s_idx]
            rel -= first_elmt
            d_idx, within = divmod(rel, dblk_elmts)
            if s_idx < iblock_sblks:
                block = _u(buf, dblk_addrs + (first_dblk + d_idx) * O, O)
                yield in_data_block(block, within, dblk_elmts)
                continue
            super_addr = _u(buf, sblk_addrs + (s_idx - iblock_sblks) * O, O)
            if super_addr == _UNDEF:
                yield None
                continue
            sb = self._base + super_addr
            if buf[sb:sb + 4] != b'EASB':
                raise Hdf5FormatError('bad extensible array super block')
            t = sb + 6 + O + off_size
            if dblk_elmts > page:
                # "page initialised" bits: room for a whole number of bytes per data block, but
                # used as one run of bits, `pages` per data block (libhdf5 1.10.6 writes it so)
                pages = dblk_elmts // page
                bit = d_idx * pages + within // page
                if not buf[t + bit // 8] & (0x80 >> (bit % 8)):
                    yield None
                    continue
                t += n_dblks * ((pages + 7) // 8)
            yield in_data_block(_u(buf, t + d_idx * O, O), within, dblk_elmts)

    # -- attributes --------------------------------------------------------------------------
    def _global_heap_object(self, coll_addr, index):
        buf = self._buf
        p = self._base + coll_addr
        if buf[p:p + 4] != b'GCOL':
            raise Hdf5FormatError('bad global heap signature')
        size = _u(buf, p + 8, self._L)
        o = p + 8 + self._L
        end = p + size
        while o + 8 + self._L <= end:
            idx = _u(buf, o, 2)
            osize = _u(buf, o + 8, self._L)
            if idx == 0:
                break
            if idx == index:
                return bytes(buf[o + 8 + self._L:o + 8 + self._L + osize])
            o += 8 + self._L + _pad8(osize)
        raise Hdf5FormatError('global heap object %d not found' % index)

    def _parse_attribute(self, off):
        buf = self._buf
        version = buf[off]
        nsz, tsz, ssz = _u(buf, off + 2, 2), _u(buf, off + 4, 2), _u(buf, off + 6, 2)
        p = off + 8
        if version == 3:
            p += 1
        elif version not in (1, 2):
            raise Hdf5FormatError('unsupported attribute version %d' % version)
        pad = _pad8 if version == 1 else (lambda n: n)
        name = bytes(buf[p:p + nsz]).split(b'\x00')[0].decode('utf-8')
        p += pad(nsz)
        dt_off = p
        p += pad(tsz)
        dims, _ = _parse_dataspace(buf, p, self._L)
        p += pad(ssz)
        try:
            dt = _Datatype(buf, dt_off)
        except Hdf5FormatError:
            return name, None
        count = int(np.prod(dims)) if dims else 1
        if dt.vlen_string:
            vals = []
            for i in range(count):
                e = p + 16 * i
                coll = _u(buf, e + 4, self._O)
                idx = _u(buf, e + 4 + self._O, 4)
                vals.append(self._global_heap_object(coll, idx) if coll else b'')
            if not dims:
                return name, vals[0]
            return name, np.array(vals, dtype=object).reshape(dims)
        raw = bytes(buf[p:p + count * dt.size])
        arr = np.frombuffer(raw, dtype=dt.numpy, count=count)
        if dt.string:
            # h5py hands fixed-length strings back as bytes with the NUL padding stripped.
            if not dims:
                return name, bytes(arr[0]).split(b'\x00')[0]
            return name, np.array([bytes(x).split(b'\x00')[0] for x in arr]).reshape(dims)
        if not dims:
            return name, arr[0]
        return name, arr.reshape(dims).copy()


def is_hdf5(path):
    """True if ``path`` starts with (or contains at a legal offset) the HDF5 signature."""
    try:
        with open(path, 'rb') as f:
            off = 0
            while True:
                f.seek(off)
                sig = f.read(8)
                if len(sig) < 8:
                    return False
                if sig == _SIGNATURE:
                    return True
                off = 512 if off == 0 else off * 2
    except OSError:
        return False

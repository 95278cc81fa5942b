"""
ctypes binding of ``libdeepbinner_fast5.so`` (C ABI: ``include/deepbinner_fast5.h``), the native
fast5 loader: read id + raw int16 signal of single- and multi-read fast5 files, and a
multi-threaded batch loader that hands back packed scan regions.

Same idiom as the reference's one native library (``deepbinner/dtw_semi_global.py:30-41``) and as
``hip_backend``.  The pure-Python reader (``hdf5_lite``) implements the same slice of the HDF5
format and is what this library is tested against (``tests/test_fast5_native.py``).
"""

import ctypes
import os
import weakref

import numpy as np

_LIB_NAME = 'libdeepbinner_fast5.so'
_lib = None

(F5_OK, F5_ERR_OPEN, F5_ERR_FORMAT, F5_ERR_NO_READ, F5_ERR_MULTI, F5_ERR_ARGUMENT,
 F5_ERR_FILTER, F5_ERR_EXISTS) = range(8)
F5_READ_ID_MAX = 64
LAYOUT_NONE, LAYOUT_SINGLE_OLD, LAYOUT_SINGLE_NEW, LAYOUT_MULTI = range(4)

EXPORTED_SYMBOLS = [
    'f5_version', 'f5_status_string', 'f5_usable_cpus', 'f5_open', 'f5_close', 'f5_layout', 'f5_read_info',
    'f5_read_signal', 'f5_load_batch', 'f5_load_reads', 'f5_batch_size', 'f5_batch_samples', 'f5_batch_offsets', 'f5_batch_status',
    'f5_batch_read_ids', 'f5_batch_free', 'f5_stream_open', 'f5_stream_next', 'f5_stream_close',
    'f5_set_sample_allocator', 'f5_release_idle_buffers', 'f5_stream_open_raw', 'f5_batch_comp',
    'f5_batch_comp_bytes', 'f5_batch_streams', 'f5_batch_n_streams', 'f5_write_single_reads',
    'f5_single_read_image', 'f5_load_batch_raw',
]


class Fast5NativeError(OSError):
    """The library is missing or refused a file (an OSError, like h5py's and hdf5_lite's)."""


def library_path():
    """The in-tree build; DEEPBINNER_FAST5_LIB points somewhere else (a sanitizer build, say)."""
    return os.environ.get('DEEPBINNER_FAST5_LIB') or os.path.join(
        os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def available():
    return os.path.isfile(library_path())


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.isfile(path):
        raise Fast5NativeError('{} not found - build it with `make -C deepbinner_amd/csrc` '
                               '(python -c "import __graft_entry__ as g; g.build()")'.format(path))
    lib = ctypes.cdll.LoadLibrary(path)
    c_int, c_i64, c_void_p, c_char_p = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_char_p
    P = ctypes.POINTER
    sigs = {
        'f5_version': (c_char_p, []),
        'f5_status_string': (c_char_p, [c_int]),
        'f5_usable_cpus': (c_int, []),
        'f5_open': (c_int, [c_char_p, P(c_void_p)]),
        'f5_close': (None, [c_void_p]),
        'f5_layout': (c_int, [c_void_p, P(c_int), P(c_i64)]),
        'f5_read_info': (c_int, [c_void_p, c_i64, ctypes.c_char * F5_READ_ID_MAX, P(c_i64)]),
        'f5_read_signal': (c_int, [c_void_p, c_i64, c_i64, c_i64,
                                   np.ctypeslib.ndpointer(np.int16, flags='C_CONTIGUOUS')]),
        'f5_load_batch': (c_int, [P(c_char_p), c_i64, c_i64, c_int, P(c_void_p)]),
        'f5_load_reads': (c_int, [c_char_p, c_i64, c_i64, c_i64, c_int, P(c_void_p)]),
        'f5_batch_size': (ctypes.c_int64, [c_void_p]),
        'f5_batch_samples': (P(ctypes.c_int16), [c_void_p]),
        'f5_batch_offsets': (P(c_i64), [c_void_p]),
        'f5_batch_status': (P(ctypes.c_int32), [c_void_p]),
        'f5_batch_read_ids': (P(ctypes.c_char), [c_void_p]),
        'f5_batch_free': (None, [c_void_p]),
        'f5_stream_open': (c_int, [P(c_char_p), c_i64, c_i64, c_int, c_int, P(c_void_p)]),
        'f5_stream_next': (c_int, [c_void_p, P(c_i64), P(c_int), P(c_void_p)]),
        'f5_stream_close': (None, [c_void_p]),
        'f5_stream_open_raw': (c_int, [P(c_char_p), c_i64, c_int, c_int, c_i64, P(c_void_p)]),
        'f5_load_batch_raw': (c_int, [P(c_char_p), c_i64, c_int, c_i64, P(c_void_p)]),
        'f5_write_single_reads': (c_int, [c_char_p, c_i64, P(c_i64), P(c_char_p), c_int,
                                          P(ctypes.c_int32), P(c_i64)]),
        'f5_single_read_image': (c_int, [c_char_p, c_i64, c_void_p, c_i64, P(c_i64)]),
        'f5_batch_comp': (P(ctypes.c_uint8), [c_void_p]),
        'f5_batch_comp_bytes': (c_i64, [c_void_p]),
        'f5_batch_streams': (c_void_p, [c_void_p]),
        'f5_batch_n_streams': (c_i64, [c_void_p]),
        'f5_set_sample_allocator': (c_int, [c_void_p, c_void_p, c_void_p]),
        'f5_release_idle_buffers': (None, []),
    }
    for name, (restype, argtypes) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def usable_cpus():
    """Hardware threads this process can keep busy (online CPUs, affinity mask, cgroup quota)."""
    return int(load_library().f5_usable_cpus())


def status_string(status):
    return load_library().f5_status_string(int(status)).decode()


class File:
    """One open fast5 file: ``layout``, ``n_reads``, ``read_info(i)``, ``read_signal(i, ...)``."""

    def __init__(self, path):
        self._lib = load_library()
        self._handle = ctypes.c_void_p()
        status = self._lib.f5_open(os.fsencode(str(path)), ctypes.byref(self._handle))
        if status != F5_OK:
            self._handle = None
            raise Fast5NativeError('{}: {}'.format(path, status_string(status)))
        layout, n = ctypes.c_int(0), ctypes.c_int64(0)
        self._lib.f5_layout(self._handle, ctypes.byref(layout), ctypes.byref(n))
        self.layout, self.n_reads = layout.value, n.value

    def close(self):
        if self._handle:
            self._lib.f5_close(self._handle)
            self._handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def read_info(self, index):
        """-> (read_id str, n_samples); KeyError when the read lacks read_id / Signal."""
        rid = (ctypes.c_char * F5_READ_ID_MAX)()
        n = ctypes.c_int64(0)
        status = self._lib.f5_read_info(self._handle, int(index), rid, ctypes.byref(n))
        if status == F5_ERR_NO_READ:
            raise KeyError('read {} (read_id / Signal)'.format(index))
        if status != F5_OK:
            raise Fast5NativeError(status_string(status))
        try:
            return rid.value.decode(), n.value
        except UnicodeDecodeError:
            raise KeyError('read {}: read_id is not text'.format(index)) from None

    def read_signal(self, index, first=0, count=None):
        _, n = self.read_info(index)
        count = n - first if count is None else count
        out = np.empty(max(count, 0), dtype=np.int16)
        status = self._lib.f5_read_signal(self._handle, int(index), int(first), int(count), out)
        if status == F5_ERR_NO_READ:
            raise KeyError('sample range {}+{} of read {}'.format(first, count, index))
        if status != F5_OK:
            raise Fast5NativeError(status_string(status))
        return out


def get_read_id_and_signal(fast5_file):
    """Native equivalent of ``load_fast5s.get_read_id_and_signal`` (reference
    load_fast5s.py:25-49): (read_id, int16 signal), (None, None) for unreadable files, SystemExit
    for a multi-read file."""
    import sys
    try:
        with File(fast5_file) as f:
            if f.layout == LAYOUT_MULTI:
                sys.exit('Error: Deepbinner does not (yet) support multi-read fast5 files')
            if f.layout == LAYOUT_NONE:
                return None, None
            read_id, _ = f.read_info(0)
            return read_id, f.read_signal(0)
    except (OSError, KeyError):
        return None, None


def iter_reads(fast5_file):
    """(read_id, signal) for every read of a single- or multi-read fast5."""
    try:
        with File(fast5_file) as f:
            for i in range(f.n_reads):
                yield f.read_info(i)[0], f.read_signal(i)
    except (OSError, KeyError):
        return


def _unpack_batch(lib, handle, n, keep_alive=None):
    """(read_ids, samples, offsets, status) from a native batch handle (which it frees, at once or
    when the zero-copy sample array - or ``keep_alive``, the zero-copy byte array of a raw batch -
    dies)."""
    try:
        offsets = np.ctypeslib.as_array(lib.f5_batch_offsets(handle), shape=(n + 1,)).copy()
        total = int(offsets[n])
        if n:
            st = np.ctypeslib.as_array(lib.f5_batch_status(handle), shape=(n,)).copy()
            raw = ctypes.string_at(lib.f5_batch_read_ids(handle), n * F5_READ_ID_MAX)
        else:
            st, raw = np.empty(0, dtype=np.int32), b''
    except Exception:
        lib.f5_batch_free(handle)
        raise
    if keep_alive is not None:
        samples = None
        weakref.finalize(keep_alive, lib.f5_batch_free, handle)
    elif total:
        # no copy: the array (and every slice of it) keeps the native batch alive
        samples = np.ctypeslib.as_array(lib.f5_batch_samples(handle), shape=(total,))
        weakref.finalize(samples, lib.f5_batch_free, handle)
    else:
        samples = np.empty(0, dtype=np.int16)
        lib.f5_batch_free(handle)
    # the slots are NUL padded: numpy's fixed-width bytes type strips that in C
    slots = np.frombuffer(raw, dtype='S%d' % F5_READ_ID_MAX).tolist() if n else []
    readable = (st == F5_OK).tolist()
    try:
        read_ids = [slot.decode() if ok else None for slot, ok in zip(slots, readable)]
    except UnicodeDecodeError:              # a damaged id somewhere: one by one
        read_ids = []
        for i, (slot, ok) in enumerate(zip(slots, readable)):
            try:
                read_ids.append(slot.decode() if ok else None)
            except UnicodeDecodeError:      # treat the read as unreadable
                read_ids.append(None)
                st[i] = F5_ERR_FORMAT
    return read_ids, samples, offsets, st


def _unpack_raw_batch(lib, batch):
    """(read_ids, offsets, status, comp, records) of a raw batch handle (freed when ``comp``, the
    zero-copy array of its bytes, dies)."""
    count = int(lib.f5_batch_size(batch))
    n_streams = int(lib.f5_batch_n_streams(batch))
    comp_bytes = int(lib.f5_batch_comp_bytes(batch))
    try:
        records = np.empty(n_streams, dtype=RAW_STREAM)
        if n_streams:
            ctypes.memmove(records.ctypes.data, lib.f5_batch_streams(batch),
                           n_streams * RAW_STREAM.itemsize)
        # (64 readable bytes behind the streams: part of the array, so that they travel)
        comp = np.ctypeslib.as_array(lib.f5_batch_comp(batch), shape=(comp_bytes + 64,)) \
            if n_streams else np.zeros(64, dtype=np.uint8)
    except Exception:
        lib.f5_batch_free(batch)
        raise
    ids, _, offsets, st = _unpack_batch(lib, batch, count, keep_alive=comp if n_streams else None)
    return ids, offsets, st, comp, records


def load_batch_raw(fast5_files, threads=0, host_inflate_above=0):
    """One-read files with their Signals AS STORED -> (read_ids, offsets, status, comp, records),
    laid out like a batch of ``stream_raw`` (read i = file i), for
    ``hip_backend.classify_pair_deflated``.  ``host_inflate_above`` as there: > 0 bytes, or minus
    the per cent of the batch's compressed bytes (its longest streams) the host's threads inflate
    themselves."""
    lib = load_library()
    n = len(fast5_files)
    paths = (ctypes.c_char_p * max(n, 1))(*[os.fsencode(str(p)) for p in fast5_files])
    handle = ctypes.c_void_p()
    status = lib.f5_load_batch_raw(paths, n, int(threads), int(host_inflate_above),
                                   ctypes.byref(handle))
    if status != F5_OK:
        raise Fast5NativeError(status_string(status))
    return _unpack_raw_batch(lib, handle)


def load_batch(fast5_files, keep=None, threads=0):
    """One-read files -> (read_ids, samples, offsets, status): read i is
    ``samples[offsets[i]:offsets[i+1]]`` (its first and last ``keep`` samples only when it is
    longer than 2*keep), ``read_ids[i]`` is None and ``status[i]`` != 0 for a file that could not
    be read.  The files are parsed and inflated by ``threads`` native threads (0 = one per
    hardware thread, at most 64); the GIL is released meanwhile."""
    lib = load_library()
    n = len(fast5_files)
    paths = (ctypes.c_char_p * max(n, 1))(*[os.fsencode(str(p)) for p in fast5_files])
    handle = ctypes.c_void_p()
    status = lib.f5_load_batch(paths, n, int(keep or 0), int(threads), ctypes.byref(handle))
    if status != F5_OK:
        raise Fast5NativeError(status_string(status))
    return _unpack_batch(lib, handle, n)


def load_reads(fast5_file, first=0, count=None, keep=None, threads=0):
    """Reads [first, first + count) of one (multi-read) fast5 file, loaded by native threads:
    (read_ids, samples, offsets, status) as from load_batch.  count None = to the end."""
    lib = load_library()
    handle = ctypes.c_void_p()
    status = lib.f5_load_reads(os.fsencode(str(fast5_file)), int(first),
                               -1 if count is None else int(count), int(keep or 0),
                               int(threads), ctypes.byref(handle))
    if status != F5_OK:
        raise Fast5NativeError('{}: {}'.format(fast5_file, status_string(status)))
    return _unpack_batch(lib, handle, int(lib.f5_batch_size(handle)))


def stream_reads(fast5_files, keep=None, threads=0, depth=0):
    """Multi-read containers as a stream (``f5_stream_*``): yields, in the order of
    ``fast5_files``, ``(index, read_ids, samples, offsets, status)`` per container as
    ``load_reads`` would return them - or ``(index, None, None, None, container_status)`` for a
    file that could not be opened - while a team of ``threads`` native threads works ``depth``
    containers ahead (opening and walking the next containers beside the inflating of the current
    one).  Closing the generator stops the team."""
    lib = load_library()
    n = len(fast5_files)
    paths = (ctypes.c_char_p * max(n, 1))(*[os.fsencode(str(p)) for p in fast5_files])
    stream = ctypes.c_void_p()
    status = lib.f5_stream_open(paths, n, int(keep or 0), int(threads), int(depth),
                                ctypes.byref(stream))
    if status != F5_OK:
        raise Fast5NativeError(status_string(status))
    try:
        index, container_status = ctypes.c_int64(0), ctypes.c_int(0)
        handle = ctypes.c_void_p()
        while lib.f5_stream_next(stream, ctypes.byref(index), ctypes.byref(container_status),
                                 ctypes.byref(handle)) == F5_OK:
            if container_status.value != F5_OK:
                yield index.value, None, None, None, container_status.value
                continue
            batch = ctypes.c_void_p(handle.value)
            yield (index.value,) + _unpack_batch(lib, batch, int(lib.f5_batch_size(batch)))
    finally:
        lib.f5_stream_close(stream)


RAW_ZLIB, RAW_STORED = 0, 1
# f5_raw_stream (include/deepbinner_fast5.h) = dbh_inflate_stream (include/deepbinner_hip.h)
RAW_STREAM = np.dtype([('comp_offset', '<i8'), ('comp_bytes', '<i8'), ('out_offset', '<i8'),
                       ('out_bytes', '<i8'), ('mode', '<i4'), ('read', '<i4')])


def stream_raw(fast5_files, threads=0, depth=0, host_inflate_above=0):
    """Multi-read containers as a stream of RAW batches (``f5_stream_open_raw``): the Signal of
    every read as it is stored - zlib streams, mostly - for a decoder elsewhere (the GPU:
    ``hip_backend.classify_pair_deflated``).  Yields, in the order of ``fast5_files``,
    ``(index, read_ids, offsets, status, comp, streams)``: ``offsets`` (samples) say where each
    read's signal lies once decoded, ``comp`` is the byte buffer (uint8, zero-copy: it keeps the
    native batch alive), ``streams`` an array of RAW_STREAM records - or ``(index, None, None,
    container_status, None, None)`` for a file that could not be opened."""
    lib = load_library()
    n = len(fast5_files)
    paths = (ctypes.c_char_p * max(n, 1))(*[os.fsencode(str(p)) for p in fast5_files])
    stream = ctypes.c_void_p()
    status = lib.f5_stream_open_raw(paths, n, int(threads), int(depth), int(host_inflate_above),
                                    ctypes.byref(stream))
    if status != F5_OK:
        raise Fast5NativeError(status_string(status))
    try:
        index, container_status = ctypes.c_int64(0), ctypes.c_int(0)
        handle = ctypes.c_void_p()
        while lib.f5_stream_next(stream, ctypes.byref(index), ctypes.byref(container_status),
                                 ctypes.byref(handle)) == F5_OK:
            if container_status.value != F5_OK:
                yield index.value, None, None, container_status.value, None, None
                continue
            ids, offsets, st, comp, records = _unpack_raw_batch(lib, ctypes.c_void_p(handle.value))
            yield index.value, ids, offsets, st, comp, records
    finally:
        lib.f5_stream_close(stream)


def write_single_reads(container, read_indices, out_paths, threads=0):
    """Reads ``read_indices`` of the multi-read container ``container`` as one-read fast5 files
    ``out_paths`` (signal as stored + the read's Raw / channel_id / tracking_id / context_tags
    attributes: include/deepbinner_fast5.h) on the library's worker threads -> (status per read,
    bytes written).  A path that exists already is left alone (status F5_ERR_EXISTS).  Raises Fast5NativeError if the container cannot be opened."""
    lib = load_library()
    n = len(out_paths)
    if len(read_indices) != n:
        raise ValueError('one path per read')
    indices = (ctypes.c_int64 * max(n, 1))(*[int(i) for i in read_indices])
    paths = (ctypes.c_char_p * max(n, 1))(*[os.fsencode(p) for p in out_paths])
    status = (ctypes.c_int32 * max(n, 1))()
    written = ctypes.c_int64(0)
    rc = lib.f5_write_single_reads(os.fsencode(container), n, indices, paths, int(threads), status,
                                   ctypes.byref(written))
    if rc != F5_OK:
        raise Fast5NativeError('{}: {}'.format(container, status_string(rc)))
    return np.frombuffer(status, dtype=np.int32, count=n).copy(), written.value


def single_read_image(container, read_index):
    """The bytes ``write_single_reads`` writes for one read (tests)."""
    lib = load_library()
    size = ctypes.c_int64(0)
    rc = lib.f5_single_read_image(os.fsencode(container), int(read_index), None, 0,
                                  ctypes.byref(size))
    if rc != F5_OK:
        raise Fast5NativeError('{}: {}'.format(container, status_string(rc)))
    buf = ctypes.create_string_buffer(size.value)
    rc = lib.f5_single_read_image(os.fsencode(container), int(read_index), buf, size.value,
                                  ctypes.byref(size))
    if rc != F5_OK:
        raise Fast5NativeError('{}: {}'.format(container, status_string(rc)))
    return buf.raw[:size.value]


def set_sample_allocator(alloc_address=None, release_address=None, user=None):
    """Where batches keep their packed samples: two C function pointers (as integers)
    ``void* alloc(size_t, void*)`` / ``void release(void*, void*)`` - ``hip_backend.
    use_pinned_loader_buffers()`` passes the pinned-host-memory pair of libdeepbinner_hip.so - or
    nothing for malloc."""
    status = load_library().f5_set_sample_allocator(alloc_address, release_address, user)
    if status != F5_OK:
        raise Fast5NativeError(status_string(status))


def release_idle_buffers():
    """Free the sample buffers waiting in the library's pool."""
    load_library().f5_release_idle_buffers()

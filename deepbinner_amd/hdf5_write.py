"""
A minimal HDF5 *writer*: just enough of the format to emit one-read fast5 files, so that
``deepbinner realtime`` can bin the reads of multi-read containers into ``barcodeNN/`` /
``unclassified/`` directories without shelling out to ``multi_to_single_fast5`` (which the
reference does, realtime.py:183-190, and which this image does not have).

What is written is the classic on-disk layout every HDF5 library reads (the one libhdf5 writes
with ``libver='earliest'``): superblock version 0, version-1 object headers, groups as symbol
tables (B-tree v1 node + local heap + symbol-table node), one 1-D ``<i2`` dataset stored as a
single deflate-compressed chunk (B-tree v1 chunk index) or contiguously, and fixed-length string
attributes.  The tree of a file is the "new" single-read layout the reference's loader accepts
(load_fast5s.py:25-49)::

    /read_<read_id>/Raw            attrs: read_id (+ read_number, duration if given)
    /read_<read_id>/Raw/Signal     int16[n]

Pinned by reading the files back with the real HDF5 library (h5py, where the image has it), with
the reference's own ``get_read_id_and_signal`` and with both of this package's readers
(tests/test_hdf5_write.py).
"""

import errno
import os
import struct
import threading
import zlib

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b'\x89HDF\r\n\x1a\n'
GROUP_LEAF_K, GROUP_INTERNAL_K, CHUNK_K = 4, 16, 32      # superblock v0 defaults


def _pad8(data):
    return data + b'\0' * (-len(data) % 8)


class _Image:
    """The file as one growing byte string; every block starts 8-byte aligned."""

    def __init__(self):
        self.buf = bytearray()

    def reserve(self, size):
        self.buf += b'\0' * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += b'\0' * size
        return addr

    def put(self, addr, data):
        self.buf[addr:addr + len(data)] = data

    def add(self, data):
        addr = self.reserve(len(data))
        self.put(addr, data)
        return addr


def _message(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack('<HHB3x', mtype, len(body), flags) + body


def _object_header(messages):
    data = b''.join(messages)
    # version 1, reserved, message count, reference count 1, size of the message data, then the
    # messages on an 8-byte boundary
    return struct.pack('<BBHII4x', 1, 0, len(messages), 1, len(data)) + data


def _string_datatype(size):
    # class 3 (string), version 1; null-terminated, ASCII
    return struct.pack('<B3BI', 0x13, 0, 0, 0, size)


def _fixed_datatype(size, signed):
    # class 0 (fixed point), version 1; little-endian, two's complement
    return struct.pack('<B3BIHH', 0x10, 0x08 if signed else 0, 0, 0, size, 0, 8 * size)


def _float_datatype(size=8):
    # class 1 (floating point), version 1; IEEE little-endian binary64 / binary32
    if size == 4:
        return struct.pack('<B3BIHHBBBBI', 0x11, 0x20, 0x1F, 0, 4, 0, 32, 23, 8, 0, 23, 127)
    return struct.pack('<B3BIHHBBBBI', 0x11, 0x20, 0x3F, 0, 8, 0, 64, 52, 11, 0, 52, 1023)


def _scalar_dataspace():
    return struct.pack('<BBB5x', 1, 0, 0)


def _simple_dataspace(n):
    return struct.pack('<BBB5xQ', 1, 1, 0, n)


def _attribute(name, datatype, value_bytes):
    name_z = name.encode() + b'\0'
    space = _scalar_dataspace()
    return _message(0x000C, struct.pack('<BBHHH', 1, 0, len(name_z), len(datatype), len(space)) +
                    _pad8(name_z) + _pad8(datatype) + _pad8(space) + value_bytes)


def string_attribute(name, text):
    raw = text.encode() + b'\0'
    return _attribute(name, _string_datatype(len(raw)), raw)


def int_attribute(name, value, size=4, signed=True):
    fmt = {(1, True): '<b', (1, False): '<B', (2, True): '<h', (2, False): '<H', (4, True): '<i',
           (4, False): '<I', (8, True): '<q', (8, False): '<Q'}[(size, signed)]
    return _attribute(name, _fixed_datatype(size, signed), struct.pack(fmt, value))


def float_attribute(name, value, size=8):
    return _attribute(name, _float_datatype(size), struct.pack('<d' if size == 8 else '<f', value))


def attribute_from_value(name, value):
    """An attribute message for a scalar as this package's reader (or h5py) returns it: bytes /
    str -> fixed-length string, NumPy or Python integers and floats -> the same width and
    signedness.  None for what a one-read file is not given (arrays, enums, references)."""
    if isinstance(value, (bytes, str)):
        return string_attribute(name, value.decode('utf-8', 'replace')
                                if isinstance(value, bytes) else value)
    if isinstance(value, (bool, np.bool_)):
        return int_attribute(name, int(value), 1, False)
    if isinstance(value, np.integer):
        return int_attribute(name, int(value), value.dtype.itemsize, value.dtype.kind == 'i')
    if isinstance(value, int):
        return int_attribute(name, value, 8, True)
    if isinstance(value, np.floating) and value.dtype.itemsize in (4, 8):
        return float_attribute(name, float(value), value.dtype.itemsize)
    if isinstance(value, float):
        return float_attribute(name, value)
    return None


def _attributes(values, skip=()):
    found = []
    for name in sorted(values):
        if name in skip:
            continue
        message = attribute_from_value(name, values[name])
        if message is not None:
            found.append(message)
    return found


def _group(image, children, attributes=()):
    """A group with the given {name: (object header address, btree, heap)} children (btree / heap
    = the child's own symbol table for groups, None for datasets) -> (header addr, btree, heap).
    All links sit in ONE symbol-table node, which holds up to 2 x the file's "group leaf node K"
    of them (``image.leaf_k``, recorded in the superblock: 4 by default, more for the root group
    of a multi-read container)."""
    names = sorted(children, key=lambda name: name.encode())
    leaf_k = getattr(image, 'leaf_k', GROUP_LEAF_K)
    assert 1 <= len(names) <= 2 * leaf_k, 'one symbol-table node holds up to 2 x leaf K links'
    # local heap data: offset 0 is the empty string, then the names, each 8-byte aligned
    heap_data, offsets = bytearray(8), {}
    for name in names:
        offsets[name] = len(heap_data)
        heap_data += _pad8(name.encode() + b'\0')
    # libhdf5 wants a free block description where there is free space: leave none
    heap_data_addr = image.add(bytes(heap_data))
    heap = image.add(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap_data), 1, heap_data_addr))
    # symbol-table node: 2K entries of 40 bytes, the first len(names) in use
    entries = b''
    for name in names:
        header, child_btree, child_heap = children[name]
        if child_btree is None:
            entries += struct.pack('<QQII16x', offsets[name], header, 0, 0)
        else:
            entries += struct.pack('<QQIIQQ', offsets[name], header, 1, 0, child_btree, child_heap)
    snod = image.add(b'SNOD' + struct.pack('<BBH', 1, 0, len(names)) + entries +
                     b'\0' * (40 * (2 * leaf_k - len(names))))
    # B-tree v1, group node (type 0), leaf level, one child: keys are heap offsets of names
    node = b'TREE' + struct.pack('<BBHQQ', 0, 0, 1, UNDEF, UNDEF)
    node += struct.pack('<QQQ', 0, snod, offsets[names[-1]])
    btree = image.add(node + b'\0' * (24 + 16 * GROUP_INTERNAL_K * 2 + 8 - len(node)))
    header = image.add(_object_header([_message(0x0011, struct.pack('<QQ', btree, heap))] +
                                      list(attributes)))
    return header, btree, heap


def _attribute_group(image, attributes):
    """A group without links that only carries attributes (channel_id, tracking_id, ...): an empty
    symbol table (B-tree node with no entries, heap with the empty name)."""
    heap_data_addr = image.add(bytes(8))
    heap = image.add(b'HEAP' + struct.pack('<B3xQQQ', 0, 8, 1, heap_data_addr))
    node = b'TREE' + struct.pack('<BBHQQ', 0, 0, 0, UNDEF, UNDEF)
    btree = image.add(node + b'\0' * (24 + 16 * GROUP_INTERNAL_K * 2 + 8 - len(node)))
    header = image.add(_object_header([_message(0x0011, struct.pack('<QQ', btree, heap))] +
                                      list(attributes)))
    return header, btree, heap


def _dataset_int16(image, samples, compress, packed=None):
    samples = np.ascontiguousarray(samples, dtype='<i2')
    n = len(samples)
    messages = [_message(0x0001, _simple_dataspace(n)),
                _message(0x0003, _fixed_datatype(2, True), flags=1),       # constant message
                # fill value, version 2: allocate late / write never / undefined
                _message(0x0005, struct.pack('<BBBB', 2, 2, 0, 0))]
    raw = samples.tobytes()
    if compress and n > 0:
        if packed is None:
            packed = zlib.compress(raw, 1)
        chunk = image.add(packed)
        # chunk index: B-tree v1 node of type 1 (raw data chunks), one entry; a key is {chunk size,
        # filter mask, offset per dimension + one for the element}; the node has room for 2K
        key_size = 8 + 8 * 2
        node = b'TREE' + struct.pack('<BBHQQ', 1, 0, 1, UNDEF, UNDEF)
        node += struct.pack('<IIQQ', len(packed), 0, 0, 0) + struct.pack('<Q', chunk)
        node += struct.pack('<IIQQ', 0, 0, n, 0)
        btree = image.add(node + b'\0' * (24 + 2 * CHUNK_K * 8 + (2 * CHUNK_K + 1) * key_size -
                                          len(node)))
        # filter pipeline, version 1: deflate (id 1), no name, one client value (the level)
        messages.append(_message(0x000B, struct.pack('<BB6xHHHHII', 1, 1, 1, 0, 0, 1, 1, 0)))
        # data layout, version 3, chunked: rank + 1 dimensions, the last the element size
        messages.append(_message(0x0008, struct.pack('<BBBQII', 3, 2, 2, btree, n, 2)))
    else:
        addr = image.add(raw) if n else UNDEF
        messages.append(_message(0x0008, struct.pack('<BBQQ', 3, 1, addr, len(raw))))
    return image.add(_object_header(messages))


def _read_group(image, read_id, signal, compress, read_number, metadata, packed_signal=None):
    """/read_<read_id> with Raw/Signal and whatever ``metadata`` holds -> (header, btree, heap)."""
    signal = np.asarray(signal)
    dataset = _dataset_int16(image, signal, compress, packed_signal)
    metadata = metadata or {}
    attrs = [string_attribute('read_id', read_id)]
    raw_values = dict(metadata.get('Raw') or {})
    if 'duration' not in raw_values:
        attrs.append(int_attribute('duration', len(signal), 4, False))
    if read_number is not None:
        raw_values.pop('read_number', None)
        attrs.append(int_attribute('read_number', int(read_number), 4, True))
    attrs += _attributes(raw_values, skip=('read_id',))
    raw = _group(image, {'Signal': (dataset, None, None)}, attrs)
    children = {'Raw': raw}
    for name in ('channel_id', 'tracking_id', 'context_tags'):
        if metadata.get(name) is not None:
            children[name] = _attribute_group(image, _attributes(metadata[name]))
    return _group(image, children, _attributes(metadata.get('read') or {}))


def _finish(image, superblock, root):
    end = len(image.buf) + (-len(image.buf) % 8)
    image.buf += b'\0' * (end - len(image.buf))
    image.put(superblock, SIGNATURE + struct.pack(
        '<BBBBBBBBHHIQQQQ', 0, 0, 0, 0, 0, 8, 8, 0, getattr(image, 'leaf_k', GROUP_LEAF_K),
        GROUP_INTERNAL_K, 0, 0, UNDEF, end, UNDEF) + struct.pack('<QQII', 0, root[0], 1, 0) +
        struct.pack('<QQ', root[1], root[2]))
    return bytes(image.buf)


def single_read_fast5_bytes(read_id, signal, compress=True, read_number=None, metadata=None,
                            packed_signal=None):
    """The bytes of a one-read fast5 file holding ``signal`` (int16) as read ``read_id``.
    ``metadata``: what else the read's group of a multi-read container held, copied as
    ont_fast5_api's multi_to_single_fast5 does (the tool the reference runs, realtime.py:183-190):
    ``{'read': {attribute: value}, 'Raw': {...}, 'channel_id': {...}, 'tracking_id': {...},
    'context_tags': {...}}`` ('read': the read group's own attributes, e.g. run_id) - basecallers
    need ``channel_id`` (digitisation, offset, range, sampling_rate).  ``packed_signal``: the
    zlib stream of the signal's bytes where the caller has it already (a chunk as stored)."""
    image = _Image()
    superblock = image.reserve(96)
    read = _read_group(image, read_id, signal, compress, read_number, metadata, packed_signal)
    root = _group(image, {'read_' + read_id: read}, [string_attribute('file_version', '2.0')])
    return _finish(image, superblock, root)


def multi_read_fast5_bytes(reads, compress=True):
    """The bytes of a multi-read fast5 container (the layout MinKNOW and ont_fast5_api write:
    one ``/read_<id>`` group per read under the root).  ``reads``: (read_id, signal) or (read_id,
    signal, metadata) or (read_id, signal, metadata, deflated) tuples - ``deflated`` being
    ``zlib.compress(signal.tobytes(), 1)`` done beforehand (on other threads, or once for many
    reads that share a signal).  The root group's links share one symbol-table node; the file
    records the node size that takes (group leaf node K) in its superblock, as libhdf5 does for
    H5Pset_sym_k.  Used to build the containers of the streaming tests and tools."""
    reads = list(reads)
    image = _Image()
    image.leaf_k = GROUP_LEAF_K
    superblock = image.reserve(96)
    groups = {}
    for read in reads:
        read_id, signal = read[0], read[1]
        metadata = read[2] if len(read) > 2 else None
        packed = read[3] if len(read) > 3 else None
        name = 'read_' + read_id
        assert name not in groups, 'duplicate read id ' + read_id
        groups[name] = _read_group(image, read_id, signal, compress, None, metadata, packed)
    image.leaf_k = max(GROUP_LEAF_K, (len(groups) + 1) // 2)
    if groups:
        root = _group(image, groups, [string_attribute('file_version', '2.0')])
    else:
        root = _attribute_group(image, [string_attribute('file_version', '2.0')])
    return _finish(image, superblock, root)


def write_single_read_fast5(path, read_id, signal, compress=True, read_number=None,
                            metadata=None):
    """Writes the file under a temporary name beside ``path`` and links it into place: a file
    that is there already is never overwritten (FileExistsError - the reference's move counts such
    a clash and leaves the earlier file alone, realtime.py:111-144), and no partial file is ever
    seen under the final name."""
    image = single_read_fast5_bytes(read_id, signal, compress, read_number, metadata)
    if os.path.lexists(path):
        raise FileExistsError(path)
    tmp = '{}.part.{}.{}'.format(path, os.getpid(), threading.get_ident())
    try:
        os.unlink(tmp)          # (a stale one of this very name: a killed run whose pid came round)
    except OSError:
        pass
    try:
        with open(tmp, 'xb') as f:
            f.write(image)
        try:
            os.link(tmp, path)
        except FileExistsError:
            raise
        except OSError as e:
            # no hard links here (exFAT / FAT, many SMB and FUSE mounts - sequencing drives): the
            # finished temporary file is RENAMED into place, after a look at the name - nobody sees
            # a partial file under the final name, a killed process leaves only the .part file;
            # rename is atomic there too, the window between the look and the rename is the one
            # thing link() had closed
            if e.errno not in _NO_LINK_ERRNOS:
                raise
            if os.path.lexists(path):
                raise FileExistsError(path)
            os.rename(tmp, path)
    finally:
        try:
            os.unlink(tmp)
        except OSError:
            pass


# what link() answers where the FILESYSTEM has no hard links (EPERM: FAT / exFAT, several FUSE and SMB
# mounts); EACCES, EXDEV, EMLINK are errors of the call and are raised
_NO_LINK_ERRNOS = frozenset(
    getattr(errno, name) for name in ('EPERM', 'ENOTSUP', 'EOPNOTSUPP', 'ENOSYS') if hasattr(errno, name))

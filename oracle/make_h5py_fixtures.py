#!/opt/conda/bin/python3.9
"""
ORACLE tooling - fast5-shaped HDF5 files written by the real HDF5 library, to pin this
repository's two HDF5 readers (deepbinner_amd/hdf5_lite.py and csrc/fast5_reader.cpp) against it.

Run with the image's conda interpreter, the only one that has h5py (3.3.0 on libhdf5 1.10.6):
    /opt/conda/bin/python3.9 oracle/make_h5py_fixtures.py [out_dir [seed [committed|full]]]
Default out_dir is tests/golden/fast5/h5py_variants/ (committed: small files + expected.json with
the read ids and a SHA-256 of every signal, profile `committed`).  tests/test_fast5_native.py
reads every file with both readers; where the conda interpreter exists it also runs this script
with profile `full` (adds the cases too big to commit: 700-read containers, a 100k-sample read,
chunk indexes with thousands to 140,000 chunks, sparsely written datasets), another seed and a
temporary out_dir, so that the comparison is not limited to the committed files.

Variants: both on-disk generations (libver earliest: superblock v0, symbol-table groups, v1 object
headers; libver latest: superblock v3, compact/dense link storage, v2 object headers), the three
fast5 layouts (old single /Raw/Reads/Read_N, new single /read_<id>/Raw, multi), group sizes across
the compact->dense and one-node->multi-level boundaries (1..700 reads), Signal stored compact,
contiguous and chunked (chunk longer than the data, many chunks, edge chunk), with gzip levels,
shuffle, fletcher32 and no filter, signal lengths 0..100k, read_id as fixed bytes / variable-length
string / NUL-padded, and enough extra attributes to push a group into dense attribute storage.
"""
import hashlib
import json
import os
import sys
import uuid

import h5py
import numpy as np


def signal(rng, n):
    levels = rng.normal(450, 80, size=n // 8 + 1)
    return np.clip(np.rint(np.repeat(levels, 8)[:n] + rng.normal(0, 8, size=n)), 0, 2047) \
        .astype(np.int16)


def put_signal(group, data, how):
    kw = {}
    kind = how.get('storage', 'chunked')
    if kind == 'chunked':
        kw['chunks'] = (max(1, how.get('chunk', max(1, len(data)))),)
        if how.get('gzip') is not None:
            kw['compression'] = 'gzip'
            kw['compression_opts'] = how['gzip']
        kw['shuffle'] = how.get('shuffle', False)
        kw['fletcher32'] = how.get('fletcher32', False)
        if kw['chunks'][0] > len(data) or how.get('unlimited'):
            kw['maxshape'] = (None,)           # (h5py wants that for a chunk longer than the data)
    if how.get('sparse'):                      # most chunks never written: they read as zeros
        ds = group.create_dataset('Signal', shape=data.shape, dtype='<i2', **kw)
        keep = np.zeros(len(data), dtype=bool)
        for lo in how['sparse']:
            keep[lo:lo + 3] = True
            ds[lo:lo + 3] = data[lo:lo + 3]
        data[~keep] = 0
        return ds
    return group.create_dataset('Signal', data=data, dtype='<i2', **kw)


def compact_signal(group, data):
    space = h5py.h5s.create_simple((len(data),))
    plist = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
    plist.set_layout(h5py.h5d.COMPACT)
    dset = h5py.h5d.create(group.id, b'Signal', h5py.h5t.STD_I16LE, space, plist)
    dset.write(h5py.h5s.ALL, h5py.h5s.ALL, np.ascontiguousarray(data))


def set_read_id(group, read_id, style):
    if style == 'bytes':
        group.attrs['read_id'] = np.bytes_(read_id)
    elif style == 'vlen':
        group.attrs['read_id'] = read_id                       # h5py: variable-length UTF-8
    elif style == 'padded':
        group.attrs.create('read_id', np.bytes_(read_id), dtype='S40')
    else:
        raise ValueError(style)


def extra_attrs(group, rng, n):
    for k in range(n):
        group.attrs['extra_%02d' % k] = int(rng.integers(0, 1 << 30))
    group.attrs['start_time'] = np.uint64(rng.integers(0, 1 << 40))
    group.attrs['duration'] = np.uint32(123)
    group.attrs['median_before'] = 201.5


def write_case(path, case, rng):
    libver = (case['libver'], 'latest')     # 'earliest' = what h5py writes by default
    reads = []
    with h5py.File(path, 'w', libver=libver) as f:
        f.attrs['file_version'] = np.bytes_('2.0')
        for k in range(case['reads']):
            read_id = str(uuid.UUID(bytes=rng.bytes(16), version=4))
            data = signal(rng, int(case['lengths'][k % len(case['lengths'])]))
            if case['layout'] == 'single_old':
                raw = f.create_group('Raw/Reads/Read_%d' % (100 + k))
                f.create_group('UniqueGlobalKey/channel_id').attrs['channel_number'] = np.bytes_('7')
                f.create_group('Analyses')
            else:
                read = f.create_group('read_' + read_id)
                read.attrs['run_id'] = np.bytes_('0a1b2c3d')
                raw = read.create_group('Raw')
                read.create_group('channel_id').attrs['offset'] = 3.0
            set_read_id(raw, read_id, case['read_id'])
            extra_attrs(raw, rng, case.get('extra_attrs', 0))
            how = case['signal']
            if how.get('storage') == 'compact':
                compact_signal(raw, data)
            else:
                put_signal(raw, data, how)
            reads.append((read_id, data))
    return reads


def cases(profile):
    full = profile == 'full'
    base = dict(reads=1, lengths=[4000], read_id='bytes',
                signal=dict(storage='chunked', chunk=4000, gzip=1))
    out = []

    def add(name, **kw):
        out.append(dict(base, name=name, **kw))

    for libver in ('earliest', 'latest'):
        tag = 'old' if libver == 'earliest' else 'new'
        add('single_old_layout_%s' % tag, libver=libver, layout='single_old')
        add('single_new_layout_%s' % tag, libver=libver, layout='multi', reads=1, read_id='vlen')
        add('multi_3_%s' % tag, libver=libver, layout='multi', reads=3, lengths=[900, 1500, 5])
        add('multi_9_%s' % tag, libver=libver, layout='multi', reads=9, lengths=[700, 64],
            read_id='vlen')
        add('multi_12_%s' % tag, libver=libver, layout='multi', reads=12, lengths=[300],
            read_id='padded', extra_attrs=9)
        if full:
            add('multi_700_%s' % tag, libver=libver, layout='multi', reads=700,
                lengths=[40, 24, 8], signal=dict(storage='chunked', chunk=64, gzip=4),
                extra_attrs=12)
        add('contiguous_%s' % tag, libver=libver, layout='single_old',
            signal=dict(storage='contiguous'))
        add('compact_%s' % tag, libver=libver, layout='single_old', lengths=[500],
            signal=dict(storage='compact'))
        add('many_chunks_shuffle_fletcher_%s' % tag, libver=libver, layout='single_old',
            lengths=[25001], signal=dict(storage='chunked', chunk=1000, gzip=9, shuffle=True,
                                         fletcher32=True))
        add('chunked_no_filter_%s' % tag, libver=libver, layout='single_old', lengths=[3000],
            signal=dict(storage='chunked', chunk=1024))
        add('chunk_longer_than_data_%s' % tag, libver=libver, layout='multi', reads=2,
            lengths=[100, 1], signal=dict(storage='chunked', chunk=4096, gzip=1, shuffle=True))
        add('fletcher_only_%s' % tag, libver=libver, layout='single_old', lengths=[2500],
            signal=dict(storage='chunked', chunk=512, fletcher32=True))
        add('empty_signal_%s' % tag, libver=libver, layout='multi', reads=2, lengths=[0, 12])
        n_long = 100000 if full else 30000
        add('long_read_%s' % tag, libver=libver, layout='single_old', lengths=[n_long],
            signal=dict(storage='chunked', chunk=n_long, gzip=1), read_id='vlen',
            extra_attrs=10)
        # chunk indexes: fixed array (fixed size) / extensible array (unlimited) with libver
        # latest, v1 B-tree otherwise; a few dozen chunks here, thousands below
        add('chunks_40_fixed_%s' % tag, libver=libver, layout='single_old', lengths=[40 * 7 + 3],
            signal=dict(storage='chunked', chunk=7, gzip=1))
        add('chunks_150_unlimited_%s' % tag, libver=libver, layout='single_old', lengths=[150 * 5],
            signal=dict(storage='chunked', chunk=5, unlimited=True))
        add('chunks_sparse_%s' % tag, libver=libver, layout='single_old', lengths=[900],
            signal=dict(storage='chunked', chunk=3, unlimited=(libver == 'latest'), gzip=1,
                        sparse=[0, 300, 451, 897]))
        if full:
            add('chunks_2000_fixed_%s' % tag, libver=libver, layout='single_old',
                lengths=[2000 * 3 + 1], signal=dict(storage='chunked', chunk=3))
            add('chunks_2000_fixed_gzip_%s' % tag, libver=libver, layout='single_old',
                lengths=[2000 * 5], signal=dict(storage='chunked', chunk=5, gzip=1, shuffle=True))
            add('chunks_5000_unlimited_gzip_%s' % tag, libver=libver, layout='single_old',
                lengths=[5000 * 2], signal=dict(storage='chunked', chunk=2, gzip=1,
                                                unlimited=True))
            add('chunks_9000_sparse_fixed_%s' % tag, libver=libver, layout='single_old',
                lengths=[9000], signal=dict(storage='chunked', chunk=3,
                                            sparse=[0, 300, 4500, 8997]))
    if full:    # a container over 8 MiB: the native reader maps those and preads the payloads
        add('container_over_8MiB_old', libver='earliest', layout='multi', reads=40,
            lengths=[120000, 90000], signal=dict(storage='chunked', chunk=30000))
    if full:    # 140,001 one-sample chunks: paged data blocks behind super blocks
        add('chunks_140k_unlimited_new', libver='latest', layout='single_old', lengths=[140001],
            signal=dict(storage='chunked', chunk=1, unlimited=True))
        add('chunks_140k_unlimited_fletcher_new', libver='latest', layout='single_old',
            lengths=[140001], signal=dict(storage='chunked', chunk=1, unlimited=True,
                                          fletcher32=True))
        add('chunks_140k_sparse_unlimited_new', libver='latest', layout='single_old',
            lengths=[140001], signal=dict(storage='chunked', chunk=1, unlimited=True,
                                          sparse=[0, 131060, 133200, 139998]))
    return out


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        here, 'tests', 'golden', 'fast5', 'h5py_variants')
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260928
    profile = sys.argv[3] if len(sys.argv) > 3 else 'committed'
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    expected = {'h5py': h5py.__version__, 'hdf5': h5py.version.hdf5_version, 'seed': seed,
                'files': {}}
    for case in cases(profile):
        name = case['name'] + '.fast5'
        reads = write_case(os.path.join(out_dir, name), case, rng)
        # what h5py itself reads back, in its own (name-sorted) iteration order
        with h5py.File(os.path.join(out_dir, name), 'r') as f:
            if 'Raw' in f:
                groups = [g for _, g in f['Raw/Reads'].items()]
            else:
                groups = [f[k]['Raw'] for k in f if k.startswith('read_')]
            back = []
            for g in groups:
                rid = g.attrs['read_id']
                rid = rid.decode() if isinstance(rid, bytes) else str(rid)
                back.append((rid, g['Signal'][:]))
        assert sorted(r[0] for r in back) == sorted(r[0] for r in reads)
        expected['files'][name] = {
            'layout': case['layout'], 'libver': case['libver'],
            'reads': [{'read_id': rid, 'n': int(len(sig)),
                       'sha256': hashlib.sha256(np.ascontiguousarray(sig, dtype='<i2').tobytes())
                       .hexdigest()} for rid, sig in back]}
    with open(os.path.join(out_dir, 'expected.json'), 'wt') as f:
        json.dump(expected, f, indent=1, sort_keys=True)
    print(len(expected['files']), 'files in', out_dir)


if __name__ == '__main__':
    main()

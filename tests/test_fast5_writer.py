"""The native one-read fast5 writer (f5_write_single_reads in deepbinner_amd/csrc/fast5_reader.cpp):
what `deepbinner realtime` files the reads of multi-read containers with.  Its oracle is the
Python writer (deepbinner_amd/hdf5_write.py), which tests/test_hdf5_write.py pins to the real HDF5
library and to the reference's own loader: same bytes for the same read, on the real MinKNOW
containers and on containers whose Signal is stored every other way (contiguous, several chunks,
shuffled) - and the files are read back by both of this package's readers and, where the image
has it, by h5py."""
import glob
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD
from deepbinner_amd import fast5_native, hdf5_lite, hdf5_write

CONDA_PYTHON = '/opt/conda/bin/python3.9'
MULTI = sorted(glob.glob(os.path.join(GOLD, 'fast5', 'multi', '*.fast5')))


def metadata_of(path, read_id):
    import deepbinner_amd.realtime as realtime
    source = realtime.MetadataSource()
    try:
        return source(path, read_id)
    finally:
        source.close()


def stored_chunks(path):
    """{read index: the zlib stream its Signal is stored as} for reads stored as ONE deflated
    chunk (what the raw loader hands over for the GPU to inflate)."""
    gen = fast5_native.stream_raw([path], threads=1)
    _, _, _, _, comp, records = next(gen)
    gen.close()
    out = {}
    for i in set(records['read'].tolist()):
        mine = records[records['read'] == i]
        if len(mine) == 1 and mine[0]['mode'] == 0:
            at = int(mine[0]['comp_offset'])
            out[i] = bytes(comp[at:at + int(mine[0]['comp_bytes'])])
    return out


def python_writer_bytes(path, i, read_id, signal, packed):
    return hdf5_write.single_read_fast5_bytes(read_id, signal, metadata=metadata_of(path, read_id),
                                              packed_signal=packed)


@pytest.mark.parametrize('path', MULTI, ids=[os.path.basename(p)[-12:] for p in MULTI])
def test_same_bytes_as_the_python_writer_on_minknow_containers(path):
    """Every read of the three real containers: attribute for attribute (read group, Raw,
    channel_id, tracking_id, context_tags), structure for structure, the chunk carried over as
    stored - the file the Python writer builds from hdf5_lite's view of the read."""
    ids, samples, offsets, status = fast5_native.load_reads(path, threads=2)
    assert (status == 0).all() and len(ids) == 10
    chunks = stored_chunks(path)
    assert len(chunks) == len(ids)                  # MinKNOW: one deflated chunk per read
    for i, read_id in enumerate(ids):
        signal = np.array(samples[offsets[i]:offsets[i + 1]])
        want = python_writer_bytes(path, i, read_id, signal, chunks[i])
        assert fast5_native.single_read_image(path, i) == want, (path, i)
        # and the chunk as stored is what zlib level 1 makes of the samples, or at least inflates
        # to them
        import zlib
        assert zlib.decompress(chunks[i]) == signal.astype('<i2').tobytes()


def build_container(path, how, rng):
    """12 reads with metadata, their Signals stored `how`."""
    reads = []
    for k in range(12):
        n = int(rng.integers(1, 60000)) if k else 0          # one empty read
        signal = rng.integers(-2000, 2000, n).astype(np.int16)
        metadata = {
            'read': {'run_id': b'run-%d' % k, 'pore_type': 'not_set'},
            'Raw': {'start_time': np.uint64(1000 * k), 'duration': np.uint32(n),
                    'read_number': np.int32(k), 'start_mux': np.uint8(k % 4),
                    'median_before': np.float64(200.5 + k), 'num_minknow_events': np.int64(-k)},
            'channel_id': {'channel_number': str(100 + k), 'digitisation': np.float64(8192.0),
                           'offset': np.float64(-3.0 + k), 'range': np.float32(1467.6),
                           'sampling_rate': np.float64(4000.0)},
            'tracking_id': {'device_id': 'MN12345', 'exp_start_time': '2026-09-28T00:00:00Z',
                            'flow_cell_id': ''},
        }
        if k % 3:
            metadata['context_tags'] = {'sequencing_kit': 'sqk-lsk109', 'barcoding_enabled': '0'}
        reads.append(('%08x-aaaa-4bbb-8ccc-%012x' % (k, 31 * k), signal, metadata))
    with open(path, 'wb') as f:
        f.write(hdf5_write.multi_read_fast5_bytes(reads, compress=(how == 'one chunk')))
    return reads


@pytest.mark.parametrize('how', ['one chunk', 'contiguous'])
def test_same_bytes_whatever_way_the_signal_is_stored(tmp_path, how):
    """A Signal that is not one deflated chunk (here: contiguous) is decoded and deflated again
    (zlib level 1, like the Python writer); integers of every width and sign, floats of both
    sizes, strings (also empty) travel; an empty read has no chunk at all."""
    rng = np.random.default_rng(len(how))
    path = str(tmp_path / 'container.fast5')
    reads = build_container(path, how, rng)
    ids, samples, offsets, status = fast5_native.load_reads(path, threads=2)
    assert (status == 0).all()
    chunks = stored_chunks(path)
    assert (len(chunks) == 11) == (how == 'one chunk')        # (the empty read has none)
    by_id = {r[0]: r for r in reads}
    for i, read_id in enumerate(ids):
        signal = np.array(samples[offsets[i]:offsets[i + 1]])
        assert np.array_equal(signal, by_id[read_id][1])
        want = python_writer_bytes(path, i, read_id, signal, chunks.get(i))
        assert fast5_native.single_read_image(path, i) == want, (how, i)


H5PY_CONTAINER = r'''
import sys
import h5py, numpy as np
path = sys.argv[1]
rng = np.random.default_rng(7)
with h5py.File(path, 'w') as f:
    f.attrs['file_version'] = np.bytes_('2.0')
    for k, (chunks, opts) in enumerate([((1000,), dict(compression='gzip', compression_opts=4)),
                                        ((5000,), dict(compression='gzip', shuffle=True)),
                                        (None, {}),
                                        ((7001,), dict(compression='gzip', compression_opts=1)),
                                        ((7001,), dict(compression='gzip', fletcher32=True))]):
        n = 7001
        read_id = '%08x-1111-4222-8333-%012x' % (k, k)
        g = f.create_group('read_' + read_id)
        g.attrs['run_id'] = np.bytes_('abc')
        raw = g.create_group('Raw')
        raw.attrs['read_id'] = read_id                       # variable-length string
        raw.attrs['start_time'] = np.uint64(5 * k)
        raw.attrs['duration'] = np.uint32(n)
        raw.attrs['end_reason'] = np.array([1, 2, 3])         # an array: left behind
        raw.create_dataset('Signal', data=rng.integers(0, 900, n).astype('<i2'), chunks=chunks, **opts)
        c = g.create_group('channel_id')
        c.attrs['channel_number'] = str(k)                    # variable-length string
        c.attrs['sampling_rate'] = 4000.0
        c.attrs['big'] = np.array(258, dtype='>i4')           # big-endian: rewritten little-endian
'''


@pytest.mark.skipif(not os.path.exists(CONDA_PYTHON), reason='no h5py interpreter in this image')
def test_containers_written_by_the_hdf5_library(tmp_path):
    """Chunked in pieces, other deflate levels, shuffle, fletcher32, contiguous; variable-length
    strings, a big-endian integer, an array attribute: same bytes as the Python writer, and h5py
    reads the one-read files back - same groups, attributes and values as the container holds."""
    path = str(tmp_path / 'h5py.fast5')
    subprocess.run([CONDA_PYTHON, '-c', H5PY_CONTAINER, path], check=True, timeout=300)
    ids, samples, offsets, status = fast5_native.load_reads(path, threads=1)
    assert (status == 0).all() and len(ids) == 5
    chunks = stored_chunks(path)
    out = [str(tmp_path / ('read_%d.fast5' % i)) for i in range(5)]
    done, written = fast5_native.write_single_reads(path, list(range(5)), out, threads=3)
    assert (done == 0).all() and written == sum(os.path.getsize(p) for p in out)
    passed_through = 0
    for i, read_id in enumerate(ids):
        signal = np.array(samples[offsets[i]:offsets[i + 1]])
        # only a read stored as ONE chunk by deflate ALONE is carried over as stored
        got = open(out[i], 'rb').read()
        for candidate in (chunks.get(i), None):
            if got == python_writer_bytes(path, i, read_id, signal, candidate):
                passed_through += candidate is not None
                break
        else:
            raise AssertionError('read %d differs from the Python writer' % i)
    assert passed_through == 1                      # (read 3; read 4's chunk ends in a checksum)
    check = r'''
import sys, json
import h5py, numpy as np
src = h5py.File(sys.argv[1], 'r')
report = []
for k, path in enumerate(sys.argv[2:]):
    with h5py.File(path, 'r') as f:
        (name,) = list(f)
        theirs = src[name]
        same = np.array_equal(f[name]['Raw/Signal'][:], theirs['Raw/Signal'][:])
        for group in ('', 'Raw', 'channel_id'):
            a = (f[name][group] if group else f[name]).attrs
            b = (theirs[group] if group else theirs).attrs
            for key in b:
                if np.ndim(b[key]):
                    same = same and key not in a
                    continue
                va, vb = a[key], b[key]
                va = va.decode() if isinstance(va, bytes) else va
                vb = vb.decode() if isinstance(vb, bytes) else vb
                same = same and va == vb
        report.append(bool(same))
print(json.dumps(report))
'''
    result = subprocess.run([CONDA_PYTHON, '-c', check, path] + out, check=True, timeout=300,
                            capture_output=True, text=True)
    assert json.loads(result.stdout) == [True] * 5


def test_files_on_disk_and_what_goes_wrong(tmp_path):
    """write_single_reads: the files both readers read back (id, signal, metadata); a read asked
    for twice is written twice; an index beyond the container, a directory that does not exist and
    a missing container are statuses / an error, not crashes."""
    path = MULTI[0]
    ids, samples, offsets, status = fast5_native.load_reads(path, threads=2)
    out_dir = tmp_path / 'out'
    out_dir.mkdir()
    wanted = [3, 0, 9, 3, 12345, 5]
    paths = [str(out_dir / ('r%d.fast5' % k)) for k in range(len(wanted))]
    paths[5] = str(tmp_path / 'no_such_directory' / 'r5.fast5')
    done, written = fast5_native.write_single_reads(path, wanted, paths, threads=4)
    assert done[:4].tolist() == [0, 0, 0, 0]
    assert done[4] == fast5_native.F5_ERR_NO_READ and done[5] != 0
    assert not os.path.exists(paths[4]) and written == sum(os.path.getsize(p) for p in paths[:4])
    assert open(paths[0], 'rb').read() == open(paths[3], 'rb').read()
    for k in range(4):
        i = wanted[k]
        for reader in (fast5_native, ):
            got_id, got = reader.get_read_id_and_signal(paths[k])
            assert got_id == ids[i] and np.array_equal(got, samples[offsets[i]:offsets[i + 1]])
        with hdf5_lite.File(paths[k], 'r') as f, hdf5_lite.File(path, 'r') as container:
            mine, theirs = f['read_' + ids[i]], container['read_' + ids[i]]
            assert np.array_equal(mine['Raw']['Signal'][:], theirs['Raw']['Signal'][:])
            for group in ('channel_id', 'tracking_id', 'context_tags'):
                assert dict(mine[group].attrs.items()) == dict(theirs[group].attrs.items())
            assert dict(mine['Raw'].attrs.items()) == dict(theirs['Raw'].attrs.items())
            assert dict(mine.attrs.items()) == dict(theirs.attrs.items())
    with pytest.raises(OSError):
        fast5_native.write_single_reads(str(tmp_path / 'missing.fast5'), [0], [paths[0]])
    empty, written = fast5_native.write_single_reads(path, [], [])
    assert len(empty) == 0 and written == 0


def test_nothing_is_ever_overwritten(tmp_path):
    """A second run into the same directory (or a read id seen before) must not replace what an
    earlier run filed there - the reference never moves a file over another one
    (realtime.py:111-144: a clash is counted and skipped): a path that exists, as a file or as a
    symlink (even a dangling one, which an open with O_CREAT would follow), is left alone and
    reported as F5_ERR_EXISTS; no temporary file stays behind; both writers agree."""
    path = MULTI[0]
    ids, samples, offsets, _ = fast5_native.load_reads(path, threads=2)
    out_dir = tmp_path / 'out'
    out_dir.mkdir()
    targets = [str(out_dir / ('r%d.fast5' % k)) for k in range(4)]
    open(targets[1], 'wb').write(b'filed by an earlier run')
    outside = tmp_path / 'elsewhere.fast5'
    os.symlink(str(outside), targets[2])                     # dangling
    done, written = fast5_native.write_single_reads(path, [0, 1, 2, 3], targets, threads=3)
    assert done.tolist() == [0, fast5_native.F5_ERR_EXISTS, fast5_native.F5_ERR_EXISTS, 0]
    assert open(targets[1], 'rb').read() == b'filed by an earlier run'
    assert not outside.exists() and os.path.islink(targets[2])
    assert written == os.path.getsize(targets[0]) + os.path.getsize(targets[3])
    assert sorted(os.listdir(str(out_dir))) == ['r0.fast5', 'r1.fast5', 'r2.fast5', 'r3.fast5']
    assert fast5_native.status_string(fast5_native.F5_ERR_EXISTS).startswith('a file of that name')
    again, _ = fast5_native.write_single_reads(path, [0, 3], [targets[0], targets[3]])
    assert again.tolist() == [fast5_native.F5_ERR_EXISTS] * 2
    # the Python writer
    signal = samples[offsets[4]:offsets[5]]
    fresh = str(out_dir / 'py.fast5')
    hdf5_write.write_single_read_fast5(fresh, ids[4], signal)
    before = open(fresh, 'rb').read()
    for taken in (fresh, targets[1], targets[2]):
        with pytest.raises(FileExistsError):
            hdf5_write.write_single_read_fast5(taken, ids[5], samples[offsets[5]:offsets[6]])
    assert open(fresh, 'rb').read() == before and not outside.exists()
    assert len(os.listdir(str(out_dir))) == 5


def test_filesystems_without_hard_links(tmp_path, monkeypatch):
    """exFAT / FAT and many SMB or FUSE mounts refuse link() (EPERM, ENOTSUP ...): both writers
    then rename the finished temporary file into place (without replacing) - same bytes, still nothing
    overwritten, no temporary file left (ADVICE rounds 4, 5).  EACCES is an error, not "no links".  The native library is run in a child
    process with DEEPBINNER_FAST5_NO_LINK=1 (it reads the switch once)."""
    import errno
    import subprocess
    import sys
    path = MULTI[0]
    ids, samples, offsets, _ = fast5_native.load_reads(path, threads=2)
    out_dir = tmp_path / 'out'
    out_dir.mkdir()

    def refuse(src, dst, **kw):
        raise OSError(errno.EPERM, 'Operation not permitted')
    monkeypatch.setattr(os, 'link', refuse)
    target = str(out_dir / 'py.fast5')
    hdf5_write.write_single_read_fast5(target, ids[0], samples[offsets[0]:offsets[1]])
    assert open(target, 'rb').read() == hdf5_write.single_read_fast5_bytes(
        ids[0], samples[offsets[0]:offsets[1]])
    with pytest.raises(FileExistsError):
        hdf5_write.write_single_read_fast5(target, ids[1], samples[offsets[1]:offsets[2]])
    dangling = str(out_dir / 'link.fast5')
    os.symlink(str(tmp_path / 'elsewhere'), dangling)
    with pytest.raises(FileExistsError):
        hdf5_write.write_single_read_fast5(dangling, ids[1], samples[offsets[1]:offsets[2]])
    assert sorted(os.listdir(str(out_dir))) == ['link.fast5', 'py.fast5']
    # a failing rename leaves neither a temporary nor a final file (the finished
    # temporary file is renamed into place, so that no partial file is ever seen under the final name)
    monkeypatch.setattr(os, 'rename', lambda a, b: (_ for _ in ()).throw(OSError(errno.ENOSPC, 'full')))
    with pytest.raises(OSError):
        hdf5_write.write_single_read_fast5(str(out_dir / 'never.fast5'), ids[2],
                                           samples[offsets[2]:offsets[3]])
    assert sorted(os.listdir(str(out_dir))) == ['link.fast5', 'py.fast5']
    monkeypatch.undo()

    def denied(src, dst, **kw):
        raise OSError(errno.EACCES, 'Permission denied')
    monkeypatch.setattr(os, 'link', denied)
    with pytest.raises(PermissionError):
        hdf5_write.write_single_read_fast5(str(out_dir / 'denied.fast5'), ids[2], samples[offsets[2]:offsets[3]])
    assert sorted(os.listdir(str(out_dir))) == ['link.fast5', 'py.fast5']
    monkeypatch.undo()

    native_dir = tmp_path / 'native'
    native_dir.mkdir()
    open(str(native_dir / 'r1.fast5'), 'wb').write(b'earlier')
    code = (
        'import sys, json\n'
        'sys.path.insert(0, %r)\n'
        'from deepbinner_amd import fast5_native\n'
        'done, written = fast5_native.write_single_reads(%r, [0, 1, 2], [%r + "/r%%d.fast5" %% k for k in range(3)], threads=2)\n'
        'print(json.dumps([done.tolist(), int(written)]))\n'
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), path, str(native_dir))
    env = dict(os.environ, DEEPBINNER_FAST5_NO_LINK='1')
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, check=True)
    import json
    done, written = json.loads(out.stdout.strip().splitlines()[-1])
    assert done == [0, fast5_native.F5_ERR_EXISTS, 0]
    assert open(str(native_dir / 'r1.fast5'), 'rb').read() == b'earlier'
    assert sorted(os.listdir(str(native_dir))) == ['r0.fast5', 'r1.fast5', 'r2.fast5']
    for k in (0, 2):
        image = open(str(native_dir / ('r%d.fast5' % k)), 'rb').read()
        assert image == fast5_native.single_read_image(path, k)
    assert written == sum(os.path.getsize(str(native_dir / ('r%d.fast5' % k))) for k in (0, 2))


def test_damaged_containers_never_crash_the_writer(tmp_path):
    """300 seeded mutations of a real container (byte flips, zeroed runs, truncations): every read
    is either written - and then both readers read the file back - or refused with a status;
    the process survives all of them."""
    rng = np.random.default_rng(20260928)
    original = open(MULTI[1], 'rb').read()
    victim = str(tmp_path / 'victim.fast5')
    written = refused = 0
    for round_ in range(300):
        data = bytearray(original)
        kind = round_ % 3
        if kind == 0:
            for _ in range(int(rng.integers(1, 40))):
                data[int(rng.integers(0, len(data)))] ^= int(rng.integers(1, 256))
        elif kind == 1:
            at = int(rng.integers(0, len(data) - 64))
            run = int(rng.integers(1, 4096))
            data[at:at + run] = bytes(len(data[at:at + run]))
        else:
            del data[int(rng.integers(len(data) // 8, len(data))):]
        with open(victim, 'wb') as f:
            f.write(bytes(data))
        targets = [str(tmp_path / ('out_%d.fast5' % i)) for i in range(10)]
        for target in targets:          # (the writer never writes over a file)
            if os.path.exists(target):
                os.unlink(target)
        try:
            status, _ = fast5_native.write_single_reads(victim, list(range(10)), targets, threads=2)
        except OSError:
            refused += 10
            continue
        for i, st in enumerate(status.tolist()):
            if st != 0:
                refused += 1
                continue
            written += 1
            try:
                got_id, got = fast5_native.get_read_id_and_signal(targets[i])
                if got_id is None:      # (a chunk carried over as stored: damaged deflate data)
                    continue
                with hdf5_lite.File(targets[i], 'r') as f:
                    (name,) = list(f.keys())
                    assert name == 'read_' + got_id
                    assert np.array_equal(f[name]['Raw']['Signal'][:], got)
            except OSError:
                pass                # a chunk carried over as stored may be damaged deflate data
    assert written > 500 and refused > 100, (written, refused)

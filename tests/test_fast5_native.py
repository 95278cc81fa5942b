"""The native fast5 loader (libdeepbinner_fast5.so, C ABI include/deepbinner_fast5.h) against the
pure-Python reader (hdf5_lite) and the reference's own loader answers
(reference tests/test_load_fast5s.py) - host only, no GPU."""
import hashlib
import json
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from conftest import REPO
from deepbinner_amd import fast5_native, hdf5_lite, load_fast5s

FAST5_DIR = os.path.join(REPO, 'tests', 'golden', 'fast5', 'single')
MULTI_DIR = os.path.join(REPO, 'tests', 'golden', 'fast5', 'multi')

pytestmark = pytest.mark.skipif(not fast5_native.available(),
                                reason='libdeepbinner_fast5.so not built')


def single_files():
    return sorted(os.path.join(FAST5_DIR, f) for f in os.listdir(FAST5_DIR) if f.endswith('.fast5'))


def multi_files():
    return sorted(os.path.join(MULTI_DIR, f) for f in os.listdir(MULTI_DIR) if f.endswith('.fast5'))


def python_reads(path):
    """[(read_id, signal)] through hdf5_lite, in h5py's (sorted-name) order."""
    with hdf5_lite.File(path) as f:
        keys = list(f.keys())
        if 'Raw' in keys:
            groups = [list(f['Raw/Reads/'].values())[0]]
        else:
            groups = [f[k + '/Raw/'] for k in keys if k.startswith('read_')]
        return [(g.attrs['read_id'].decode(), g['Signal'][:]) for g in groups]


def test_library_exports_every_declared_symbol():
    lib = fast5_native.load_library()
    header = open(os.path.join(REPO, 'include', 'deepbinner_fast5.h')).read()
    declared = set(re.findall(r'\b(f5_[a-z_0-9]+)\s*\(', header))
    assert declared == set(fast5_native.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.f5_version().startswith(b'deepbinner_fast5')


@pytest.mark.parametrize('path', single_files())
def test_single_read_files_match_python_reader(path):
    (want_id, want_signal), = python_reads(path)
    read_id, signal = fast5_native.get_read_id_and_signal(path)
    assert read_id == want_id
    assert signal.dtype == np.int16 and np.array_equal(signal, want_signal)
    with fast5_native.File(path) as f:
        assert f.n_reads == 1
        assert f.layout in (fast5_native.LAYOUT_SINGLE_OLD, fast5_native.LAYOUT_SINGLE_NEW)
        assert f.read_info(0) == (want_id, len(want_signal))
        # partial reads: only the chunks that overlap are decoded
        n = len(want_signal)
        for first, count in ((0, 1), (0, min(n, 6656)), (max(n - 6656, 0), min(n, 6656)),
                             (n // 2, min(1000, n - n // 2)), (n - 1, 1), (n, 0), (17, 4097)):
            if first + count > n:
                continue
            assert np.array_equal(f.read_signal(0, first, count), want_signal[first:first + count])
        with pytest.raises(KeyError):
            f.read_signal(0, n - 3, 10)
        with pytest.raises(KeyError):
            f.read_info(1)


def test_reference_loader_answers():
    """reference tests/test_load_fast5s.py:46-49,55-58,69-72"""
    by_name = {os.path.basename(p): p for p in single_files()}
    cases = [('5210_N128870_20180511_FAH70336_MN20200_sequencing_run_057_Deepbinner_amplicon_43629_'
              'read_11206_ch_157_strand.fast5', '63c20e8e-1d92-4a69-9cf8-9fc9c72ca4fb', 4971, 714,
              4950, 396)]
    for name, read_id, length, first, idx, val in cases:
        got_id, signal = fast5_native.get_read_id_and_signal(by_name[name])
        want_id, want = load_fast5s.get_read_id_and_signal(by_name[name])
        assert got_id == want_id and np.array_equal(signal, want)
        if want_id == read_id:
            assert len(signal) == length and signal[0] == first and signal[idx] == val


@pytest.mark.parametrize('path', multi_files())
def test_multi_read_files_match_python_reader(path):
    want = python_reads(path)
    got = list(fast5_native.iter_reads(path))
    assert len(got) == len(want) > 1
    for (rid, sig), (want_id, want_sig) in zip(got, want):
        assert rid == want_id and np.array_equal(sig, want_sig)
    with fast5_native.File(path) as f:
        assert f.layout == fast5_native.LAYOUT_MULTI and f.n_reads == len(want)
    with pytest.raises(SystemExit):
        fast5_native.get_read_id_and_signal(path)


# ---- files written by the real HDF5 library (oracle/make_h5py_fixtures.py) ---------------------
VARIANT_DIR = os.path.join(REPO, 'tests', 'golden', 'fast5', 'h5py_variants')
CONDA_PYTHON = '/opt/conda/bin/python3.9'          # the image's only interpreter with h5py


def sha(signal):
    return hashlib.sha256(np.ascontiguousarray(signal, dtype='<i2').tobytes()).hexdigest()


def check_against_expected(folder):
    """Every file of `folder` through both readers against what h5py read back from it."""
    with open(os.path.join(folder, 'expected.json')) as f:
        expected = json.load(f)['files']
    for name, info in sorted(expected.items()):
        path = os.path.join(folder, name)
        want = [(r['read_id'], r['n'], r['sha256']) for r in info['reads']]
        for reader in (load_fast5s._python_iter_reads, fast5_native.iter_reads):
            got = [(rid, len(sig), sha(sig)) for rid, sig in reader(path)]
            assert got == want, (name, reader.__module__)
        # partial reads through the native chunk indexes: the scanned ends only
        with fast5_native.File(path) as f:
            for k, (_, n, _) in enumerate(want[:3]):
                whole = f.read_signal(k, 0, n)
                assert sha(whole) == want[k][2]
                for first, count in ((0, min(n, 11)), (max(n - 11, 0), min(n, 11)), (n // 2, n // 4)):
                    assert np.array_equal(f.read_signal(k, first, count), whole[first:first + count])
    return len(expected)


def test_files_written_by_h5py():
    """Both on-disk generations (libver earliest / latest: layout messages v3 / v4 with single
    chunk, implicit, fixed array and extensible array chunk indexes), all storage kinds and
    filters, group sizes across the compact/dense boundary."""
    assert check_against_expected(VARIANT_DIR) >= 30


@pytest.mark.skipif(not os.path.exists(CONDA_PYTHON), reason='no interpreter with h5py here')
def test_files_written_by_h5py_just_now(tmp_path):
    """The full profile with another seed: adds 700-read containers, a 100k-sample read, chunk
    indexes with 2,000 .. 140,000 chunks (paged fixed arrays; extensible arrays with super blocks
    and paged data blocks) and sparsely written datasets."""
    script = os.path.join(REPO, 'oracle', 'make_h5py_fixtures.py')
    done = subprocess.run([CONDA_PYTHON, script, str(tmp_path), '4242', 'full'],
                          capture_output=True, text=True)
    if done.returncode != 0 and 'No module named' in done.stderr:
        pytest.skip('h5py is not importable: ' + done.stderr.strip().splitlines()[-1])
    assert done.returncode == 0, done.stderr
    assert check_against_expected(str(tmp_path)) >= 40


def test_the_reference_loader_run_on_every_fixture():
    """tests/golden/loader_reference.json is what the reference's own load_fast5s.py returned for
    every committed fast5 (oracle/make_loader_golden.py, run under h5py in the build container).
    The one case where this package deliberately differs: a variable-length-string read_id, which
    h5py 3 hands the reference as str and the reference then fails to .decode() (an uncaught
    AttributeError) - here the read loads."""
    with open(os.path.join(REPO, 'tests', 'golden', 'loader_reference.json')) as f:
        golden = json.load(f)
    seen = set()
    for rel, want in sorted(golden['files'].items()):
        path = os.path.join(REPO, 'tests', 'golden', 'fast5', rel)
        seen.add(want['result'])
        assert sorted(load_fast5s.get_root_level_keys(path)) == want['root_keys'], rel
        for reader in (load_fast5s._python_get_read_id_and_signal,
                       fast5_native.get_read_id_and_signal):
            if want['result'] == 'multi':
                with pytest.raises(SystemExit) as e:
                    reader(path)
                assert str(e.value) == want['message']
                continue
            read_id, signal = reader(path)
            if want['result'] == 'none':
                assert read_id is None and signal is None
            elif want['result'] == 'read':
                assert (read_id, len(signal), str(signal.dtype), sha(signal)) == \
                    (want['read_id'], want['n'], want['dtype'], want['sha256']), rel
            else:
                assert want['result'] == 'vlen' and len(read_id) == 36 and len(signal) > 0
    assert seen >= {'read', 'multi', 'vlen'}
    for sub, want in golden['directories'].items():
        folder = os.path.join(REPO, 'tests', 'golden', 'fast5', sub)
        files = sorted(os.path.join(folder, f) for f in os.listdir(folder) if f.endswith('.fast5'))
        assert sorted({load_fast5s.determine_single_or_multi_fast5s([f]) for f in files}) == \
            want['per_file']
        if want['first_five'].startswith('exit: '):
            with pytest.raises(SystemExit) as e:
                load_fast5s.determine_single_or_multi_fast5s(files[:5])
            assert str(e.value) == want['first_five'][6:]
        else:
            assert load_fast5s.determine_single_or_multi_fast5s(files[:5]) == want['first_five']


def test_both_inflaters_give_the_same_samples(monkeypatch):
    """Chunks are inflated by libdeflate where the system has it, by zlib otherwise or when
    DEEPBINNER_FAST5_INFLATE=zlib says so: same samples either way, file by file."""
    paths = single_files() + [os.path.join(VARIANT_DIR, n) for n in sorted(os.listdir(VARIANT_DIR))
                              if n.endswith('.fast5')]
    results = {}
    for mode in ('zlib', 'libdeflate'):
        monkeypatch.setenv('DEEPBINNER_FAST5_INFLATE', mode)
        batch = fast5_native.load_batch(single_files() * 3, 6656, 3)
        per_file = [fast5_native.load_reads(p, threads=2) for p in paths]
        results[mode] = (batch, per_file)
    a, b = results['zlib'], results['libdeflate']
    assert a[0][0] == b[0][0] and np.array_equal(a[0][1], b[0][1]) and np.array_equal(a[0][2], b[0][2])
    for x, y in zip(a[1], b[1]):
        assert x[0] == y[0] and np.array_equal(x[1], y[1]) and np.array_equal(x[3], y[3])


def test_a_filter_that_is_not_implemented_is_reported_as_such(tmp_path, capsys):
    """A deflate-compressed file whose filter id is patched to 32020 (ONT's VBZ) stands in for a
    VBZ file: both readers refuse it, the native one with a status of its own, and the classify
    loop says once why reads are being skipped."""
    import struct
    from deepbinner_amd import classify
    source = open(os.path.join(VARIANT_DIR, 'single_old_layout_old.fast5'), 'rb').read()
    # v1 filter pipeline entry: id 1, name of 8 bytes, flags optional, one client value
    entry = struct.pack('<HHHH', 1, 8, 1, 1) + b'deflate\x00'
    assert source.count(entry) == 1
    path = str(tmp_path / 'vbz_like.fast5')
    with open(path, 'wb') as f:
        f.write(source.replace(entry, struct.pack('<HHHH', 32020, 8, 1, 1) + b'vbz\x00\x00\x00\x00\x00'))
    assert fast5_native.get_read_id_and_signal(path) == (None, None)
    assert load_fast5s._python_get_read_id_and_signal(path) == (None, None)
    ids, samples, offsets, status = fast5_native.load_batch([path, single_files()[0]], 6656, 2)
    assert list(status) == [fast5_native.F5_ERR_FILTER, 0] and ids[0] is None and ids[1]
    assert offsets[1] == 0 and offsets[2] == len(samples)
    assert fast5_native.load_reads(path)[3][0] == fast5_native.F5_ERR_FILTER
    classify._FILTER_WARNING_GIVEN = False
    classify.warn_about_filters(status)
    classify.warn_about_filters(status)
    assert capsys.readouterr().err.count('cannot decode (VBZ?)') == 1


def test_unreadable_files(tmp_path):
    assert fast5_native.get_read_id_and_signal(str(tmp_path / 'missing.fast5')) == (None, None)
    empty = tmp_path / 'empty.fast5'
    empty.write_bytes(b'')
    assert fast5_native.get_read_id_and_signal(str(empty)) == (None, None)
    garbage = tmp_path / 'garbage.fast5'
    garbage.write_bytes(os.urandom(4096))
    assert fast5_native.get_read_id_and_signal(str(garbage)) == (None, None)
    # a real file cut short and a real file with its middle overwritten: an error, never a crash
    src = open(single_files()[0], 'rb').read()
    for k, data in enumerate((src[:len(src) // 2], src[:3000] + os.urandom(len(src) - 3000),
                              src[:600] + b'\xff' * 64 + src[664:])):
        p = tmp_path / ('damaged%d.fast5' % k)
        p.write_bytes(data)
        rid, sig = fast5_native.get_read_id_and_signal(str(p))
        want_id, want_sig = load_fast5s._python_get_read_id_and_signal(str(p))
        if rid is not None and want_id is not None:
            assert rid == want_id and np.array_equal(sig, want_sig)


@pytest.mark.parametrize('keep,threads', [(None, 1), (6656, 3), (1000, 0)])
def test_load_batch(tmp_path, keep, threads):
    files = single_files()
    bad = str(tmp_path / 'garbage.fast5')
    open(bad, 'wb').write(b'not hdf5')
    multi = multi_files()[0]
    paths = files + [bad, str(tmp_path / 'missing.fast5'), multi] + files[::-1]
    read_ids, samples, offsets, status = fast5_native.load_batch(paths, keep, threads)
    assert len(read_ids) == len(paths) and offsets[0] == 0 and offsets[-1] == len(samples)
    for i, path in enumerate(paths):
        got = samples[offsets[i]:offsets[i + 1]]
        if path in files:
            want_id, want = load_fast5s._python_get_read_id_and_signal(path)
            assert status[i] == 0 and read_ids[i] == want_id
            assert np.array_equal(got, load_fast5s.keep_ends(want, keep))
        else:
            assert status[i] != 0 and read_ids[i] is None and len(got) == 0
    assert status[paths.index(multi)] == fast5_native.F5_ERR_MULTI
    assert fast5_native.load_batch([], keep, threads)[0] == []


@pytest.mark.parametrize('path', multi_files() + single_files()[:2])
def test_load_reads_of_one_file(path):
    """f5_load_reads: every read of a (multi-read) file through native threads equals the Python
    reader's, whole or cut to the scanned ends, any sub-range, any thread count."""
    want = python_reads(path)
    for keep, threads in ((None, 1), (1000, 4), (6656, 0)):
        ids, samples, offsets, status = fast5_native.load_reads(path, keep=keep, threads=threads)
        assert ids == [w[0] for w in want] and (status == 0).all()
        for i, (_, signal) in enumerate(want):
            assert np.array_equal(samples[offsets[i]:offsets[i + 1]],
                                  load_fast5s.keep_ends(signal, keep))
    if len(want) > 6:
        ids, samples, offsets, _ = fast5_native.load_reads(path, first=3, count=4, threads=2)
        assert ids == [w[0] for w in want[3:7]]
        assert np.array_equal(samples[offsets[1]:offsets[2]], want[4][1])
    assert fast5_native.load_reads(path, first=len(want), count=0)[0] == []
    with pytest.raises(OSError):
        fast5_native.load_reads(path, first=len(want), count=1)
    with pytest.raises(OSError):
        fast5_native.load_reads(path + '.missing')


def test_stream_of_containers(tmp_path):
    """f5_stream_*: containers loaded several at once by one thread team come out in path order
    with exactly what f5_load_reads gives for each - for every window depth and team size, with
    one-read files, an unreadable file and a missing one among the containers, and with the same
    container many times over (one chunk cache per thread serving several files)."""
    bad = tmp_path / 'not_hdf5.fast5'
    bad.write_bytes(b'nothing of the kind' * 100)
    paths = (multi_files() + [str(bad)] + single_files()[:2] + [str(tmp_path / 'missing.fast5')] +
             multi_files()[::-1] * 3)
    want = {}
    for p in set(paths):
        try:
            want[p] = {keep: fast5_native.load_reads(p, keep=keep, threads=1)
                       for keep in (None, 700, 6656)}
        except OSError:
            want[p] = None
    for keep, threads, depth in ((6656, 5, 3), (None, 2, 1), (700, 16, 8), (6656, 1, 2)):
        seen = []
        for index, ids, samples, offsets, status in fast5_native.stream_reads(
                paths, keep=keep, threads=threads, depth=depth):
            seen.append(index)
            w = want[paths[index]]
            if w is None:
                assert ids is None and status != 0
                continue
            w_ids, w_samples, w_offsets, w_status = w[keep]
            assert ids == w_ids and np.array_equal(offsets, w_offsets)
            assert np.array_equal(samples, w_samples) and np.array_equal(status, w_status)
        assert seen == list(range(len(paths)))
    assert list(fast5_native.stream_reads([], keep=6656)) == []
    # a consumer that stops early: closing the generator stops the team
    stream = fast5_native.stream_reads(paths, keep=6656, threads=4, depth=4)
    assert next(stream)[0] == 0
    stream.close()


def decode_raw_batch(offsets, comp, records):
    """What a decoder makes of a raw batch: every piece inflated (or copied) to its place."""
    import zlib
    out = np.zeros(int(offsets[-1]) * 2, dtype=np.uint8)
    for r in records:
        data = bytes(comp[int(r['comp_offset']):int(r['comp_offset']) + int(r['comp_bytes'])])
        if r['mode'] == fast5_native.RAW_ZLIB:
            data = zlib.decompress(data)
        want = int(r['out_bytes'])
        assert len(data) >= want
        out[int(r['out_offset']):int(r['out_offset']) + want] = np.frombuffer(data, np.uint8, want)
    return out.view('<i2')


def test_raw_stream_of_containers(tmp_path):
    """f5_stream_open_raw: the Signal as stored, fetched in file order by runs of neighbouring
    chunks (Fast5::read_many) - decoded on the spot with zlib it is what f5_load_reads gives, for
    every team size, window depth and host-inflate policy, with unreadable files among the
    containers and the same container many times over; the pieces of a batch cover every good
    read's range exactly once, longest deflate stream first."""
    bad = tmp_path / 'not_hdf5.fast5'
    bad.write_bytes(b'nothing of the kind' * 100)
    paths = (multi_files() + [str(bad)] + single_files()[:1] + [str(tmp_path / 'missing.fast5')] +
             multi_files()[::-1] * 2)
    want = {}
    for p in set(paths):
        try:
            want[p] = fast5_native.load_reads(p, threads=1)
        except OSError:
            want[p] = None
    for threads, depth, host_above in ((1, 1, 0), (4, 3, 0), (16, 8, 4096), (3, 2, -50), (2, 2, 1)):
        seen = []
        for index, ids, offsets, status, comp, records in fast5_native.stream_raw(
                paths, threads=threads, depth=depth, host_inflate_above=host_above):
            seen.append(index)
            w = want[paths[index]]
            if w is None:
                assert ids is None and status != 0
                continue
            w_ids, w_samples, w_offsets, w_status = w
            assert ids == w_ids and np.array_equal(offsets, w_offsets)
            assert np.array_equal(status, w_status)
            assert np.array_equal(decode_raw_batch(offsets, comp, records), w_samples)
            covered = np.zeros(len(ids), dtype=np.int64)
            np.add.at(covered, records['read'], records['out_bytes'])
            good = np.asarray(status) == 0
            assert np.array_equal(covered[good], 2 * np.diff(offsets)[good])
            z = records[records['mode'] == fast5_native.RAW_ZLIB]['comp_bytes']
            assert (np.diff(z) <= 0).all()
            if host_above > 0:
                assert (z <= host_above).all()
            assert int((records['comp_offset'] + records['comp_bytes']).max(initial=0)) <= len(comp)
        assert seen == list(range(len(paths)))
    assert list(fast5_native.stream_raw([])) == []


def test_the_host_keeps_no_share_of_a_container_of_long_streams(tmp_path):
    """host_inflate_above = -p: the host inflates the longest streams holding p per cent of an
    ordinary container's bytes - and none of a container whose streams average more than 64 KiB
    (long reads throughout are left to the GPU's decoder: profiles/r06_loader)."""
    import uuid
    import zlib
    from deepbinner_amd import hdf5_write
    rng = np.random.default_rng(11)

    def container(path, n_reads, samples):
        reads = []
        for k in range(n_reads):
            n = samples + 100 * k
            levels = np.repeat(rng.normal(450, 80, n // 8 + 1), 8)[:n]
            sig = np.clip(np.rint(levels + rng.normal(0, 8, n)), 0, 2047).astype(np.int16)
            reads.append((str(uuid.UUID(int=k + 1)), sig, None, zlib.compress(sig.tobytes(), 1)))
        with open(path, 'wb') as f:
            f.write(hdf5_write.multi_read_fast5_bytes(reads))

    short, long_ = str(tmp_path / 'short.fast5'), str(tmp_path / 'long.fast5')
    container(short, 12, 9000)
    container(long_, 6, 70000)
    kept = {}
    for index, ids, offsets, status, comp, records in fast5_native.stream_raw(
            [short, long_], threads=2, host_inflate_above=-50):
        assert (np.asarray(status) == 0).all()
        kept[index] = int((records['mode'] == fast5_native.RAW_STORED).sum())
        want = fast5_native.load_reads([short, long_][index], threads=1)
        assert np.array_equal(decode_raw_batch(offsets, comp, records), want[1])
        z = records[records['mode'] == fast5_native.RAW_ZLIB]['comp_bytes']
        assert index == 0 or z.mean() > 64 * 1024
    assert kept[0] > 0 and kept[1] == 0


def test_sample_buffers_are_recycled_and_can_come_from_the_caller():
    """The packed samples of a batch come from a pool of recycled buffers, or from an allocator
    the caller installs (pinned host memory on a GPU box; here: counted malloc)."""
    import ctypes
    path = multi_files()[0]
    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    live, sizes = set(), []
    ALLOC = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
    FREE = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p)

    def alloc(nbytes, user):
        ptr = libc.malloc(nbytes)
        live.add(ptr)
        sizes.append(nbytes)
        return ptr

    def release(ptr, user):
        live.discard(ptr)
        libc.free(ptr)

    alloc_c, release_c = ALLOC(alloc), FREE(release)
    want = fast5_native.load_reads(path, keep=6656, threads=2)
    fast5_native.set_sample_allocator(ctypes.cast(alloc_c, ctypes.c_void_p).value,
                                      ctypes.cast(release_c, ctypes.c_void_p).value)
    try:
        for _ in range(5):
            ids, samples, offsets, status = fast5_native.load_reads(path, keep=6656, threads=2)
            assert ids == want[0] and np.array_equal(samples, want[1])
            address = samples.ctypes.data
            assert address in live                      # the batch lies in the caller's memory
            del samples
        assert len(sizes) == 1                          # one allocation served all five batches
        held = fast5_native.load_reads(path, keep=6656, threads=2)
        assert held[1].ctypes.data == address
    finally:
        fast5_native.set_sample_allocator(None, None)   # flushes the pool of idle buffers ...
    assert live == {address}                            # ... but not a batch still alive
    assert np.array_equal(held[1], want[1])
    del held
    assert live == set()                                # freed through the allocator it came from
    again = fast5_native.load_reads(path, keep=6656, threads=2)
    assert np.array_equal(again[1], want[1]) and len(sizes) == 1


POOL_EVICTION = r"""
import ctypes, sys
import numpy as np
from deepbinner_amd import fast5_native
path = sys.argv[1]
libc = ctypes.CDLL(None)
libc.malloc.restype = ctypes.c_void_p
libc.malloc.argtypes = [ctypes.c_size_t]
libc.free.argtypes = [ctypes.c_void_p]
sizes, freed = [], []
ALLOC = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
FREE = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p)
def alloc(nbytes, user):
    sizes.append(nbytes)
    return libc.malloc(nbytes)
def release(ptr, user):
    freed.append(ptr)
    libc.free(ptr)
a, r = ALLOC(alloc), FREE(release)
fast5_native.set_sample_allocator(ctypes.cast(a, ctypes.c_void_p).value,
                                  ctypes.cast(r, ctypes.c_void_p).value)
small = [fast5_native.load_reads(path, keep=64, threads=1) for _ in range(3)]   # three small batches
assert len(sizes) == 3 and len(set(sizes)) == 1
small_block = sizes[0]
del small                                  # ... now idle in the pool: it is full of them
whole = fast5_native.load_reads(path, threads=1)
assert len(sizes) == 4 and sizes[3] > 4 * small_block       # none of them fits: a fresh block
want = np.array(whole[1])
del whole                                  # it goes back, the oldest small ones make room
assert len(freed) >= 1
for _ in range(5):
    whole = fast5_native.load_reads(path, threads=1)
    assert np.array_equal(whole[1], want)
    del whole
assert len(sizes) == 4, sizes              # ... and serves every batch of the new size
fast5_native.set_sample_allocator(None, None)
print('ok', small_block, sizes[3], len(freed))
"""


def test_the_buffer_pool_makes_room_for_a_new_size(tmp_path):
    """The pool of idle sample buffers is bounded (DEEPBINNER_FAST5_POOL_MB); full of one
    workload's sizes it must still take in the blocks of the next workload - the blocks that
    waited longest go - instead of allocating and freeing (pinned memory: slowly, and waiting
    for the GPU) for every batch from then on."""
    import subprocess
    import sys
    lengths = [200000] * 12
    from deepbinner_amd import hdf5_write
    rng = np.random.default_rng(3)
    reads = [('read-%02d' % i, rng.integers(0, 2000, n).astype(np.int16))
             for i, n in enumerate(lengths)]
    path = tmp_path / 'twelve.fast5'
    path.write_bytes(hdf5_write.multi_read_fast5_bytes(reads))
    env = dict(os.environ, DEEPBINNER_FAST5_POOL_MB='8',
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, '-c', POOL_EVICTION, str(path)], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith('ok')


def test_many_copies_in_parallel(tmp_path):
    """Thread-safety smoke test: 400 files on 16 threads give what one thread gives."""
    files = single_files()
    paths = []
    for i in range(400):
        dst = tmp_path / ('copy_%03d.fast5' % i)
        shutil.copyfile(files[i % len(files)], dst)
        paths.append(str(dst))
    a = fast5_native.load_batch(paths, 6656, 16)
    b = fast5_native.load_batch(paths, 6656, 1)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert (a[3] == 0).all()


def test_mutated_files_never_crash_and_agree_with_python_reader(tmp_path):
    """800 seeded mutations (byte flips, zeroed runs, truncations) of real files: the native
    reader must return what the Python reader returns - the same data or the same refusal."""
    rng = np.random.default_rng(20260927)
    variants = [os.path.join(VARIANT_DIR, name + '.fast5') for name in (
        'chunks_150_unlimited_new', 'chunks_40_fixed_new', 'chunks_sparse_new', 'multi_12_new',
        'many_chunks_shuffle_fletcher_new', 'compact_new')]
    sources = [open(f, 'rb').read() for f in single_files()[:3] + multi_files()[:1] + variants]
    path = str(tmp_path / 'mutant.fast5')
    outcomes = {'same data': 0, 'both refuse': 0}
    for trial in range(800):
        data = bytearray(sources[trial % len(sources)])
        kind = trial % 4
        if kind == 0:       # a few random byte flips, mostly in the metadata-heavy first 8 KB
            for _ in range(int(rng.integers(1, 6))):
                data[int(rng.integers(0, min(len(data), 8192)))] ^= int(rng.integers(1, 256))
        elif kind == 1:     # flips anywhere
            for _ in range(int(rng.integers(1, 20))):
                data[int(rng.integers(0, len(data)))] ^= int(rng.integers(1, 256))
        elif kind == 2:     # a zeroed run
            a = int(rng.integers(0, len(data) - 64))
            data[a:a + int(rng.integers(8, 64))] = bytes(64)[:int(rng.integers(8, 64))]
        else:               # truncation
            data = data[:int(rng.integers(16, len(data)))]
        with open(path, 'wb') as f:
            f.write(data)
        try:
            want = list(load_fast5s._python_iter_reads(path))
        except Exception:       # the Python reader may trip over damage it does not expect
            want = None
        got = list(fast5_native.iter_reads(path))
        try:                                    # the threaded batch entry must not crash either
            fast5_native.load_reads(path, keep=1000, threads=3)
        except OSError:
            pass
        # ... nor the raw stream (coalesced fetch over whatever addresses the damage left): its
        # pieces stay inside the buffers it hands over
        for _, ids, offsets, status, comp, records in fast5_native.stream_raw([path], threads=2):
            if ids is None:
                continue
            assert (records['comp_offset'] >= 0).all() and (records['out_offset'] >= 0).all()
            assert int((records['comp_offset'] + records['comp_bytes']).max(initial=0)) <= len(comp)
            assert int((records['out_offset'] + records['out_bytes']).max(initial=0)) <= 2 * int(offsets[-1])
        if want is None or not want:
            # the native reader may still have salvaged reads the generator gave up on midway;
            # what matters here is that it returned at all
            outcomes['both refuse'] += 1
            continue
        if len(got) == len(want) and all(a[0] == b[0] and np.array_equal(a[1], b[1])
                                         for a, b in zip(got, want)):
            outcomes['same data'] += 1
        else:
            # partial agreement is acceptable only as a prefix (a generator that stops at the
            # first damaged read): every read both produced must be identical
            for a, b in zip(got, want):
                assert a[0] == b[0] and np.array_equal(a[1], b[1]), trial
    assert outcomes['same data'] > 50 and outcomes['both refuse'] > 50, outcomes


"""The HDF5 writer (deepbinner_amd/hdf5_write.py) that bins the reads of multi-read containers
into one-read fast5 files.  Pinned to the real HDF5 library and to the reference's own loader:
tests/golden/writer_reference.json holds what h5py and the reference's load_fast5s.py read from the
files of cases() (oracle/make_writer_golden.py, build container); here both of this package's
readers must read the same, and - wherever the image's h5py interpreter exists - h5py is asked
again, live."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD

CONDA_PYTHON = '/opt/conda/bin/python3.9'


def cases():
    """(name, read_id, int16 signal, compress) - seeded, so the generator and the tests agree."""
    rng = np.random.default_rng(20260928)
    def squiggle(n):
        levels = np.repeat(rng.normal(450, 80, n // 8 + 1), 8)[:n]
        return np.clip(np.rint(levels + rng.normal(0, 8, n)), 0, 2047).astype(np.int16)
    uid = lambda k: '%08x-0000-4000-8000-%012x' % (k, k * 7919)      # noqa: E731
    return [
        ('typical', uid(1), squiggle(27553), True),
        ('typical_contiguous', uid(2), squiggle(27553), False),
        ('short', uid(3), squiggle(9), True),
        ('one_sample', uid(4), np.array([-5], dtype=np.int16), True),
        ('empty', uid(5), np.zeros(0, dtype=np.int16), True),
        ('long', uid(6), squiggle(400001), True),
        ('full_range', uid(7), rng.integers(-32768, 32768, 5000).astype(np.int16), True),
        ('constant', uid(8), np.full(70000, 512, dtype=np.int16), True),
    ]


def sha(signal):
    return hashlib.sha256(np.asarray(signal).astype('<i2').tobytes()).hexdigest()


@pytest.fixture(scope='module')
def written(tmp_path_factory):
    from deepbinner_amd import hdf5_write
    root = tmp_path_factory.mktemp('written')
    out = {}
    for name, read_id, signal, compress in cases():
        path = str(root / (name + '.fast5'))
        hdf5_write.write_single_read_fast5(path, read_id, signal, compress=compress)
        out[name] = (path, read_id, signal)
    return out


def test_golden_is_what_the_reference_read(written):
    with open(os.path.join(GOLD, 'writer_reference.json')) as f:
        gold = json.load(f)
    assert gold['read_back']['determine_single_or_multi_fast5s'] == 'single'
    for name, (path, read_id, signal) in written.items():
        g = gold['read_back'][name + '.fast5']
        assert g['reference_loader']['read_id'] == g['h5py']['read_id'] == read_id
        assert g['reference_loader']['sha256'] == g['h5py']['sha256'] == sha(signal)
        assert g['reference_loader']['length'] == len(signal)
        assert g['h5py']['dtype'] == 'int16' and g['h5py']['keys'] == ['read_' + read_id]
        assert g['reference_loader']['root_keys'] == ['read_' + read_id]


@pytest.mark.parametrize('reader', ['python', 'native'])
def test_own_readers_read_what_was_written(written, reader, monkeypatch):
    from deepbinner_amd import load_fast5s
    monkeypatch.setenv('DEEPBINNER_FAST5_READER', reader)
    for name, (path, read_id, signal) in written.items():
        got_id, got = load_fast5s.get_read_id_and_signal(path)
        assert got_id == read_id, name
        assert got.dtype == np.int16 and np.array_equal(got, signal), name
        assert load_fast5s.get_root_level_keys(path) == ['read_' + read_id]
    assert load_fast5s.determine_single_or_multi_fast5s([p for p, _, _ in written.values()]) == 'single'


@pytest.mark.skipif(not os.path.exists(CONDA_PYTHON), reason='no interpreter with h5py in this image')
def test_the_real_hdf5_library_reads_them(written):
    code = ('import h5py, hashlib, json, sys\n'
            'out = {}\n'
            'for p in sys.argv[1:]:\n'
            '    with h5py.File(p, "r") as f:\n'
            '        k = list(f.keys()); raw = f[k[0] + "/Raw"]; s = raw["Signal"]\n'
            '        out[p] = [k, raw.attrs["read_id"].decode(), int(raw.attrs["duration"]), '
            'str(s.dtype), s.compression, hashlib.sha256(s[:].astype("<i2").tobytes()).hexdigest()]\n'
            'print(json.dumps(out))\n')
    paths = [p for p, _, _ in written.values()]
    got = json.loads(subprocess.check_output([CONDA_PYTHON, '-c', code] + paths))
    for name, (path, read_id, signal) in written.items():
        keys, rid, duration, dtype, compression, digest = got[path]
        assert keys == ['read_' + read_id] and rid == read_id and duration == len(signal)
        assert dtype == 'int16' and digest == sha(signal)
        assert compression == ('gzip' if name not in ('typical_contiguous', 'empty') else None)


def test_realtime_bins_the_reads_of_multi_read_files(oracle_backend, tmp_path, capsys, monkeypatch):
    """No multi_to_single_fast5: every read of the containers becomes a one-read fast5 in the bin
    of its call, with its whole signal, and is also listed in the table."""
    import argparse
    import shutil
    from conftest import MODEL_DIR
    from deepbinner_amd import hdf5_lite, load_fast5s
    import deepbinner_amd.realtime as realtime
    monkeypatch.setattr(realtime, 'POLL_SECONDS', 0)
    monkeypatch.setattr(shutil, 'which', lambda name: None)
    in_dir, out_dir = tmp_path / 'in', tmp_path / 'out'
    shutil.copytree(os.path.join(GOLD, 'fast5', 'multi'), in_dir)
    originals = {}
    for name in sorted(os.listdir(in_dir)):
        for read_id, signal in load_fast5s.iter_reads(str(in_dir / name)):
            originals[read_id] = np.asarray(signal)
    args = argparse.Namespace(in_dir=str(in_dir), out_dir=str(out_dir), stop=True,
                              start_model=os.path.join(MODEL_DIR, 'SQK-RBK004_read_starts.dbw'),
                              end_model=None, scan_size=6144.0, score_diff=0.5, batch_size=16,
                              require_either=False, require_start=False, require_both=False)
    realtime.realtime(args)
    assert 'Wrote 30 one-read fast5 files' in capsys.readouterr().out
    rows = [r.split('\t') for r in
            open(out_dir / 'multi_read_classifications.tsv').read().splitlines()]
    assert len(rows) == 30 == len(originals)
    seen = set()
    for read_id, call, _ in rows:
        path = out_dir / realtime.bin_name(call) / (read_id + '.fast5')
        assert path.is_file()
        got_id, got = load_fast5s.get_read_id_and_signal(str(path))
        assert got_id == read_id and np.array_equal(got, originals[read_id])
        with hdf5_lite.File(str(path), 'r') as f:
            assert list(f.keys()) == ['read_' + read_id]
            # what multi_to_single_fast5 carries over besides the signal (basecallers need
            # channel_id): every attribute of the read's groups, same names, types and values
            source = [n for n in sorted(os.listdir(in_dir))
                      if 'read_' + read_id in hdf5_lite.File(str(in_dir / n), 'r')][0]
            with hdf5_lite.File(str(in_dir / source), 'r') as container:
                want_group, got_group = container['read_' + read_id], f['read_' + read_id]
                assert sorted(got_group.keys()) == sorted(want_group.keys()) == \
                    ['Raw', 'channel_id', 'context_tags', 'tracking_id']
                for sub in (None, 'Raw', 'channel_id', 'context_tags', 'tracking_id'):
                    want_attrs = dict((want_group[sub] if sub else want_group).attrs.items())
                    got_attrs = dict((got_group[sub] if sub else got_group).attrs.items())
                    assert sorted(got_attrs) == sorted(want_attrs) and len(want_attrs) > 0
                    for key, value in want_attrs.items():
                        same = got_attrs[key] == value or (value != value and
                                                           got_attrs[key] != got_attrs[key])
                        assert same, (sub, key)            # (median_before is NaN in places)
                        assert getattr(got_attrs[key], 'dtype', None) == \
                            getattr(value, 'dtype', None), (sub, key)
        seen.add(read_id)
    assert seen == set(originals)
    n_files = sum(len(files) for _, _, files in os.walk(out_dir)) - 1      # minus the table
    assert n_files == 30
    if os.path.exists(CONDA_PYTHON):
        # the real HDF5 library: every binned file holds its container's groups, attributes
        # (names, types, values) and signal
        code = (
            'import glob, sys\nimport h5py, numpy as np\n'
            'containers = [h5py.File(c, "r") for c in glob.glob(sys.argv[2] + "/*.fast5")]\n'
            'n = 0\n'
            'for p in glob.glob(sys.argv[1] + "/*/*.fast5"):\n'
            '    f = h5py.File(p, "r"); k = list(f.keys())[0]; g = f[k]\n'
            '    src = [h[k] for h in containers if k in h][0]\n'
            '    assert sorted(g.keys()) == sorted(src.keys())\n'
            '    for sub in (".", "Raw", "channel_id", "tracking_id", "context_tags"):\n'
            '        a = dict(g.attrs if sub == "." else g[sub].attrs)\n'
            '        b = dict(src.attrs if sub == "." else src[sub].attrs)\n'
            '        assert sorted(a) == sorted(b), sub\n'
            '        for key in a:\n'
            '            assert a[key] == b[key] or a[key] != a[key], key\n'
            '            assert getattr(a[key], "dtype", None) == getattr(b[key], "dtype", None)\n'
            '    assert np.array_equal(g["Raw/Signal"][()], src["Raw/Signal"][()])\n'
            '    n += 1\n'
            'print(n)\n')
        import subprocess
        done = subprocess.run([CONDA_PYTHON, '-c', code, str(out_dir), str(in_dir)],
                              capture_output=True, text=True)
        assert done.returncode == 0 and done.stdout.strip() == '30', done.stderr[-2000:]


def test_multi_read_containers_are_written_too(tmp_path):
    """hdf5_write.multi_read_fast5_bytes (the containers of the streaming tests and tools): 300
    reads under one root group; both readers - and the real HDF5 library where the image has it -
    find every read, its signal and its attributes."""
    import uuid
    from deepbinner_amd import fast5_native, hdf5_lite, hdf5_write, load_fast5s
    rng = np.random.default_rng(5)
    reads = []
    for k in range(300):
        signal = rng.integers(-300, 2047, int(rng.integers(0, 5000))).astype(np.int16)
        meta = {'Raw': {'read_number': np.int32(k), 'start_mux': np.uint8(k % 4)},
                'channel_id': {'digitisation': np.float64(8192.0), 'channel_number': b'%d' % k}}
        reads.append((str(uuid.UUID(bytes=rng.bytes(16), version=4)), signal, meta))
    path = str(tmp_path / 'container.fast5')
    with open(path, 'wb') as f:
        f.write(hdf5_write.multi_read_fast5_bytes(reads))
    want = sorted(reads, key=lambda r: r[0])
    assert load_fast5s.determine_single_or_multi_fast5s([path]) == 'multi'
    if fast5_native.available():
        ids, samples, offsets, status = fast5_native.load_reads(path, threads=3)
        assert ids == [r[0] for r in want] and (status == 0).all()
        for i, r in enumerate(want):
            assert np.array_equal(samples[offsets[i]:offsets[i + 1]], r[1])
    with hdf5_lite.File(path, 'r') as f:
        assert list(f.keys()) == ['read_' + r[0] for r in want]
        for read_id, signal, meta in want[::17]:
            group = f['read_' + read_id]
            assert np.array_equal(np.asarray(group['Raw/Signal']), signal)
            assert group['Raw'].attrs['read_number'] == meta['Raw']['read_number']
            assert group['channel_id'].attrs['channel_number'] == meta['channel_id']['channel_number']
    with open(str(tmp_path / 'empty.fast5'), 'wb') as f:
        f.write(hdf5_write.multi_read_fast5_bytes([]))
    with hdf5_lite.File(str(tmp_path / 'empty.fast5'), 'r') as f:
        assert list(f.keys()) == []
    if os.path.exists(CONDA_PYTHON):
        import subprocess
        code = ('import h5py, sys\n'
                'f = h5py.File(sys.argv[1], "r")\n'
                'print(len(f), sum(int(f[k]["Raw/Signal"][()].astype("i8").sum()) for k in f),'
                ' sum(int(f[k]["Raw"].attrs["read_number"]) for k in f))\n')
        done = subprocess.run([CONDA_PYTHON, '-c', code, path], capture_output=True, text=True)
        assert done.returncode == 0, done.stderr[-2000:]
        assert done.stdout.split() == ['300', str(sum(int(r[1].astype('i8').sum()) for r in reads)),
                                       str(sum(range(300)))]

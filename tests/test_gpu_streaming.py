"""
BASELINE.json configs[4] on the GPU (`-m gpu`): multi-read containers streamed through the native
loader's thread team (f5_stream_*), pinned batches, both models in one call of the C ABI
(dbh_classify_pair_i16), the dispatcher's device queues - at the configuration's own scale
(>= 100,000 reads), with the reference's own calls (tests/golden/calls.json: its call_batch +
combine_calls on the 37 fixture reads) as the anchor and bit-exact invariance everywhere else.
"""
import argparse
import os
import shutil
import uuid
import zlib

import numpy as np
import pytest

from conftest import GOLD, MODEL_DIR
from oracle import classify_ref

pytestmark = pytest.mark.gpu
START, END = 'EXP-NBD103_read_starts', 'EXP-NBD103_read_ends'


def pack(signals):
    offsets = np.zeros(len(signals) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(s) for s in signals])
    samples = (np.concatenate(signals) if len(signals) and offsets[-1] else
               np.zeros(0)).astype(np.int16)
    return samples, offsets


def names(calls):
    return ['none' if c == 0 else str(int(c)) for c in calls]


def reference_final_calls(gold, mode='require_either'):
    """{read id: final call} of the 37 fixture reads: the reference's own call_batch per side
    (calls.json) through its combine_calls rule (oracle restatement, pinned to its truth table)."""
    ids = gold['read_ids'] + gold['multi_read_ids']
    starts = gold['calls'][START + '/start']
    ends = gold['calls'][END + '/end']
    return {rid: classify_ref.combine_calls(s, e, mode) for rid, s, e in zip(ids, starts, ends)}


@pytest.mark.parametrize('mode', ['require_either', 'require_start', 'require_both'])
def test_pair_entry_point_is_the_two_models_and_combine_calls(hip, hip_models, gold, all_signals,
                                                              mode):
    """dbh_classify_pair_i16 on ragged real reads: final calls = the reference's (calls.json),
    per-side calls and probabilities bit-identical to each model's own dbh_classify_i16, for
    pinned and pageable buffers, any group size, either model alone, empty and tiny reads."""
    start, end = hip_models[START], hip_models[END]
    signals = list(all_signals) + [np.zeros(0, np.int16), np.arange(5, dtype=np.int16),
                                   np.full(300, 7, np.int16)]
    samples, offsets = pack(signals)
    want = reference_final_calls(gold, mode)
    ids = gold['read_ids'] + gold['multi_read_ids']
    s_probs, s_calls = start.classify_packed(samples, offsets, 'start', 6144, 0.5)
    e_probs, e_calls = end.classify_packed(samples, offsets, 'end', 6144, 0.5)
    expected = [classify_ref.combine_calls(s, e, mode)
                for s, e in zip(names(s_calls), names(e_calls))]
    assert expected[:37] == [want[rid] for rid in ids]

    def check(calls, sides, probs):
        assert names(calls) == expected
        assert np.array_equal(sides[0], s_calls) and np.array_equal(sides[1], e_calls)
        assert np.array_equal(probs[0], s_probs) and np.array_equal(probs[1], e_probs)

    check(*hip.classify_pair(start, end, samples, offsets, 6144, 0.5, mode, True, True))
    assert names(hip.classify_pair(start, end, samples, offsets, 6144, 0.5, mode)) == expected
    # the same buffer in pinned memory: read by the DMA engine in place
    lib = hip.load_library()
    import ctypes
    ptr = ctypes.c_void_p()
    hip.check(lib.dbh_malloc_host(ctypes.byref(ptr), samples.nbytes))
    try:
        pinned = np.ctypeslib.as_array(ctypes.cast(ptr.value, ctypes.POINTER(ctypes.c_int16)),
                                       shape=samples.shape)
        pinned[:] = samples
        assert hip.is_pinned(pinned) and not hip.is_pinned(samples)
        check(*hip.classify_pair(start, end, pinned, offsets, 6144, 0.5, mode, True, True))
        for windows in (12, 13 * 12, 5000):        # 1 read, 13 reads, all reads per group
            start.set_host_group(windows)
            end.set_host_group(windows)
            check(*hip.classify_pair(start, end, pinned, offsets, 6144, 0.5, mode, True, True))
            check(*hip.classify_pair(start, end, samples, offsets, 6144, 0.5, mode, True, True))
    finally:
        start.set_host_group(0)
        end.set_host_group(0)
        hip.check(lib.dbh_free_host(ptr))
    # one model alone
    assert np.array_equal(hip.classify_pair(start, None, samples, offsets, 6144, 0.5, mode), s_calls)
    only_end = hip.classify_pair(None, end, samples, offsets, 6144, 0.5, mode, True, True)
    assert np.array_equal(only_end[0], e_calls) and np.array_equal(only_end[1][1], e_calls)
    assert only_end[1][0] is None and np.array_equal(only_end[2][1], e_probs)
    # nothing to do / bad arguments
    assert len(hip.classify_pair(start, end, np.zeros(0, np.int16), np.zeros(1, np.int64), 6144,
                                 0.5, mode)) == 0
    with pytest.raises(ValueError):
        hip.classify_pair(None, None, samples, offsets, 6144, 0.5, mode)
    other = hip_models['SQK-RBK004_read_starts']
    if other.n_classes != start.n_classes:
        with pytest.raises(hip.HipBackendError):
            hip.classify_pair(start, other, samples, offsets, 6144, 0.5, mode)


def test_host_path_at_one_window_per_read_is_the_device_path(hip, hip_models):
    """dbh_classify_i16 over 70,000 one-window reads (three groups through the three slots) equals
    the device-resident entry point bit for bit, pinned or pageable."""
    from bench import config_reads
    reads = config_reads(70000, 99)
    model = hip_models[START]
    n = len(reads)
    offsets = np.arange(n + 1, dtype=np.int64) * 1024
    d_samples = hip.DeviceBuffer.from_array(reads)
    d_offsets = hip.DeviceBuffer.from_array(offsets)
    d_probs = hip.DeviceBuffer(n * 13 * 4)
    d_calls = hip.DeviceBuffer(n * 4)
    model.classify_batched_dev(d_samples.ptr, d_offsets.ptr, n, 256, 'start', 512, 0.5,
                               d_probs.ptr, d_calls.ptr, None)
    hip.synchronize()
    want_probs = d_probs.download((n, 13), np.float32)
    want_calls = d_calls.download((n,), np.int32)
    probs, calls = model.classify_packed(reads.reshape(-1), offsets, 'start', 512, 0.5)
    assert np.array_equal(calls, want_calls) and np.array_equal(probs, want_probs)
    assert (calls != 0).sum() > 50


# ---- the stream at scale ------------------------------------------------------------------------
# BASELINE.json configs[4] streams 1 M reads over 8 GPUs: one GPU's share is 125,000.  32 containers
# of 4,000 reads (+ the 30 fixture reads) = 128,030 >= that share; DEEPBINNER_STREAM_READS=1000000
# runs the same tests over the whole configuration on the one GPU (250 containers, ~7 GB of
# scratch files, opt-in).
READS_PER_CONTAINER = 4000
N_CONTAINERS = max(1, -(-int(os.environ.get('DEEPBINNER_STREAM_READS', '128000')) // READS_PER_CONTAINER))


def build_containers(directory, gold, n_containers=None):
    """N_CONTAINERS containers of 4,000 reads (+ the 30 multi-read fixture reads dealt over them), written
    with this package's own container writer: 2,000 distinct seeded squiggles of 2,000-9,000
    samples, deflated once each, under fresh read ids."""
    from concurrent.futures import ThreadPoolExecutor
    from deepbinner_amd import hdf5_write
    N_CONTAINERS = n_containers or globals()['N_CONTAINERS']
    rng = np.random.default_rng(20260928)
    pool = []
    for _ in range(2000):
        n = int(rng.integers(2000, 9000))
        levels = np.repeat(rng.normal(450, 80, n // 8 + 1), 8)[:n]
        pool.append(np.clip(np.rint(levels + rng.normal(0, 8, n)), 0, 2047).astype(np.int16))
    with ThreadPoolExecutor(16) as workers:
        deflated = list(workers.map(lambda s: zlib.compress(s.tobytes(), 1), pool))
    fixture = list(zip(gold['multi_read_ids'], gold['multi_signals']))
    paths, every_id = [], []
    jobs = []
    for c in range(N_CONTAINERS):
        reads = []
        for k in range(READS_PER_CONTAINER):
            j = int(rng.integers(0, len(pool)))
            reads.append((str(uuid.UUID(bytes=rng.bytes(16), version=4)), pool[j], None,
                          deflated[j]))
        for rid, signal in fixture[c::N_CONTAINERS]:
            reads.append((rid, signal))
        every_id += [r[0] for r in reads]
        path = os.path.join(directory, 'stream_%02d.fast5' % c)
        paths.append(path)
        jobs.append((path, reads))

    def write(job):
        with open(job[0], 'wb') as f:
            f.write(hdf5_write.multi_read_fast5_bytes(job[1]))

    with ThreadPoolExecutor(8) as workers:
        list(workers.map(write, jobs))
    return paths, every_id


def test_a_model_on_fewer_cus_and_its_clone_give_the_same_calls(hip, hip_models, all_signals):
    """dbh_model_reserve_cus leaves CUs to other kernels (the streaming path's inflate queues),
    HipModel.clone() is one more queue on the same GPU: neither changes a result."""
    start, end = hip_models[START], hip_models[END]
    signals = list(all_signals)
    want = [m.classify_signals(signals, side, 6144, 0.5)
            for m, side in ((start, 'start'), (end, 'end'))]
    try:
        for reserve in (32, 200, 255, 100000):
            start.reserve_cus(reserve)
            end.reserve_cus(reserve)
            got = [m.classify_signals(signals, side, 6144, 0.5)
                   for m, side in ((start, 'start'), (end, 'end'))]
            for (wp, wc), (gp, gc) in zip(want, got):
                assert np.array_equal(wp, gp) and np.array_equal(wc, gc)
    finally:
        start.reserve_cus(0)
        end.reserve_cus(0)
    twin = start.clone()
    try:
        assert twin.device == start.device and twin.handle.value != start.handle.value
        gp, gc = twin.classify_signals(signals, 'start', 6144, 0.5)
        assert np.array_equal(want[0][0], gp) and np.array_equal(want[0][1], gc)
    finally:
        twin.close()
    with pytest.raises(hip.HipBackendError):
        start.reserve_cus(-1)


@pytest.fixture(scope='module')
def containers(tmp_path_factory, gold):
    directory = str(tmp_path_factory.mktemp('stream'))
    paths, ids = build_containers(directory, gold)
    yield directory, paths, ids
    shutil.rmtree(directory, ignore_errors=True)


def run_realtime(in_dir, out_dir, devices, monkeypatch, capsys, ordinals=None):
    from deepbinner_amd import deepbinner as cli
    import deepbinner_amd.realtime as realtime
    monkeypatch.setattr(realtime, 'POLL_SECONDS', 0)
    monkeypatch.setattr(shutil, 'which', lambda tool: None)       # no multi_to_single_fast5
    monkeypatch.setenv('DEEPBINNER_REALTIME_TABLE_ONLY', '1')
    if ordinals:
        monkeypatch.setenv('DEEPBINNER_DEVICE_ORDINALS', ordinals)
    else:
        monkeypatch.delenv('DEEPBINNER_DEVICE_ORDINALS', raising=False)
    argv = ['realtime', '--in_dir', in_dir, '--out_dir', out_dir, '--stop',
            '-s', os.path.join(MODEL_DIR, START + '.dbw'),
            '-e', os.path.join(MODEL_DIR, END + '.dbw')]
    if devices > 1:
        argv += ['--devices', str(devices)]
    capsys.readouterr()
    cli.main(argv)
    text = capsys.readouterr().out
    with open(os.path.join(out_dir, 'multi_read_classifications.tsv')) as f:
        return [line.split('\t') for line in f.read().splitlines()], text


def test_realtime_streams_100k_reads_of_multi_read_containers(hip, gold, containers, tmp_path,
                                                              monkeypatch, capsys):
    """configs[4] at one GPU's share of its stated scale (128,030 reads >= 1 M / 8; the whole
    1 M with DEEPBINNER_STREAM_READS=1000000): `deepbinner realtime` over 32 containers of 4,000
    reads - loader team -> pinned batches -> dbh_classify_pair_i16 -> table.  Every read is
    tabulated exactly once; the fixture reads get the reference's calls; two device queues (both
    on GPU 0) or one, and the per-read path of `classify_signals`, give the same table bit for bit."""
    import time
    directory, paths, every_id = containers
    t0 = time.perf_counter()
    table, text = run_realtime(directory, str(tmp_path / 'one'), 1, monkeypatch, capsys)
    seconds = time.perf_counter() - t0
    assert len(table) == len(every_id) == N_CONTAINERS * READS_PER_CONTAINER + 30
    assert sorted(r[0] for r in table) == sorted(every_id) and len(set(every_id)) == len(every_id)
    assert {r[2] for r in table} == set(paths)
    calls = {r[0]: r[1] for r in table}
    want = reference_final_calls(gold)
    for rid in gold['multi_read_ids']:
        assert calls[rid] == want[rid]
    assert sum(1 for r in table if r[1] != 'none') >= 10
    # the summary tables of the passes add up to the reads (5 containers per pass: realtime.py:86-94)
    assert text.count('Barcode     Count') == -(-N_CONTAINERS // 5)
    print('realtime: %d reads in %.1f s = %.0f reads/s (table only, incl. model loading)'
          % (len(table), seconds, len(table) / seconds))
    # two device queues on the one GPU: same rows in the same order
    table2, _ = run_realtime(directory, str(tmp_path / 'two'), 2, monkeypatch, capsys, '0,0')
    assert table2 == table
    # and the per-read route on a sample of the containers: call_batch per model through
    # dbh_classify_i16 + combine_calls on the host
    from deepbinner_amd import classify, fast5_native
    from deepbinner_amd.model_format import ModelWeights
    start = hip.HipModel(ModelWeights.load(os.path.join(MODEL_DIR, START + '.dbw'))[0])
    end = hip.HipModel(ModelWeights.load(os.path.join(MODEL_DIR, END + '.dbw'))[0])
    args = argparse.Namespace(scan_size=6144, score_diff=0.5, batch_size=256, verbose=True,
                              require_either=True, require_start=False, require_both=False)
    for path in paths[::12]:
        ids, samples, offsets, status = fast5_native.load_reads(path, threads=8)
        assert (status == 0).all()
        signals = [samples[offsets[i]:offsets[i + 1]] for i in range(len(ids))]
        found = {}
        for lo in range(0, len(ids), 1000):
            classify.classify_read_batch(ids[lo:lo + 1000], signals[lo:lo + 1000], start, 1024,
                                         end, 1024, 13, args, found)
        assert all(calls[rid] == found[rid] for rid in ids)
    # ... and against the ORACLE, not only against the GPU's other routes: one
    # whole container read by the pure-Python HDF5 reader (zlib's inflate), both sides through the
    # oracle's C port, combine_calls by the oracle's restatement - the table's rows for these
    # 4,000 reads.  A loader or inflate bug that garbled a read the same way on every GPU route
    # would show here.
    from deepbinner_amd import load_fast5s
    from oracle import dbref
    from deepbinner_amd.model_format import ModelWeights as MW
    path = paths[len(paths) // 2]
    reads = list(load_fast5s._python_iter_reads(path))
    assert READS_PER_CONTAINER <= len(reads) <= READS_PER_CONTAINER + 30      # (+ fixture reads)
    o_samples, o_offsets = pack([np.asarray(sig, dtype=np.int16) for _, sig in reads])
    side_calls = {}
    for side, name in (('start', START), ('end', END)):
        weights = MW.load(os.path.join(MODEL_DIR, name + '.dbw'))[0]
        side_calls[side] = dbref.CModel(weights).classify(o_samples, o_offsets, side, 6144, 0.5)[1]
    as_name = lambda c: 'none' if c == 0 else str(int(c))                # noqa: E731
    differ = [rid for (rid, _), a, b in zip(reads, side_calls['start'], side_calls['end'])
              if calls[rid] != classify_ref.combine_calls(as_name(a), as_name(b), 'require_either')]
    assert not differ, '%d of %d rows differ from the oracle' % (len(differ), len(reads))


@pytest.mark.skipif(os.environ.get('DEEPBINNER_STREAM_FULL', '1') == '0', reason='DEEPBINNER_STREAM_FULL=0')
def test_config4_streams_its_full_million_reads(hip, gold, tmp_path_factory, tmp_path, monkeypatch, capsys):
    """BASELINE.json configs[4] at its STATED size on the one GPU the test has: 250 containers of
    4,000 reads (+ the 30 fixture reads) = 1,000,030 reads through `deepbinner realtime` - loader
    team -> pinned batches -> both models + combine_calls per container -> table.  Every read is
    tabulated exactly once, the fixture reads get the reference's calls, and one whole container's
    4,000 rows are what the ORACLE's C port makes of the samples the pure-Python HDF5 reader (zlib's
    inflate) finds in the file.  (~7 GB of scratch files; DEEPBINNER_STREAM_FULL=0 skips it.)"""
    import time
    from deepbinner_amd import load_fast5s
    from deepbinner_amd.model_format import ModelWeights as MW
    from oracle import dbref
    n_containers = 250
    directory = str(tmp_path_factory.mktemp('stream_full'))
    try:
        t0 = time.perf_counter()
        paths, every_id = build_containers(directory, gold, n_containers)
        built = time.perf_counter() - t0
        t0 = time.perf_counter()
        table, text = run_realtime(directory, str(tmp_path / 'full'), 1, monkeypatch, capsys)
        seconds = time.perf_counter() - t0
        assert len(table) == len(every_id) == n_containers * READS_PER_CONTAINER + 30
        assert len(set(every_id)) == len(every_id) and sorted(r[0] for r in table) == sorted(every_id)
        assert {r[2] for r in table} == set(paths)
        calls = {r[0]: r[1] for r in table}
        want = reference_final_calls(gold)
        for rid in gold['multi_read_ids']:
            assert calls[rid] == want[rid]
        assert text.count('Barcode     Count') == -(-n_containers // 5)
        print('configs[4] at full size: %d reads in %.1f s = %.0f reads/s (table only, incl. model loading; '
              'containers written in %.1f s)' % (len(table), seconds, len(table) / seconds, built))
        path = paths[len(paths) // 3]
        reads = list(load_fast5s._python_iter_reads(path))
        o_samples, o_offsets = pack([np.asarray(sig, dtype=np.int16) for _, sig in reads])
        side_calls = {}
        for side, name in (('start', START), ('end', END)):
            weights = MW.load(os.path.join(MODEL_DIR, name + '.dbw'))[0]
            side_calls[side] = dbref.CModel(weights).classify(o_samples, o_offsets, side, 6144, 0.5)[1]
        as_name = lambda c: 'none' if c == 0 else str(int(c))                # noqa: E731
        differ = [rid for (rid, _), a, b in zip(reads, side_calls['start'], side_calls['end'])
                  if calls[rid] != classify_ref.combine_calls(as_name(a), as_name(b), 'require_either')]
        assert not differ, '%d of %d rows differ from the oracle' % (len(differ), len(reads))
    finally:
        shutil.rmtree(directory, ignore_errors=True)


def test_realtime_bins_multi_read_reads_on_the_gpu(hip, gold, tmp_path, monkeypatch, capsys):
    """The three fixture containers through `realtime` with the HIP backend, binning on: the table
    is the reference's calls (calls.json: its call_batch + combine_calls), and every read sits in
    the bin of its call as a one-read fast5 with its whole signal and its container's metadata."""
    from deepbinner_amd import deepbinner as cli, hdf5_lite, load_fast5s
    import deepbinner_amd.realtime as realtime
    monkeypatch.setattr(realtime, 'POLL_SECONDS', 0)
    monkeypatch.setattr(shutil, 'which', lambda tool: None)
    monkeypatch.delenv('DEEPBINNER_REALTIME_TABLE_ONLY', raising=False)
    in_dir, out_dir = tmp_path / 'in', tmp_path / 'out'
    shutil.copytree(os.path.join(GOLD, 'fast5', 'multi'), in_dir)
    cli.main(['realtime', '--in_dir', str(in_dir), '--out_dir', str(out_dir), '--stop',
              '-s', os.path.join(MODEL_DIR, START + '.dbw'),
              '-e', os.path.join(MODEL_DIR, END + '.dbw')])
    assert 'Wrote 30 one-read fast5 files' in capsys.readouterr().out
    rows = [r.split('\t') for r in
            (out_dir / 'multi_read_classifications.tsv').read_text().splitlines()]
    want = reference_final_calls(gold)
    assert {r[0]: r[1] for r in rows} == {rid: want[rid] for rid in gold['multi_read_ids']}
    originals = dict(zip(gold['multi_read_ids'], gold['multi_signals']))
    for read_id, call, source in rows:
        path = out_dir / realtime.bin_name(call) / (read_id + '.fast5')
        got_id, got = load_fast5s.get_read_id_and_signal(str(path))
        assert got_id == read_id and np.array_equal(got, originals[read_id])
        with hdf5_lite.File(str(path), 'r') as f, hdf5_lite.File(source, 'r') as container:
            mine, theirs = f['read_' + read_id], container['read_' + read_id]
            assert sorted(mine.keys()) == sorted(theirs.keys())
            for sub in ('channel_id', 'tracking_id', 'context_tags'):
                assert dict(mine[sub].attrs.items()) == dict(theirs[sub].attrs.items())
    assert sum(len(files) for _, _, files in os.walk(str(out_dir))) == 31


# ---- inflating on the GPU -----------------------------------------------------------------------
def every_fixture_file():
    import glob
    fast5 = os.path.join(GOLD, 'fast5')
    return (sorted(glob.glob(os.path.join(fast5, 'multi', '*.fast5'))) +
            sorted(glob.glob(os.path.join(fast5, 'single', '*.fast5'))) +
            sorted(glob.glob(os.path.join(fast5, 'h5py_variants', '*.fast5'))))


@pytest.mark.parametrize('host_inflate_above', [0, 20000, 1])
def test_raw_batches_decoded_on_the_gpu_are_the_loaders_samples(hip, hip_models, host_inflate_above):
    """Every committed fast5 file - the real MinKNOW containers, the one-read files, the 32 h5py
    variants (both on-disk generations, every layout, storage kind and filter, chunk indexes of
    every kind, sparse datasets) - as a raw batch through dbh_classify_pair_deflated: the decoded
    signals are bit for bit what the CPU loader (pinned to h5py and to the reference's own
    loader) returns, the calls what the CPU-loaded samples give; whichever side inflates (all
    streams on the GPU; long ones on the host; every stream on the host)."""
    from deepbinner_amd import fast5_native
    start, end = hip_models[START], hip_models[END]
    files = every_fixture_file()
    assert len(files) >= 40
    seen_zlib = seen_stored = 0
    for index, ids, offsets, status, comp, records in fast5_native.stream_raw(
            files, threads=4, depth=3, host_inflate_above=host_inflate_above):
        path = files[index]
        try:
            want = fast5_native.load_reads(path, threads=2)
        except OSError:
            assert ids is None, path
            continue
        assert ids == want[0] and np.array_equal(offsets, want[2]) and \
            np.array_equal(status, want[3]), path
        calls, stream_status, samples = hip.classify_pair_deflated(
            start, end, comp, records, offsets, 6144, 0.5, want_samples=True)
        assert (stream_status == 0).all(), (path, stream_status)
        assert np.array_equal(samples, want[1]), path
        want_calls = hip.classify_pair(start, end, want[1], want[2], 6144, 0.5)
        assert np.array_equal(calls, want_calls), path
        seen_zlib += int((records['mode'] == 0).sum())
        seen_stored += int((records['mode'] == 1).sum())
    assert seen_stored > 0 and (seen_zlib > 40 or host_inflate_above == 1)


@pytest.mark.parametrize('forward_stream', ['shared', 'own'])
def test_queues_in_parallel_give_what_one_queue_gives(hip, hip_models, forward_stream, monkeypatch):
    """Three queues (model replicas) on one GPU, a thread each, the same raw batches through
    dbh_classify_pair_deflated at the same time - with every queue's forward launches going
    through the device's ONE forward stream (what ships: dbh_api.hip) and with every queue on a
    stream of its own: the calls of every batch are what one queue alone gives."""
    import threading
    from deepbinner_amd import fast5_native, realtime
    monkeypatch.setenv('DEEPBINNER_FORWARD_STREAM', forward_stream)
    start, end = hip_models[START], hip_models[END]
    files = every_fixture_file()
    batches = [b for b in fast5_native.stream_raw(files, threads=4, depth=3, host_inflate_above=0)
               if b[1] is not None]
    assert batches
    alone = [hip.classify_pair_deflated(start, end, b[4], b[5], b[2], 6144, 0.5)[0] for b in batches]
    queues = [(start, end)] + realtime.queue_clones((start, end), 2)
    got = [[None] * len(batches) for _ in queues]
    failed = []

    def run(k):
        try:
            for _ in range(3):
                for j, b in enumerate(batches):
                    got[k][j] = hip.classify_pair_deflated(queues[k][0], queues[k][1], b[4], b[5], b[2],
                                                           6144, 0.5)[0]
        except Exception as e:          # noqa: BLE001 - reported below
            failed.append(repr(e))

    threads = [threading.Thread(target=run, args=(k,)) for k in range(len(queues))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failed, failed
    for k in range(len(queues)):
        for j in range(len(batches)):
            assert np.array_equal(got[k][j], alone[j]), (forward_stream, k, j)


def test_realtime_with_and_without_gpu_inflate(hip, gold, containers, tmp_path, monkeypatch, capsys):
    """The 100,000-read stream again, inflated by the GPU (the default) and by the host's threads
    (DEEPBINNER_GPU_INFLATE=0): the same table, row for row; and a container with a damaged chunk
    and an unreadable read gives what the CPU path gives (zlib on the host has the last word)."""
    import time
    directory, paths, every_id = containers
    monkeypatch.setenv('DEEPBINNER_GPU_INFLATE', '1')
    t0 = time.perf_counter()
    on, _ = run_realtime(directory, str(tmp_path / 'gpu'), 1, monkeypatch, capsys)
    seconds = time.perf_counter() - t0
    print('realtime, GPU inflate: %d reads in %.1f s = %.0f reads/s' % (len(on), seconds,
                                                                         len(on) / seconds))
    monkeypatch.setenv('DEEPBINNER_GPU_INFLATE', '0')
    off, _ = run_realtime(directory, str(tmp_path / 'cpu'), 1, monkeypatch, capsys)
    assert on == off and len(on) == len(every_id)
    # the split this box gets when nothing is forced (host_inflate_share: the host's threads keep
    # the longest streams, three queues on the GPU, 32 CUs left to the inflate kernels)
    monkeypatch.delenv('DEEPBINNER_GPU_INFLATE')
    import deepbinner_amd.realtime as realtime
    t0 = time.perf_counter()
    shared, _ = run_realtime(directory, str(tmp_path / 'shared'), 1, monkeypatch, capsys)
    seconds = time.perf_counter() - t0
    print('realtime, host share %d %%: %d reads in %.1f s = %.0f reads/s'
          % (realtime.host_inflate_share(1), len(shared), seconds, len(shared) / seconds))
    assert shared == off
    # damage: one byte in the middle of a deflate stream of one container
    from deepbinner_amd import fast5_native
    broken_dir = tmp_path / 'broken'
    broken_dir.mkdir()
    victim = str(broken_dir / 'stream_00.fast5')
    data = bytearray(open(paths[0], 'rb').read())
    gen = fast5_native.stream_raw([paths[0]], threads=2)
    _, ids, offsets, status, comp, records = next(gen)
    gen.close()
    rec = records[len(records) // 2]
    chunk = bytes(comp[rec['comp_offset']:rec['comp_offset'] + rec['comp_bytes']])
    at = bytes(data).find(chunk)
    assert at > 0
    data[at + len(chunk) // 2] ^= 0x5A
    with open(victim, 'wb') as f:
        f.write(bytes(data))
    tables = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('DEEPBINNER_GPU_INFLATE', flag)
        tables[flag], _ = run_realtime(str(broken_dir), str(tmp_path / ('broken_out_' + flag)), 1,
                                       monkeypatch, capsys)
    assert tables['1'] == tables['0']
    assert len(tables['1']) in (READS_PER_CONTAINER + 1, READS_PER_CONTAINER + 2) or \
        len(tables['1']) >= READS_PER_CONTAINER


def test_classify_one_read_files_with_the_gpu_inflating(hip, gold, tmp_path, monkeypatch, capsys):
    """`deepbinner classify` over a directory of one-read files with their Signals handed to the
    GPU as stored (classify.raw_inflate_share: on by itself for big directories on hosts with few
    cores per GPU; forced here) prints the table the CPU loader's path prints - every fixture
    file (old and new layouts, the 32 h5py variants: chunked, shuffled, checksummed, contiguous
    ...), copies of them by the hundred, a file that is not HDF5 and one with a damaged chunk
    (which the host's zlib gets the last word on) - whatever share of the inflating the host
    keeps."""
    sources = sorted(os.path.join(GOLD, 'fast5', sub, f)
                     for sub in ('single', 'h5py_variants')
                     for f in os.listdir(os.path.join(GOLD, 'fast5', sub)) if f.endswith('.fast5'))
    from deepbinner_amd import fast5_native
    sources = [f for f in sources if fast5_native.load_batch([f], None, 1)[3][0] == 0]
    assert len(sources) >= 25
    in_dir = tmp_path / 'reads'
    in_dir.mkdir()
    for k in range(1500):
        os.symlink(sources[k % len(sources)], str(in_dir / ('read_%05d.fast5' % k)))
    (in_dir / 'not_hdf5.fast5').write_bytes(b'definitely not HDF5' * 100)
    # a damaged deflate stream: flip a byte in the middle of the largest file's biggest chunk
    victim_source = max(sources, key=os.path.getsize)
    data = bytearray(open(victim_source, 'rb').read())
    _, _, _, comp, records = fast5_native.load_batch_raw([victim_source], 1, 0)
    rec = records[np.argmax(records['comp_bytes'])]
    chunk = bytes(comp[rec['comp_offset']:rec['comp_offset'] + rec['comp_bytes']])
    at = bytes(data).find(chunk)
    assert at > 0 and rec['mode'] == 0
    data[at + len(chunk) // 2] ^= 0x3C
    (in_dir / 'damaged.fast5').write_bytes(bytes(data))

    from deepbinner_amd import classify
    files = sorted(str(p) for p in in_dir.iterdir())
    models = classify.load_and_check_models(os.path.join(MODEL_DIR, START + '.dbw'),
                                            os.path.join(MODEL_DIR, END + '.dbw'), 6144)
    args = argparse.Namespace(verbose=False, batch_size=256, scan_size=6144, score_diff=0.5,
                              require_either=True, require_start=False, require_both=False,
                              loader_procs=0)

    def table(env):
        for name in ('DEEPBINNER_GPU_INFLATE', 'DEEPBINNER_HOST_INFLATE_SHARE',
                     'DEEPBINNER_RAW_CLASSIFY_MIN_FILES'):
            monkeypatch.delenv(name, raising=False)
        for name, value in env.items():
            monkeypatch.setenv(name, value)
        capsys.readouterr()
        calls, where = classify.classify_fast5_files(files, *models[:5], args,
                                                     verified_single_read=True)
        out = capsys.readouterr().out.splitlines()
        assert len(calls) == len(where)
        return out[0], sorted(out[1:])

    want = table({'DEEPBINNER_GPU_INFLATE': '0'})
    assert len(want[1]) >= 1500 and want[0].split('\t')[:2] == ['read_ID', 'barcode_call']
    for share in ('0', '40', '99'):
        got = table({'DEEPBINNER_RAW_CLASSIFY_MIN_FILES': '1',
                     'DEEPBINNER_HOST_INFLATE_SHARE': share})
        assert got == want, share
    # the verbose table (classify.py:157-171: probabilities and each side's own call beside the
    # final one) takes the same route - dbh_classify_pair_deflated_verbose hands them back - and
    # prints what the CPU loader's route prints, row for row, with two models and with one
    args.verbose = True
    want_verbose = table({'DEEPBINNER_GPU_INFLATE': '0'})
    assert len(want_verbose[0].split('\t')) == 2 + 2 * (13 + 1)
    for share in ('0', '60'):
        got = table({'DEEPBINNER_RAW_CLASSIFY_MIN_FILES': '1',
                     'DEEPBINNER_HOST_INFLATE_SHARE': share})
        assert got == want_verbose, share
    assert [row.split('\t')[:2] for row in want_verbose[1]] == [row.split('\t') for row in want[1]]
    one = classify.load_and_check_models(os.path.join(MODEL_DIR, START + '.dbw'), None, 6144)
    models = one
    want_one = table({'DEEPBINNER_GPU_INFLATE': '0'})
    assert len(want_one[0].split('\t')) == 2 + 13
    assert table({'DEEPBINNER_RAW_CLASSIFY_MIN_FILES': '1',
                  'DEEPBINNER_HOST_INFLATE_SHARE': '0'}) == want_one

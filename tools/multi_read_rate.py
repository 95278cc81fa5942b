#!/usr/bin/env python3
"""BASELINE.json configs[4] in the small: stream multi-read fast5 containers of MinKNOW's size
(4,000 reads per file) through the native loader and both models, the way ``deepbinner realtime``
does when it finds multi-read files (Session._tabulate_multi_read_files).

The containers are written on the spot by the image's one interpreter with h5py
(/opt/conda/bin/python3.9; gzip level 1, one chunk per read, the layout MinKNOW / ont_fast5_api
write); without it the tool says so and exits.  Reported:
  1. f5_load_reads alone (scanned ends only) at several thread counts, and the Python reader;
  2. load + classify with start and end models, scan_size 6144, batch 256, loading of container
     k + 1 overlapped with classification of container k on a background thread;
  3. per number of device queues of the single-process dispatcher (classify.dispatch_batches; on a
     one-GPU box the queues beyond the first share GPU 0): the GPU side alone (batches already in
     memory), the loader alone, both together - and which of the two bounds the whole.
Usage: python tools/multi_read_rate.py [--files 3] [--reads 4000] [--mean-length 27000]"""
import argparse
import io
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
CONDA_PYTHON = '/opt/conda/bin/python3.9'

WRITER = r'''
import sys, uuid
import h5py, numpy as np
path, n_reads, mean_length, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(seed)
with h5py.File(path, 'w') as f:
    f.attrs['file_version'] = np.bytes_('2.0')
    for k in range(n_reads):
        n = int(np.clip(rng.lognormal(np.log(mean_length) - 0.32, 0.8), 2000, 400000))
        levels = rng.normal(450, 80, size=n // 8 + 1)
        signal = np.clip(np.rint(np.repeat(levels, 8)[:n] + rng.normal(0, 8, size=n)), 0, 2047)
        read_id = str(uuid.UUID(bytes=rng.bytes(16), version=4))
        raw = f.create_group('read_' + read_id + '/Raw')
        raw.attrs['read_id'] = np.bytes_(read_id)
        raw.attrs['read_number'] = np.int32(k)
        raw.attrs['start_time'] = np.uint64(k * 4000)
        raw.attrs['duration'] = np.uint32(n)
        raw.attrs['median_before'] = 220.0
        raw.create_dataset('Signal', data=signal.astype('<i2'), chunks=(n,), compression='gzip',
                           compression_opts=1)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--files', type=int, default=3)
    ap.add_argument('--reads', type=int, default=4000)
    ap.add_argument('--mean-length', type=int, default=27000)
    opts = ap.parse_args()
    if not os.path.exists(CONDA_PYTHON):
        sys.exit('no interpreter with h5py at {}: cannot write the containers'.format(CONDA_PYTHON))
    from deepbinner_amd import classify, fast5_native, load_fast5s
    out = {'files': opts.files, 'reads_per_file': opts.reads, 'host_threads': os.cpu_count()}
    with tempfile.TemporaryDirectory() as tmp:
        paths = [os.path.join(tmp, 'batch_%d.fast5' % k) for k in range(opts.files)]
        t0 = time.perf_counter()
        jobs = [subprocess.Popen([CONDA_PYTHON, '-c', WRITER, p, str(opts.reads),
                                  str(opts.mean_length), str(100 + k)])
                for k, p in enumerate(paths)]
        if any(j.wait() != 0 for j in jobs):
            sys.exit('writing the containers with h5py failed')
        out['h5py_write_seconds'] = round(time.perf_counter() - t0, 1)
        out['container_MB'] = round(sum(os.path.getsize(p) for p in paths) / 1e6 / opts.files, 1)

        ids, samples, offsets, status = fast5_native.load_reads(paths[0], threads=8)
        assert (status == 0).all() and len(ids) == opts.reads
        out['mean_samples_per_read'] = int(offsets[-1] // opts.reads)
        rates = {}
        for threads in (1, 8, 16, 32, 64, 128):
            if threads > (os.cpu_count() or 1):
                continue
            t0 = time.perf_counter()
            for p in paths:
                fast5_native.load_reads(p, keep=6656, threads=threads)
            rates['%d threads' % threads] = round(opts.files * opts.reads /
                                                  (time.perf_counter() - t0))
        os.environ['DEEPBINNER_FAST5_READER'] = 'python'
        t0 = time.perf_counter()
        n = 0
        for _, _ in load_fast5s.iter_reads(paths[0]):
            n += 1
            if n == 500:
                break
        rates['python reader (500 reads)'] = round(n / (time.perf_counter() - t0))
        os.environ['DEEPBINNER_FAST5_READER'] = 'native'
        out['f5_load_reads, scanned ends'] = {'reads_per_s': rates}

        models = os.path.join(REPO, 'deepbinner_amd', 'models')
        sm, si, em, ei, osz, _ = classify.load_and_check_models(
            os.path.join(models, 'EXP-NBD103_read_starts.dbw'),
            os.path.join(models, 'EXP-NBD103_read_ends.dbw'), 6144, out_dest=io.StringIO())
        args = argparse.Namespace(verbose=False, batch_size=256, scan_size=6144, score_diff=0.5,
                                  require_either=True, require_start=False, require_both=False)
        threads = min(32, max(1, (os.cpu_count() or 4) // 4))

        def load(path, box):
            box.append(fast5_native.load_reads(path, keep=6144 + 512, threads=threads))

        calls = {}
        t0 = time.perf_counter()
        box = []
        worker = threading.Thread(target=load, args=(paths[0], box))
        worker.start()
        for k in range(opts.files):
            worker.join()
            ids, samples, offsets, _ = box.pop()
            if k + 1 < opts.files:
                worker = threading.Thread(target=load, args=(paths[k + 1], box))
                worker.start()
            signals = [samples[offsets[i]:offsets[i + 1]] for i in range(len(ids))]
            for lo in range(0, len(ids), args.batch_size):
                hi = min(lo + args.batch_size, len(ids))
                chunk = classify.PackedSignals(signals[lo:hi], samples[offsets[lo]:offsets[hi]],
                                               offsets[lo:hi + 1] - offsets[lo])
                classify.classify_read_batch(ids[lo:hi], chunk, sm, si, em, ei, osz, args, calls)
        dt = time.perf_counter() - t0
        out['load + classify (start and end models, scan 6144, batch 256)'] = {
            'loader_threads': threads, 'seconds': round(dt, 3),
            'reads_per_s': round(opts.files * opts.reads / dt), 'distinct_reads': len(calls)}
        # ---- 3. the dispatcher, per number of device queues -------------------------------
        from deepbinner_amd import hip_backend
        visible = hip_backend.device_count()
        loaded = [fast5_native.load_reads(p, keep=6144 + 512, threads=threads) for p in paths]

        def batches_of(container):
            ids, samples, offsets, _ = container
            signals = [samples[offsets[i]:offsets[i + 1]] for i in range(len(ids))]
            for lo in range(0, len(ids), args.batch_size):
                hi = min(lo + args.batch_size, len(ids))
                yield ids[lo:hi], classify.PackedSignals(
                    signals[lo:hi], samples[offsets[lo]:offsets[hi]],
                    offsets[lo:hi + 1] - offsets[lo])

        def work(batch, start_replica, end_replica):
            found = {}
            classify.classify_read_batch(batch[0], batch[1], start_replica, si, end_replica, ei,
                                         osz, args, found)
            return len(found)

        t0 = time.perf_counter()
        for p in paths:
            fast5_native.load_reads(p, keep=6144 + 512, threads=threads)
        loader_rate = opts.files * opts.reads / (time.perf_counter() - t0)
        per_devices = {}
        for n_queues in (1, 2, 4, 8):
            ordinals = [d % max(visible, 1) for d in range(n_queues)]
            if n_queues > 1 and visible == 1 and n_queues > 2:
                continue                    # more than two queues on one GPU says nothing new
            os.environ['DEEPBINNER_DEVICE_ORDINALS'] = ','.join(map(str, ordinals))
            classify.set_tensorflow_threads(argparse.Namespace(devices=n_queues))
            sm_n, _, em_n, _, _, _ = classify.load_and_check_models(
                os.path.join(models, 'EXP-NBD103_read_starts.dbw'),
                os.path.join(models, 'EXP-NBD103_read_ends.dbw'), 6144, out_dest=io.StringIO())
            replicas = classify.device_replicas(sm_n, em_n)
            every = [b for c in loaded for b in batches_of(c)]
            list(classify.dispatch_batches(iter(every[:8]), replicas, work))          # warm-up
            t0 = time.perf_counter()
            done = sum(classify.dispatch_batches(iter(every), replicas, work))
            gpu_rate = done / (time.perf_counter() - t0)

            def stream():                   # containers loaded one ahead, as realtime does
                box = []
                worker = threading.Thread(target=load, args=(paths[0], box))
                worker.start()
                for k in range(opts.files):
                    worker.join()
                    container = box.pop()
                    if k + 1 < opts.files:
                        worker = threading.Thread(target=load, args=(paths[k + 1], box))
                        worker.start()
                    yield from batches_of(container)

            t0 = time.perf_counter()
            done = sum(classify.dispatch_batches(stream(), replicas, work))
            both = done / (time.perf_counter() - t0)
            per_devices['%d queue(s) on GPU(s) %s' % (n_queues, sorted(set(ordinals)))] = {
                'gpu_side_alone_reads_per_s': round(gpu_rate),
                'loader_alone_reads_per_s': round(loader_rate),
                'load_and_classify_reads_per_s': round(both),
                'bound_by': 'loader' if loader_rate < gpu_rate else 'gpu side (incl. its host work)'}
        os.environ.pop('DEEPBINNER_DEVICE_ORDINALS', None)
        out['dispatcher (start + end models, scan 6144, batch 256, %d loader threads)' % threads] = \
            per_devices
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU box: stream multi-read fast5 containers of MinKNOW's size
(4,000 reads per file, gzip level 1, one chunk per read) through the native loader and both
models, the way ``deepbinner realtime`` does when it finds multi-read files
(realtime.Session._tabulate_multi_read_files).

The containers are written on the spot by the image's interpreter with h5py
(/opt/conda/bin/python3.9: the real HDF5 library, the layout MinKNOW / ont_fast5_api write), or,
without it, by this package's own container writer (hdf5_write.multi_read_fast5_bytes).  Reported:
  1. the loader alone (scanned ends only): one f5_load_reads call per container, as round 2
     measured it, and the stream (f5_stream_*: a thread team working several containers ahead),
     per team size, into pageable and into pinned buffers;
  2. the GPU side alone: the containers' packed batches already in (pinned) host memory, both
     models + combine_calls per container in one dbh_classify_pair_i16 call;
  3. both together through the dispatcher, per number of device queues (on a one-GPU box the
     queues beyond the first share GPU 0) - and which of the two bounds the whole;
  4. the same with the inflating moved to the GPU (f5_stream_open_raw -> dbh_classify_pair_deflated):
     the host's share per read (CPU seconds of the whole process per read), the raw loader alone,
     and the device's time per stage (upload / inflate / classify).
Usage: python tools/multi_read_rate.py [--files 16] [--reads 4000] [--mean-length 27000]"""
import argparse
import io
import json
import os
import subprocess
import sys
import tempfile
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


def write_with_own_writer(path, n_reads, mean_length, seed):
    import uuid
    import numpy as np
    from deepbinner_amd import hdf5_write
    rng = np.random.default_rng(seed)
    reads = []
    for _ in range(n_reads):
        n = int(np.clip(rng.lognormal(np.log(mean_length) - 0.32, 0.8), 2000, 400000))
        levels = rng.normal(450, 80, size=n // 8 + 1)
        signal = np.clip(np.rint(np.repeat(levels, 8)[:n] + rng.normal(0, 8, size=n)), 0, 2047)
        reads.append((str(uuid.UUID(bytes=rng.bytes(16), version=4)), signal.astype(np.int16)))
    with open(path, 'wb') as f:
        f.write(hdf5_write.multi_read_fast5_bytes(reads))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--files', type=int, default=16)
    ap.add_argument('--reads', type=int, default=4000)
    ap.add_argument('--mean-length', type=int, default=27000)
    ap.add_argument('--no-gpu', action='store_true', help='loader measurements only')
    opts = ap.parse_args()
    from deepbinner_amd import classify, fast5_native
    cpus = os.cpu_count() or 1
    total = opts.files * opts.reads
    out = {'files': opts.files, 'reads_per_file': opts.reads, 'host_threads': cpus}
    with tempfile.TemporaryDirectory() as tmp:
        paths = [os.path.join(tmp, 'batch_%02d.fast5' % k) for k in range(opts.files)]
        t0 = time.perf_counter()
        if os.path.exists(CONDA_PYTHON):
            out['containers_written_by'] = 'h5py (libhdf5)'
            jobs = [subprocess.Popen([CONDA_PYTHON, '-c', WRITER, p, str(opts.reads),
                                      str(opts.mean_length), str(100 + k)])
                    for k, p in enumerate(paths)]
            if any(j.wait() != 0 for j in jobs):
                sys.exit('writing the containers with h5py failed')
        else:
            out['containers_written_by'] = 'hdf5_write.multi_read_fast5_bytes'
            for k, p in enumerate(paths):
                write_with_own_writer(p, opts.reads, opts.mean_length, 100 + k)
        out['write_seconds'] = round(time.perf_counter() - t0, 1)
        out['container_MB'] = round(sum(os.path.getsize(p) for p in paths) / 1e6 / opts.files, 1)
        keep = classify.scanned_end_samples(6144)

        ids, samples, offsets, status = fast5_native.load_reads(paths[0], threads=8)
        assert (status == 0).all() and len(ids) == opts.reads
        out['mean_samples_per_read'] = int(offsets[-1] // opts.reads)
        del samples

        # ---- 1. the loader alone ----------------------------------------------------------
        teams = [t for t in (1, 8, 16, 32, 64, 128, 192, 256) if t <= cpus]
        per_call = {}
        for threads in teams:
            if threads > 64 and threads not in (128,):
                continue
            subset = paths[:max(2, min(len(paths), threads))]
            t0 = time.perf_counter()
            for p in subset:
                fast5_native.load_reads(p, keep=keep, threads=threads)
            per_call['%d threads' % threads] = round(len(subset) * opts.reads /
                                                     (time.perf_counter() - t0))
        out['loader alone, one f5_load_reads call per container'] = {'reads_per_s': per_call}

        def stream_rate(threads, depth, subset):
            t0 = time.perf_counter()
            n = 0
            for _, ids, samples, _, _ in fast5_native.stream_reads(subset, keep=keep,
                                                                   threads=threads, depth=depth):
                n += len(ids)
            return round(n / (time.perf_counter() - t0))

        streamed = {}
        for threads in teams:
            subset = paths[:max(2, min(len(paths), threads // 2))]
            stream_rate(threads, 4, subset[:2])                     # warm the buffer pool
            streamed['%d threads' % threads] = stream_rate(threads, 4, subset)
        out['loader alone, f5_stream (4 containers in flight), pageable buffers'] = {
            'reads_per_s': streamed}
        best_team = max(teams, key=lambda t: streamed['%d threads' % t])
        out['depth sweep at %d threads' % best_team] = {
            'depth %d' % d: stream_rate(best_team, d, paths) for d in (1, 2, 3, 4, 6, 8)}
        if opts.no_gpu:
            print(json.dumps(out, indent=1))
            return

        # ---- 2./3. with the GPU ---------------------------------------------------------------
        from deepbinner_amd import hip_backend
        models = os.path.join(REPO, 'deepbinner_amd', 'models')
        args = argparse.Namespace(verbose=False, batch_size=256, scan_size=6144, score_diff=0.5,
                                  require_either=True, require_start=False, require_both=False)
        visible = hip_backend.device_count()

        def load_models(n_queues):
            ordinals = [d % max(visible, 1) for d in range(n_queues)]
            os.environ['DEEPBINNER_DEVICE_ORDINALS'] = ','.join(map(str, ordinals))
            classify.set_tensorflow_threads(argparse.Namespace(devices=n_queues))
            sm, _, em, _, _, _ = classify.load_and_check_models(
                os.path.join(models, 'EXP-NBD103_read_starts.dbw'),
                os.path.join(models, 'EXP-NBD103_read_ends.dbw'), 6144, out_dest=io.StringIO())
            return classify.device_replicas(sm, em), ordinals

        def work(batch, start_replica, end_replica):
            _, ids, samples, offsets, _ = batch
            return len(classify.classify_packed_numbers(samples, offsets, start_replica,
                                                        end_replica, args))

        replicas, _ = load_models(1)          # (build_model switches the loader to pinned buffers)
        pinned = {}
        for threads in teams:
            if threads < 16:
                continue
            subset = paths[:max(2, min(len(paths), threads // 2))]
            stream_rate(threads, 4, subset[:4])
            pinned['%d threads' % threads] = stream_rate(threads, 4, subset)
        out['loader alone, f5_stream (4 containers in flight), pinned buffers'] = {
            'reads_per_s': pinned}
        loader_team = max(teams, key=lambda t: pinned.get('%d threads' % t, 0))
        loader_rate = pinned['%d threads' % loader_team]

        loaded = list(fast5_native.stream_reads(paths[:8], keep=keep, threads=loader_team, depth=4))
        out['batches_in_pinned_memory'] = bool(hip_backend.is_pinned(loaded[0][2]))
        per_queues = {}
        for n_queues in (1, 2, 4, 8):
            if n_queues > 2 and visible == 1:
                continue                    # more than two queues on one GPU says nothing new
            replicas, ordinals = load_models(n_queues)
            sum(classify.dispatch_batches(iter(loaded[:2]), replicas, work))          # warm-up
            t0 = time.perf_counter()
            done = sum(classify.dispatch_batches(iter(loaded * 3), replicas, work))
            gpu_rate = done / (time.perf_counter() - t0)
            best, best_cpu = 0.0, 0.0
            for _ in range(2):
                t0, c0 = time.perf_counter(), time.process_time()
                stream = fast5_native.stream_reads(paths, keep=keep, threads=loader_team, depth=4)
                done = sum(classify.dispatch_batches(stream, replicas, work))
                wall, cpu = time.perf_counter() - t0, time.process_time() - c0
                if done / wall > best:
                    best, best_cpu = done / wall, cpu / done
            assert done == total
            per_queues['%d queue(s) on GPU(s) %s' % (n_queues, sorted(set(ordinals)))] = {
                'gpu_side_alone_reads_per_s': round(gpu_rate),
                'loader_alone_reads_per_s': round(loader_rate),
                'load_and_classify_reads_per_s': round(best),
                'host_cpu_us_per_read': round(best_cpu * 1e6, 1),
                'bound_by': 'loader' if loader_rate < gpu_rate else 'gpu side'}
        os.environ.pop('DEEPBINNER_DEVICE_ORDINALS', None)
        out['stream -> dispatcher -> dbh_classify_pair_i16 (start + end models, scan 6144, '
            '%d loader threads)' % loader_team] = per_queues
        # ---- 4. inflate on the GPU -------------------------------------------------------------
        from deepbinner_amd import realtime
        share = realtime.host_inflate_share(1)
        above = -share
        out['host_inflate_share_per_cent (realtime.host_inflate_share: %d usable cpus, 1 GPU)'
            % classify.usable_cpus()] = share
        raw_rates = {}
        for threads in (1, 2, 4, 8, 16):
            if threads > cpus:
                continue
            t0, c0 = time.perf_counter(), time.process_time()
            n = 0
            for item in fast5_native.stream_raw(paths, threads=threads, depth=4,
                                                host_inflate_above=0):
                n += len(item[1])
            wall, cpu = time.perf_counter() - t0, time.process_time() - c0
            raw_rates['%d threads' % threads] = {'reads_per_s': round(n / wall),
                                                 'cpu_us_per_read': round(cpu / n * 1e6, 1)}
        out['raw loader alone (chunks as stored, nothing inflated)'] = raw_rates

        def raw_work(item, start_replica, end_replica):
            _, ids, offsets, _, comp, records = item
            calls, status = hip_backend.classify_pair_deflated(
                start_replica, end_replica, comp, records, offsets, 6144, 0.5)
            assert (status == 0).all()
            return len(calls)

        gpu_inflate = {}
        team = min(16, classify.usable_cpus())
        # the split: from everything on the GPU to everything on the host, with the queues and
        # the CUs realtime.py gives the inflate kernels at that split
        sweep = {}
        replicas, _ = load_models(1)
        clones = {}
        for host_share in (0, 20, 40, 50, 60, 70, 100):
            queues, models = realtime.inflate_queues(replicas, host_share)
            best, best_cpu = 0.0, 0.0
            for _ in range(2):
                t0, c0 = time.perf_counter(), time.process_time()
                stream = fast5_native.stream_raw(paths, threads=team, depth=len(queues) + 2,
                                                 host_inflate_above=-host_share)
                done = sum(classify.dispatch_batches(stream, queues, raw_work))
                wall, cpu = time.perf_counter() - t0, time.process_time() - c0
                if done / wall > best:
                    best, best_cpu = done / wall, cpu / done
            for model in models:
                model.reserve_cus(0)
            sweep['host inflates %d %% of the bytes (%d queues)' % (host_share, len(queues))] = {
                'load_and_classify_reads_per_s': round(best),
                'host_cpu_us_per_read': round(best_cpu * 1e6, 1)}
        gpu_inflate['split sweep, %d loader threads' % team] = sweep
        raw_loaded = list(fast5_native.stream_raw(paths[:4], threads=8, depth=4,
                                                  host_inflate_above=above))
        stages = [hip_backend.classify_pair_deflated(
            replicas[0][0], replicas[0][1], item[4], item[5], item[2], 6144, 0.5,
            want_stages=True)[2] for item in raw_loaded * 2][len(raw_loaded):]
        gpu_inflate['device_ms_per_container alone on the GPU (upload, inflate, classify)'] = [
            round(float(sum(s[k] for s in stages) / len(stages)), 2) for k in range(3)]
        rec = raw_loaded[0][5]
        gpu_inflate['streams_per_container'] = {
            'zlib (GPU)': int((rec['mode'] == 0).sum()),
            'stored / host-inflated': int((rec['mode'] == 1).sum())}
        # the pipeline as deepbinner_amd/realtime.py runs it: several containers in flight on the
        # one GPU (model replicas = queues), CUs left out of the forward launches for the inflate
        # kernels of the containers behind
        for n_queues, n_cus in ((1, 0), (2, 32), (3, 0), (3, 32), (4, 32)):
            os.environ['DEEPBINNER_INFLATE_QUEUES'] = str(n_queues)
            os.environ['DEEPBINNER_INFLATE_CUS'] = str(n_cus)
            queues, models = realtime.inflate_queues(replicas, share)
            sum(classify.dispatch_batches(iter(raw_loaded[:2]), queues, raw_work))       # warm-up
            t0 = time.perf_counter()
            done = sum(classify.dispatch_batches(iter(raw_loaded * 4), queues, raw_work))
            gpu_rate = done / (time.perf_counter() - t0)
            best, best_cpu = 0.0, 0.0
            for _ in range(2):
                t0, c0 = time.perf_counter(), time.process_time()
                stream = fast5_native.stream_raw(paths, threads=team, depth=n_queues + 2,
                                                 host_inflate_above=above)
                done = sum(classify.dispatch_batches(stream, queues, raw_work))
                wall, cpu = time.perf_counter() - t0, time.process_time() - c0
                if done / wall > best:
                    best, best_cpu = done / wall, cpu / done
            assert done == total
            for model in models:
                model.reserve_cus(0)
            gpu_inflate['%d queue(s), %d CUs left to the inflate kernels' % (n_queues, n_cus)] = {
                'gpu_side_alone_reads_per_s': round(gpu_rate),
                'load_and_classify_reads_per_s': round(best),
                'host_cpu_us_per_read': round(best_cpu * 1e6, 1)}
        os.environ.pop('DEEPBINNER_INFLATE_QUEUES', None)
        os.environ.pop('DEEPBINNER_INFLATE_CUS', None)
        os.environ.pop('DEEPBINNER_DEVICE_ORDINALS', None)
        out['inflate shared with the GPU: raw stream -> dispatcher -> dbh_classify_pair_deflated '
            '(host share %d %% unless stated)' % share] = gpu_inflate
        # the GPU side without the host path: 24 windows per read at the resident kernel rate
        out['note'] = ('GPU-resident ceiling for this workload: bench.py windows/s / 24 windows '
                       'per read (two models x 12 scan steps)')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

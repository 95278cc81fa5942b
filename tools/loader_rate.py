#!/usr/bin/env python3
"""The real-file path end to end: `deepbinner classify` over a directory of one-read fast5 files
(the seven reads the reference's tests ship, replicated as symlinks), start + end models, default
geometry (scan_size 6144, batch 256) - how fast the host loader feeds the GPU.
  1. loader alone: serial (the reference's loop) and LoaderPool at several process counts;
  2. classify_fast5_files on the HIP backend with the same loader settings.
Usage: python tools/loader_rate.py [n_files [batch_size]]"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepbinner_amd import classify, load_fast5s             # noqa: E402

FAST5_DIR = os.path.join(REPO, 'tests', 'golden', 'fast5', 'single')
MODELS = os.path.join(REPO, 'deepbinner_amd', 'models')


def main():
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    batch_size = int(sys.argv[2]) if len(sys.argv) > 2 else 256      # --batch_size of the classify runs
    sources = sorted(load_fast5s.find_all_fast5s(FAST5_DIR))
    out = {'files': n_files, 'host_threads': os.cpu_count()}
    with tempfile.TemporaryDirectory() as tmp:
        files = []
        for i in range(n_files):
            dst = os.path.join(tmp, 'read_%06d.fast5' % i)
            os.symlink(sources[i % len(sources)], dst)
            files.append(dst)

        os.environ['DEEPBINNER_FAST5_READER'] = 'python'
        t0 = time.perf_counter()
        samples = 0
        for f in files[:1024]:
            samples += len(load_fast5s.get_read_id_and_signal(f)[1])
        dt = time.perf_counter() - t0
        out['loader serial (1,024 files)'] = {'reads_per_s': round(1024 / dt),
                                              'mean_samples_per_read': samples // 1024}
        for procs in (4, 8, 16, 32):
            if procs > (os.cpu_count() or 1):
                continue
            with load_fast5s.LoaderPool(procs) as pool:
                list(pool.load(files[:256]))            # workers started and warm
                rates = {}
                for label, keep in (('whole reads', None), ('scanned ends only', 6656)):
                    t0 = time.perf_counter()
                    n = sum(1 for _ in pool.load(files, keep))
                    rates[label] = round(n / (time.perf_counter() - t0))
            out['loader pool, %d processes' % procs] = {'reads_per_s': rates}

        from deepbinner_amd import fast5_native
        if fast5_native.available():
            rates = {}
            for threads in (1, 4, 8, 16, 32, 64):
                if threads > (os.cpu_count() or 1):
                    continue
                fast5_native.load_batch(files[:256], 6656, threads)
                t0 = time.perf_counter()
                for i in range(0, len(files), 256):
                    fast5_native.load_batch(files[i:i + 256], 6656, threads)
                rates['%d threads' % threads] = round(len(files) / (time.perf_counter() - t0))
            out['native loader (libdeepbinner_fast5.so), batches of 256, scanned ends'] = {
                'reads_per_s': rates}

        if fast5_native.available():
            # multi-read containers: all reads of a file through f5_load_reads (the three sample
            # files hold 10 reads each, so this is the per-call floor rather than a streaming rate)
            multi = sorted(load_fast5s.find_all_fast5s(os.path.join(os.path.dirname(FAST5_DIR),
                                                                   'multi')))
            rates = {}
            for threads in (1, 4, 10):
                t0 = time.perf_counter()
                n = 0
                for _ in range(200):
                    for path in multi:
                        n += len(fast5_native.load_reads(path, keep=6656, threads=threads)[0])
                rates['%d threads' % threads] = round(n / (time.perf_counter() - t0))
            os.environ['DEEPBINNER_FAST5_READER'] = 'python'
            t0 = time.perf_counter()
            n = sum(1 for _ in range(20) for path in multi for _ in load_fast5s.iter_reads(path))
            rates['python reader'] = round(n / (time.perf_counter() - t0))
            out['multi-read files (10 reads each), scanned ends'] = {'reads_per_s': rates}

        sm, si, em, ei, osz, _ = classify.load_and_check_models(
            os.path.join(MODELS, 'EXP-NBD103_read_starts.dbw'),
            os.path.join(MODELS, 'EXP-NBD103_read_ends.dbw'), 6144, out_dest=io.StringIO())
        for reader, procs in (('python', 1), ('python', 8), ('native', 0)):
            if procs > (os.cpu_count() or 1) or (reader == 'native'
                                                 and not fast5_native.available()):
                continue
            os.environ['DEEPBINNER_FAST5_READER'] = reader
            args = argparse.Namespace(verbose=False, batch_size=batch_size, scan_size=6144,
                                      score_diff=0.5, require_either=True, require_start=False,
                                      require_both=False, loader_procs=procs)
            subset = files if (procs > 1 or reader == 'native') else files[:1024]
            sink = io.StringIO()
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(io.StringIO()):
                calls, _ = classify.classify_fast5_files(subset, sm, si, em, ei, osz, args,
                                                         verified_single_read=True)
            dt = time.perf_counter() - t0
            out['classify_fast5_files start+end models, %s reader, loader_procs %d'
                % (reader, procs)] = {
                'files': len(subset), 'seconds': round(dt, 3),
                'reads_per_s': round(len(subset) / dt), 'distinct_reads': len(calls)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""`deepbinner realtime` on multi-read containers with the binning ON (its default: every read
ends up as a one-read fast5 file in the directory of its barcode): reads/s from the first
container to the last file written, and where the host's time goes.

    python tools/realtime_bin_rate.py [--files 4] [--reads 4000]

The containers are written first (h5py where the box has it, the package's own writer
otherwise); model loading is timed apart.
"""
import argparse
import io
import json
import os
import shutil
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--files', type=int, default=4)
    ap.add_argument('--reads', type=int, default=4000)
    ap.add_argument('--table-only', action='store_true')
    ap.add_argument('--single-files', type=int, default=0,
                    help='instead of containers: this many one-read files (links to 4,000 '
                         'distinct ones), which realtime classifies and MOVES into their bins')
    opts = ap.parse_args()
    import subprocess
    import multi_read_rate
    from deepbinner_amd import deepbinner as cli
    import deepbinner_amd.realtime as realtime
    tmp = tempfile.mkdtemp(prefix='dbrt_')
    try:
        in_dir, out_dir = os.path.join(tmp, 'in'), os.path.join(tmp, 'out')
        os.makedirs(in_dir)
        paths = [os.path.join(in_dir, 'batch_%02d.fast5' % k) for k in range(opts.files)]
        if opts.single_files:
            # distinct reads: containers first, their reads written out as one-read files by the
            # native writer
            from deepbinner_amd import fast5_native
            n_containers = (opts.single_files + 3999) // 4000
            scratch = os.path.join(tmp, 'containers')
            os.makedirs(scratch)
            left = opts.single_files
            for k in range(n_containers):
                container = os.path.join(scratch, 'c%03d.fast5' % k)
                if os.path.exists(multi_read_rate.CONDA_PYTHON):
                    subprocess.run([multi_read_rate.CONDA_PYTHON, '-c', multi_read_rate.WRITER,
                                    container, '4000', '27000', str(100 + k)], check=True)
                else:
                    multi_read_rate.write_with_own_writer(container, 4000, 27000, 100 + k)
                take = min(4000, left)
                status, _ = fast5_native.write_single_reads(
                    container, list(range(take)),
                    [os.path.join(in_dir, 'read_%03d_%04d.fast5' % (k, i)) for i in range(take)])
                assert (status == 0).all()
                os.remove(container)
                left -= take
            opts.files, opts.reads = opts.single_files, 1
        elif os.path.exists(multi_read_rate.CONDA_PYTHON):
            jobs = [subprocess.Popen([multi_read_rate.CONDA_PYTHON, '-c', multi_read_rate.WRITER, p,
                                      str(opts.reads), '27000', str(100 + k)])
                    for k, p in enumerate(paths)]
            assert all(j.wait() == 0 for j in jobs)
        else:
            for k, p in enumerate(paths):
                multi_read_rate.write_with_own_writer(p, opts.reads, 27000, 100 + k)
        realtime.POLL_SECONDS = 0
        shutil.which = lambda tool: None                     # no multi_to_single_fast5
        if opts.table_only:
            os.environ['DEEPBINNER_REALTIME_TABLE_ONLY'] = '1'
        models = os.path.join(REPO, 'deepbinner_amd', 'models')
        argv = ['realtime', '--in_dir', in_dir, '--out_dir', out_dir, '--stop',
                '-s', os.path.join(models, 'EXP-NBD103_read_starts.dbw'),
                '-e', os.path.join(models, 'EXP-NBD103_read_ends.dbw')]
        out = io.StringIO()
        stdout, sys.stdout = sys.stdout, out
        t0, c0 = time.perf_counter(), time.process_time()
        try:
            cli.main(argv)
        finally:
            sys.stdout = stdout
        wall, cpu = time.perf_counter() - t0, time.process_time() - c0
        n = opts.files * opts.reads
        written = sum(len(files) for _, _, files in os.walk(out_dir)) - 1
        size = sum(os.path.getsize(os.path.join(d, f)) for d, _, files in os.walk(out_dir)
                   for f in files)
        print(json.dumps({'containers': opts.files, 'reads': n, 'binning': not opts.table_only,
                          'files_written': written, 'MB_written': round(size / 1e6, 1),
                          'seconds (incl. model loading)': round(wall, 2),
                          'reads_per_s': round(n / wall),
                          'host_cpu_us_per_read': round(cpu / n * 1e6, 1)}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()

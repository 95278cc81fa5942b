#!/usr/bin/env python3
"""`deepbinner classify` (start + end models) over a directory of one-read fast5 files, end to
end: the CPU loader's path against the path that hands the Signals to the GPU as stored
(classify.raw_inflate_share).  The files: written by the package's writer like the reads of
tools/multi_read_rate.py (log-normal lengths, mean 27 k samples, gzip 1), N distinct ones linked
over and over.

    python tools/classify_rate.py [n_files=40000] [distinct=4000]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time
import uuid

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepbinner_amd import classify, hdf5_write                    # noqa: E402

MODELS = os.path.join(REPO, 'deepbinner_amd', 'models')


def main():
    n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    rng = np.random.default_rng(7)
    out = {'files': n_files, 'distinct_files': distinct, 'usable_cpus': classify.usable_cpus()}
    with tempfile.TemporaryDirectory() as tmp:
        originals = []
        for k in range(distinct):
            n = int(np.clip(rng.lognormal(np.log(27000) - 0.32, 0.8), 2000, 400000))
            levels = rng.normal(450, 80, size=n // 8 + 1)
            signal = np.clip(np.rint(np.repeat(levels, 8)[:n] + rng.normal(0, 8, size=n)), 0, 2047)
            path = os.path.join(tmp, 'orig_%05d.fast5' % k)
            if os.path.lexists(path):
                os.unlink(path)          # (the writer never overwrites: a re-run starts clean)
            hdf5_write.write_single_read_fast5(path, str(uuid.UUID(bytes=rng.bytes(16), version=4)),
                                               signal.astype(np.int16))
            originals.append(path)
        os.makedirs(os.path.join(tmp, 'in'))
        files = []
        for i in range(n_files):
            dst = os.path.join(tmp, 'in', 'read_%06d.fast5' % i)
            os.symlink(originals[i % distinct], dst)
            files.append(dst)
        out['mean_file_KB'] = round(sum(os.path.getsize(p) for p in originals) / distinct / 1e3, 1)
        sm, si, em, ei, osz, _ = classify.load_and_check_models(
            os.path.join(MODELS, 'EXP-NBD103_read_starts.dbw'),
            os.path.join(MODELS, 'EXP-NBD103_read_ends.dbw'), 6144, out_dest=io.StringIO())
        args = argparse.Namespace(verbose=False, batch_size=256, scan_size=6144, score_diff=0.5,
                                  require_either=True, require_start=False, require_both=False,
                                  loader_procs=0)
        tables = {}
        for label, env in (('CPU loader', {'DEEPBINNER_GPU_INFLATE': '0'}),
                           ('Signals as stored, GPU inflating its share', {})):
            for name in ('DEEPBINNER_GPU_INFLATE',):
                os.environ.pop(name, None)
            os.environ.update(env)
            best = None
            for _ in range(2):
                sink = io.StringIO()
                t0, c0 = time.perf_counter(), time.process_time()
                with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(io.StringIO()):
                    calls, _ = classify.classify_fast5_files(files, sm, si, em, ei, osz, args,
                                                             verified_single_read=True)
                dt, cpu = time.perf_counter() - t0, time.process_time() - c0
                if best is None or dt < best[0]:
                    best = (dt, cpu)
            tables[label] = sorted(sink.getvalue().splitlines())
            share = classify.raw_inflate_share(sm, em, args, n_files,
                                               classify.device_replicas(sm, em))
            out[label] = {'seconds': round(best[0], 3), 'reads_per_s': round(n_files / best[0]),
                          'host_cpu_us_per_read': round(best[1] / n_files * 1e6, 1),
                          'host_inflate_share_per_cent': 100 if share is None else share}
        out['same_table'] = tables['CPU loader'] == tables['Signals as stored, GPU inflating its share']
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

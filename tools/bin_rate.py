#!/usr/bin/env python3
"""Rate of ``deepbinner bin`` on a synthetic FASTQ (N reads of L bases, 13 classes), optionally
beside the reference's own ``bin_reads`` (``--reference /root/reference``: build container only).

  python tools/bin_rate.py [--reads 50000] [--bases 4000] [--threads 0] [--reference DIR]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time
import types
import uuid

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepbinner_amd import bin as binner      # noqa: E402


def synthesise(folder, n_reads, n_bases):
    rng = np.random.default_rng(7)
    reads, table = os.path.join(folder, 'reads.fastq'), os.path.join(folder, 'classes.tsv')
    bases = np.frombuffer(b'ACGT', dtype=np.uint8)
    with open(reads, 'wb') as f, open(table, 'wt') as t:
        t.write('read_ID\tbarcode_call\n')
        for k in range(n_reads):
            read_id = str(uuid.UUID(bytes=rng.bytes(16), version=4))
            call = int(rng.integers(0, 13))
            t.write('{}\t{}\n'.format(read_id, call if call else 'none'))
            seq = bases[rng.integers(0, 4, size=n_bases)].tobytes()
            qual = rng.integers(35, 70, size=n_bases, dtype=np.uint8).tobytes()
            f.write(b'@' + read_id.encode() + b' runid=0 ch=%d\n' % (k % 512) + seq + b'\n+\n'
                    + qual + b'\n')
    return reads, table


def timed(fn, args):
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        fn(args)
        return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reads', type=int, default=50000)
    ap.add_argument('--bases', type=int, default=4000)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--reference', default=None)
    opts = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        reads, table = synthesise(tmp, opts.reads, opts.bases)
        size = os.path.getsize(reads)
        out = {'reads': opts.reads, 'bases_per_read': opts.bases, 'fastq_bytes': size,
               'host_threads': os.cpu_count()}
        seconds = timed(binner.bin_reads, types.SimpleNamespace(
            classes=table, reads=reads, out_dir=os.path.join(tmp, 'mine'), threads=opts.threads))
        out['deepbinner_amd'] = {'seconds': seconds, 'reads_per_s': opts.reads / seconds,
                                 'MB_per_s': size / seconds / 1e6}
        if opts.reference:
            sys.path.insert(0, opts.reference)
            import deepbinner.bin as ref_bin
            seconds = timed(ref_bin.bin_reads, types.SimpleNamespace(
                classes=table, reads=reads, out_dir=os.path.join(tmp, 'ref')))
            out['reference'] = {'seconds': seconds, 'reads_per_s': opts.reads / seconds,
                                'MB_per_s': size / seconds / 1e6}
        print(json.dumps(out))


if __name__ == '__main__':
    main()

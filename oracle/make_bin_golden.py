#!/usr/bin/env python3
"""
ORACLE tooling - fixtures for the ``bin`` command under ``tests/golden/bin/``.  Runs ONLY in the
build container, where /root/reference is mounted: it imports the reference's own
``deepbinner/bin.py`` (standard library only) and runs its ``bin_reads`` on seeded synthetic
inputs; what is committed is data - the inputs and what the reference wrote for them.

Inputs (all seeded):
  reads.fastq / reads.fasta   150 records; headers in the styles basecallers write (id first, id
                              inside a longer header, upper-case id); the last FASTQ record has no
                              trailing newline
  classes.tsv                 what ``deepbinner classify`` prints: header row, then read_id<TAB>call
  classes_verbose.tsv         the same with probability columns behind the call, a short row and
                              one read id that is not a UUID (the reference warns about it)
Expected (expected.json): per case the reference's stdout (progress lines removed) and the
length and SHA-256 of the decompressed contents of every file it wrote.
"""
import contextlib
import gzip
import hashlib
import io
import json
import os
import re
import sys
import tempfile
import types
import uuid

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden', 'bin')


def make_inputs():
    rng = np.random.default_rng(20260928)
    ids = [str(uuid.UUID(bytes=rng.bytes(16), version=4)) for _ in range(150)]
    ids = [i.upper() if k % 3 == 2 else i for k, i in enumerate(ids)]   # lookups are case sensitive
    calls = [('none' if c == 0 else str(c)) for c in rng.choice([0, 0, 1, 2, 3, 7, 12], size=150)]
    fastq, fasta = [], []
    for k, read_id in enumerate(ids):
        n = int(rng.integers(40, 160))
        seq = ''.join(rng.choice(list('ACGT'), size=n))
        qual = ''.join(chr(int(q)) for q in rng.integers(35, 70, size=n))
        style = k % 3
        if style == 0:
            header = '{} runid=0a1b2c read={} ch={} start_time=2018-05-11T04:5{}:00Z'.format(
                read_id, k, 1 + k % 512, k % 10)
        elif style == 1:
            header = 'read_{}_{} template'.format(k, read_id)
        else:
            header = read_id
        fastq.append('@{}\n{}\n+\n{}\n'.format(header, seq, qual))
        fasta.append('>{}\n{}\n'.format(header, seq))
    fastq[-1] = fastq[-1][:-1]                       # no newline at the end of the file
    plain = ['read_ID\tbarcode_call\n'] + ['{}\t{}\n'.format(i, c) for i, c in zip(ids, calls)]
    verbose = ['read_ID\tbarcode_call\tnone\t1\t2\n', '\n', 'orphan\n']
    for i, c in zip(ids, calls):
        verbose.append('{}\t{}\t0.10\t0.85\t0.05\n'.format(i, c))
    verbose.append('not-a-uuid\t5\t0.0\t1.0\t0.0\n')
    return {'reads.fastq': ''.join(fastq), 'reads.fasta': ''.join(fasta),
            'classes.tsv': ''.join(plain), 'classes_verbose.tsv': ''.join(verbose)}


def run_reference(ref_bin, classes, reads):
    with tempfile.TemporaryDirectory() as tmp:
        out_dir = os.path.join(tmp, 'binned')
        args = types.SimpleNamespace(classes=classes, reads=reads, out_dir=out_dir)
        stdout = io.StringIO()
        with contextlib.redirect_stdout(stdout):
            ref_bin.bin_reads(args)
        files = {}
        for name in sorted(os.listdir(out_dir)):
            with gzip.open(os.path.join(out_dir, name), 'rb') as f:
                data = f.read()
            files[name] = {'bytes': len(data), 'sha256': hashlib.sha256(data).hexdigest(),
                           'records': data.count(b'\n@' if b'fastq' in name.encode() else b'>')}
        text = re.sub(r'Writing reads: [\d,]+ \r', '', stdout.getvalue())
        return {'stdout': text.replace(out_dir, '<OUT>'), 'files': files}


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, text in make_inputs().items():
        with open(os.path.join(OUT, name), 'wt', newline='') as f:
            f.write(text)
    with gzip.GzipFile(os.path.join(OUT, 'reads.fastq.gz'), 'wb', mtime=0) as f:
        f.write(open(os.path.join(OUT, 'reads.fastq'), 'rb').read())
    sys.path.insert(0, REF)
    import deepbinner.bin as ref_bin
    expected = {}
    for case, (classes, reads) in {
            'fastq': ('classes.tsv', 'reads.fastq'),
            'fasta': ('classes.tsv', 'reads.fasta'),
            'fastq_gz_verbose_table': ('classes_verbose.tsv', 'reads.fastq.gz')}.items():
        expected[case] = dict(run_reference(ref_bin, os.path.join(OUT, classes),
                                            os.path.join(OUT, reads)),
                              classes=classes, reads=reads)
        print(case, {k: v['bytes'] for k, v in expected[case]['files'].items()})
    with open(os.path.join(OUT, 'expected.json'), 'wt') as f:
        json.dump(expected, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()

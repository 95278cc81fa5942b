"""``deepbinner bin`` against what the reference's own ``bin_reads`` wrote for the same inputs
(tests/golden/bin/, made by oracle/make_bin_golden.py) - host only."""
import argparse
import gzip
import hashlib
import json
import os
import re

import pytest

from conftest import REPO
from deepbinner_amd import bin as binner
from deepbinner_amd import deepbinner as cli

GOLD = os.path.join(REPO, 'tests', 'golden', 'bin')
EXPECTED = json.load(open(os.path.join(GOLD, 'expected.json')))


def run(capsys, classes, reads, out_dir, **kw):
    args = argparse.Namespace(classes=str(classes), reads=str(reads), out_dir=str(out_dir), **kw)
    binner.bin_reads(args)
    text = re.sub(r'Writing reads: [\d,]+ \r', '', capsys.readouterr().out)
    return text.replace(str(out_dir), '<OUT>')


def contents(out_dir):
    found = {}
    for name in sorted(os.listdir(out_dir)):
        with gzip.open(os.path.join(out_dir, name), 'rb') as f:
            found[name] = f.read()
    return found


@pytest.mark.parametrize('case', sorted(EXPECTED))
def test_same_files_and_messages_as_the_reference(case, tmp_path, capsys):
    want = EXPECTED[case]
    out_dir = tmp_path / 'binned'
    text = run(capsys, os.path.join(GOLD, want['classes']), os.path.join(GOLD, want['reads']),
               out_dir)
    assert text == want['stdout']
    got = contents(out_dir)
    assert sorted(got) == sorted(want['files'])
    for name, data in got.items():
        assert len(data) == want['files'][name]['bytes'], name
        assert hashlib.sha256(data).hexdigest() == want['files'][name]['sha256'], name


def test_many_gzip_members_keep_their_order(tmp_path, capsys, monkeypatch):
    """Small members and several threads: what comes out is the input, dealt out in order."""
    monkeypatch.setattr(binner, 'MEMBER_BYTES', 700)
    monkeypatch.setattr(binner, 'READ_BLOCK', 4099)       # records straddle the block edges
    monkeypatch.setattr(binner, 'MAX_PENDING', 2)
    want = EXPECTED['fastq']
    out_dir = tmp_path / 'binned'
    run(capsys, os.path.join(GOLD, 'classes.tsv'), os.path.join(GOLD, 'reads.fastq'), out_dir,
        threads=5)
    for name, data in contents(out_dir).items():
        assert hashlib.sha256(data).hexdigest() == want['files'][name]['sha256'], name
    raw = open(os.path.join(str(out_dir), 'barcode01.fastq.gz'), 'rb').read()
    assert raw.count(b'\x1f\x8b\x08') > 5


def test_reads_missing_from_the_table_are_counted_and_left_out(tmp_path, capsys):
    rows = open(os.path.join(GOLD, 'classes.tsv')).read().splitlines(keepends=True)
    table = tmp_path / 'some.tsv'
    table.write_text(''.join(rows[:101]))                  # header + the first 100 reads
    out_dir = tmp_path / 'binned'
    text = run(capsys, table, os.path.join(GOLD, 'reads.fastq'), out_dir)
    assert re.search(r'^  not found\s+50\s*$', text, re.M)
    assert 'Writing reads: 150 ' in text
    total = sum(data.count(b'\n+\n') for data in contents(out_dir).values())
    assert total == 100


def test_refusals(tmp_path, capsys):
    reads, classes = os.path.join(GOLD, 'reads.fastq'), os.path.join(GOLD, 'classes.tsv')

    def fails(message, **kw):
        args = dict(classes=classes, reads=reads, out_dir=str(tmp_path / 'o'))
        args.update(kw)
        with pytest.raises(SystemExit) as e:
            binner.bin_reads(argparse.Namespace(**args))
        assert message in str(e.value), str(e.value)
        capsys.readouterr()

    fails('does not exist', classes=str(tmp_path / 'nope.tsv'))
    fails('could not find', reads=str(tmp_path / 'nope.fastq'))
    bad = tmp_path / 'bad.tsv'
    bad.write_text('read_ID\tbarcode_call\nabc\tseven\n')
    fails('Error: read abc has a non-integer bin of seven', classes=str(bad))
    other = tmp_path / 'reads.txt'
    other.write_text('hello\n')
    fails('could not determine file format', reads=str(other))
    bz = tmp_path / 'reads.bz2'
    bz.write_bytes(b'BZh91AY')
    fails('cannot use bzip2 format', reads=str(bz))
    zipped = tmp_path / 'reads.zip'
    zipped.write_bytes(b'PK\x03\x04rest')
    fails('cannot use zip format', reads=str(zipped))
    a_file = tmp_path / 'a_file'
    a_file.write_text('x')
    fails('is an existing file', out_dir=str(a_file))
    anonymous = tmp_path / 'anon.fastq'
    anonymous.write_text('@read1\nACGT\n+\n!!!!\n')
    fails('could not find read ID in header: @read1', reads=str(anonymous))
    lines = open(reads, 'rb').read().split(b'\n')
    for k, n_lines in enumerate((6, 7)):                   # 1.5 records; 1.75 records
        cut = tmp_path / ('cut%d.fastq' % k)
        cut.write_bytes(b'\n'.join(lines[:n_lines]) + b'\n')
        fails('ends in the middle of a record', reads=str(cut), out_dir=str(tmp_path / ('c%d' % k)))
    # an output that exists - gzipped or not - is never overwritten
    for k, name in enumerate(('barcode01.fastq', 'unclassified.fastq.gz')):
        out_dir = tmp_path / ('exists%d' % k)
        out_dir.mkdir()
        (out_dir / name).write_text('keep me')
        fails('{} already exists'.format(out_dir / name), out_dir=str(out_dir))
        assert (out_dir / name).read_text() == 'keep me'


def test_command_line(tmp_path, capsys):
    out_dir = tmp_path / 'binned'
    cli.main(['bin', '--classes', os.path.join(GOLD, 'classes.tsv'),
              '--reads', os.path.join(GOLD, 'reads.fasta'), '--out_dir', str(out_dir)])
    assert 'barcode12' in capsys.readouterr().out
    assert sorted(os.listdir(str(out_dir))) == sorted(EXPECTED['fasta']['files'])
    with pytest.raises(SystemExit):
        cli.main(['bin', '--classes', 'x'])
    with pytest.raises(SystemExit) as e:
        cli.main(['train'])
    assert 'not part of this build' in str(e.value)

"""Argument handling of the classify / realtime commands (reference deepbinner.py:283-345)."""
import argparse
import os

import pytest

from deepbinner_amd import deepbinner as cli


def ns(**kw):
    base = dict(native=False, rapid=False, start_model=None, end_model=None, score_diff=0.5,
                require_either=False, require_start=False, require_both=False)
    base.update(kw)
    return argparse.Namespace(**base)


def test_native_preset_defaults_to_require_either():
    args = ns(native=True)
    cli.check_classify_and_realtime_arguments(args)
    assert os.path.basename(args.start_model).startswith('EXP-NBD103_read_starts')
    assert os.path.basename(args.end_model).startswith('EXP-NBD103_read_ends')
    assert args.require_either and not args.require_start and not args.require_both


def test_rapid_preset():
    args = ns(rapid=True)
    cli.check_classify_and_realtime_arguments(args)
    assert os.path.basename(args.start_model).startswith('SQK-RBK004_read_starts')
    assert args.end_model is None


@pytest.mark.parametrize('kw,msg', [
    (dict(native=True, rapid=True), 'only use one model preset'),
    (dict(native=True, start_model='x'), 'cannot explicitly specify a model'),
    (dict(), 'must provide at least one model'),
    (dict(start_model='x', score_diff=0.0), '--score_diff must be in the range (0, 1]'),
    (dict(start_model='x', score_diff=1.5), '--score_diff must be in the range (0, 1]'),
    (dict(start_model='x', require_both=True), '--require_both can only be used with two models'),
    (dict(start_model='x', end_model='y', require_both=True, require_start=True),
     'only one of the following options'),
])
def test_argument_errors(kw, msg):
    with pytest.raises(SystemExit) as e:
        cli.check_classify_and_realtime_arguments(ns(**kw))
    assert msg in str(e.value)


def test_parser_flags_and_defaults(monkeypatch):
    seen = {}
    import deepbinner_amd.classify as classify
    monkeypatch.setattr(classify, 'classify', lambda args: seen.update(vars(args)))
    cli.main(['classify', '--native', '--verbose', '--omp_num_threads', '4', 'some_dir'])
    assert seen['input'] == 'some_dir' and seen['verbose'] and seen['batch_size'] == 256
    assert seen['scan_size'] == 6144
    cli.main(['classify', '--rapid', '--scan_size', '3072', 'd'])
    assert seen['scan_size'] == 3072.0 and isinstance(seen['scan_size'], float)   # type=float
    assert seen['score_diff'] == 0.5 and seen['intra_op_parallelism_threads'] == 12
    with pytest.raises(SystemExit):
        cli.main([])
    with pytest.raises(SystemExit) as e:
        cli.main(['train', '--x'])
    assert 'not part of this build' in str(e.value)


def test_realtime_moves_files(oracle_backend, tmp_path, capsys, monkeypatch):
    """realtime --stop over a copy of the seven single-read files (reference realtime.py:28-150)."""
    import shutil
    from conftest import GOLD, MODEL_DIR
    import deepbinner_amd.realtime as realtime
    monkeypatch.setattr(realtime, 'POLL_SECONDS', 0)
    in_dir, out_dir = tmp_path / 'in', tmp_path / 'out'
    shutil.copytree(os.path.join(GOLD, 'fast5', 'single'), in_dir)
    args = argparse.Namespace(in_dir=str(in_dir), out_dir=str(out_dir), stop=True,
                              start_model=os.path.join(MODEL_DIR, 'EXP-NBD103_read_starts.dbw'),
                              end_model=None, scan_size=6144.0, score_diff=0.5, batch_size=4,
                              require_either=False, require_start=False, require_both=False)
    realtime.realtime(args)
    out = capsys.readouterr().out
    assert 'Found 7 fast5 files' in out and 'Barcode     Count' in out
    binned = {d: len(os.listdir(out_dir / d)) for d in os.listdir(out_dir)}
    assert binned == {'barcode01': 2, 'barcode02': 2, 'barcode03': 2, 'barcode12': 1}
    assert list(in_dir.glob('*.fast5')) == []


def test_realtime_multi_read_direct(oracle_backend, tmp_path, capsys, monkeypatch):
    import shutil
    from conftest import GOLD, MODEL_DIR
    import deepbinner_amd.realtime as realtime
    monkeypatch.setattr(realtime, 'POLL_SECONDS', 0)
    monkeypatch.setattr(shutil, 'which', lambda name: None)
    in_dir, out_dir = tmp_path / 'in', tmp_path / 'out'
    shutil.copytree(os.path.join(GOLD, 'fast5', 'multi'), in_dir)
    args = argparse.Namespace(in_dir=str(in_dir), out_dir=str(out_dir), stop=True,
                              start_model=os.path.join(MODEL_DIR, 'SQK-RBK004_read_starts.dbw'),
                              end_model=None, scan_size=6144.0, score_diff=0.5, batch_size=16,
                              require_either=False, require_start=False, require_both=False)
    realtime.realtime(args)
    rows = open(out_dir / 'multi_read_classifications.tsv').read().splitlines()
    assert len(rows) == 30 and all(len(r.split('\t')) == 3 for r in rows)


# ---- the reference's own command line, end to end (oracle/make_cli_golden.py) -------------------
def reference_cli_cases():
    import json
    from conftest import GOLD
    with open(os.path.join(GOLD, 'reference_cli.json')) as f:
        return json.load(f)


def run_like_the_reference(case, capsys):
    """Our command line on the argv the reference was run with (models mapped to the .dbw
    files): (header, sorted rows, the summary table's tokens)."""
    from conftest import MODEL_DIR, REPO
    argv = []
    for a in case['argv']:
        if a.startswith('MODELS/'):
            a = os.path.join(MODEL_DIR, a[len('MODELS/'):] + '.dbw')
        elif a.startswith('tests/'):
            a = os.path.join(REPO, a)
        argv.append(a)
    capsys.readouterr()
    cli.main(argv)
    captured = capsys.readouterr()
    rows = captured.out.splitlines()
    assert 'Barcode     Count' in captured.err
    return rows[0], sorted(rows[1:]), captured.err.split('Barcode     Count')[-1].split()


def classify_cases():
    return sorted(k for k, v in reference_cli_cases().items() if 'argv' in v)


def run_realtime_like_the_reference(tmp_path, capsys, monkeypatch):
    """Our `realtime --stop` with both NBD103 models on a copy of the single-read fixtures:
    (stdout with the paths made relative, the tree under out_dir, what is left in in_dir)."""
    import shutil
    from conftest import GOLD, MODEL_DIR
    import deepbinner_amd.realtime as realtime
    monkeypatch.setattr(realtime, 'POLL_SECONDS', 0)
    in_dir, out_dir = tmp_path / 'in', tmp_path / 'out'
    shutil.copytree(os.path.join(GOLD, 'fast5', 'single'), in_dir)
    capsys.readouterr()
    cli.main(['realtime', '--in_dir', str(in_dir), '--out_dir', str(out_dir), '--stop',
              '-s', os.path.join(MODEL_DIR, 'EXP-NBD103_read_starts.dbw'),
              '-e', os.path.join(MODEL_DIR, 'EXP-NBD103_read_ends.dbw')])
    text = capsys.readouterr().out.replace(str(tmp_path), '<WORK>') \
        .replace(MODEL_DIR + '/', 'MODELS/').replace('.dbw', '')
    tree = {d: sorted(os.listdir(out_dir / d)) for d in sorted(os.listdir(out_dir))}
    return text, tree, sorted(os.listdir(in_dir))


def test_realtime_as_the_reference_runs_it(oracle_backend, tmp_path, capsys, monkeypatch):
    """The reference's realtime.py run end to end by oracle/make_cli_golden.py (same stand-in for
    Keras as above): same stdout - progress lines included - and the same files in the same bins."""
    want = reference_cli_cases()['realtime_two_models']
    for k, reader in enumerate(('python', 'native')):
        monkeypatch.setenv('DEEPBINNER_FAST5_READER', reader)
        text, tree, left = run_realtime_like_the_reference(tmp_path / str(k), capsys, monkeypatch)
        assert text == want['stdout']
        assert tree == want['tree'] and left == want['left_in_in_dir']


@pytest.mark.parametrize('name', classify_cases())
def test_same_table_as_the_reference_command_line(name, oracle_backend, capsys, monkeypatch):
    """tests/golden/reference_cli.json: stdout and summary of the reference's own deepbinner.py +
    classify.py + load_fast5s.py (on h5py) for thirteen command lines (two of them on a
    training-data text file), with only model.predict
    replaced (by the oracle's network).  Same header, same rows (probabilities to two decimals,
    per-model calls, final call), same summary - here with the oracle behind seam b1 too."""
    for reader in ('python', 'native'):
        monkeypatch.setenv('DEEPBINNER_FAST5_READER', reader)
        case = reference_cli_cases()[name]
        header, rows, summary = run_like_the_reference(case, capsys)
        assert header == case['header']
        assert rows == case['rows']
        assert summary == case['summary']


def test_the_readme_walkthrough(oracle_backend, tmp_path, capsys, monkeypatch):
    """BASELINE.json configs[0]: sample_reads.tar.gz through classify and bin, as the reference's
    own code did it (oracle/make_cli_golden.py): same table, same binned FASTQ files, same words."""
    from conftest import check_the_readme_walkthrough, run_the_readme_walkthrough
    want = reference_cli_cases()['sample_reads_walkthrough']
    for k, reader in enumerate(('python', 'native')):
        monkeypatch.setenv('DEEPBINNER_FAST5_READER', reader)
        work = tmp_path / str(k)
        work.mkdir()
        check_the_readme_walkthrough(run_the_readme_walkthrough(work, capsys), want)


def test_training_data_outside_int16_is_an_error_with_a_line_number(oracle_backend, tmp_path,
                                                                     monkeypatch):
    """the device path carries int16; an out-of-range value in a training-data file is
    reported as an Error naming the line instead of a traceback from deep inside."""
    import deepbinner_amd.classify as classify

    class Int16Only:            # stands for the device model: it has the packed entry point
        def __init__(self, w):
            self.inner = __import__('conftest').OracleModel(w)
            self.inputs, self.outputs = self.inner.inputs, self.inner.outputs

        def predict(self, x, batch_size=None):
            return self.inner.predict(x)

    monkeypatch.setattr(classify, 'build_model', lambda w: Int16Only(w))
    from conftest import MODEL_DIR
    path = tmp_path / 'train.txt'
    ok = ','.join(str(400 + (i * 37) % 200) for i in range(1100))
    bad = ','.join(str(400 + (i * 37) % 200) for i in range(500)) + ',70000,' + ok
    path.write_text('1\t%s\n2\t%s\n' % (ok, bad))
    with pytest.raises(SystemExit) as e:
        cli.main(['classify', '-s', os.path.join(MODEL_DIR, 'EXP-NBD103_read_starts.dbw'),
                  '--scan_size', '1024', str(path)])
    assert 'line 2' in str(e.value) and 'int16' in str(e.value)

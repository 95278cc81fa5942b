import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, 'tests', 'golden')
MODEL_DIR = os.path.join(REPO, 'deepbinner_amd', 'models')
MODELS = ['EXP-NBD103_read_starts', 'EXP-NBD103_read_ends', 'SQK-RBK004_read_starts']
# (model, side) pairs the golden merged vectors exist for
PLAN = [('EXP-NBD103_read_starts', 'start'), ('EXP-NBD103_read_ends', 'end'),
        ('SQK-RBK004_read_starts', 'start')]


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def gold():
    with open(os.path.join(GOLD, 'calls.json')) as f:
        calls = json.load(f)
    reads = np.load(os.path.join(GOLD, 'reads.npz'))

    def split(samples, offsets):
        return [samples[offsets[i]:offsets[i + 1]] for i in range(len(offsets) - 1)]

    return {
        'calls': calls,
        'files': [str(x) for x in reads['files']],
        'read_ids': [str(x) for x in reads['read_ids']],
        'signals': split(reads['samples'], reads['offsets']),
        'multi_read_ids': [str(x) for x in reads['multi_read_ids']],
        'multi_signals': split(reads['multi_samples'], reads['multi_offsets']),
        'dir': GOLD,
    }


@pytest.fixture(scope='session')
def all_signals(gold):
    return gold['signals'] + gold['multi_signals']


@pytest.fixture(scope='session')
def weights():
    from deepbinner_amd.model_format import ModelWeights
    return {m: ModelWeights.load(os.path.join(MODEL_DIR, m + '.dbw'))[0] for m in MODELS}


class OracleModel:
    """Test double for the model object at seam b1: predict() backed by the NumPy oracle."""

    def __init__(self, weights, dtype=np.float32):
        from deepbinner_amd.hip_backend import _TensorSpec
        self.weights = weights
        self.dtype = dtype
        self.inputs = [_TensorSpec((None, weights.input_size, 1))]
        self.outputs = [_TensorSpec((None, weights.n_classes))]

    def predict(self, x, batch_size=None):
        from oracle import network_ref
        x = np.asarray(x, dtype=np.float32)
        return network_ref.forward(self.weights, x, dtype=self.dtype).astype(np.float32)


@pytest.fixture()
def oracle_backend(monkeypatch):
    """Route classify.build_model to the oracle so host logic runs without a GPU."""
    import deepbinner_amd.classify as classify
    monkeypatch.setattr(classify, 'build_model', lambda w: OracleModel(w))
    return classify


@pytest.fixture(scope='session')
def hip():
    """The real backend; skips (not passes) when no GPU is visible."""
    from deepbinner_amd import hip_backend
    if hip_backend.device_count() < 1:
        pytest.skip('no HIP device visible')
    return hip_backend


@pytest.fixture(scope='session')
def hip_models(hip, weights):
    return {m: hip.HipModel(w, device=0) for m, w in weights.items()}


def run_the_readme_walkthrough(tmp_path, capsys):
    """README walk-through = BASELINE.json configs[0]: ``classify --native`` over the six fast5
    files of sample_reads.tar.gz into a table, then ``bin`` of its basecalled.fastq.gz with that
    table -> (table header, sorted rows, bin stdout, {file: (bytes, sha256)})."""
    import gzip
    import hashlib
    import re
    from conftest import GOLD
    from deepbinner_amd import deepbinner as command_line
    single = os.path.join(GOLD, 'fast5', 'single')
    fast5_dir = tmp_path / 'sample_reads'
    fast5_dir.mkdir()
    for name in sorted(os.listdir(single)):
        if name.startswith('5210_'):
            os.symlink(os.path.join(single, name), str(fast5_dir / name))
    capsys.readouterr()
    command_line.main(['classify', '--native', str(fast5_dir)])
    table_text = capsys.readouterr().out
    table = tmp_path / 'classifications'
    table.write_text(table_text)
    out_dir = tmp_path / 'binned'
    command_line.main(['bin', '--classes', str(table), '--out_dir', str(out_dir), '--reads',
                       os.path.join(GOLD, 'sample_reads', 'basecalled.fastq.gz')])
    text = re.sub(r'Writing reads: [\d,]+ \r', '', capsys.readouterr().out)
    files = {}
    for name in sorted(os.listdir(out_dir)):
        with gzip.open(str(out_dir / name), 'rb') as f:
            data = f.read()
        files[name] = (len(data), hashlib.sha256(data).hexdigest())
    rows = table_text.splitlines()
    return rows[0], sorted(rows[1:]), text.replace(str(tmp_path), '<WORK>'), files


def check_the_readme_walkthrough(result, want):
    header, rows, bin_stdout, files = result
    assert header == want['classify_header'] and rows == want['classify_rows']
    assert bin_stdout == want['bin_stdout']
    assert files == {n: (v['bytes'], v['sha256']) for n, v in want['bin_files'].items()}
    # reference README.md:128: "you should get two reads each from barcodes 1, 2 and 3"
    assert sorted(r.split('\t')[1] for r in rows) == ['1', '1', '2', '2', '3', '3']

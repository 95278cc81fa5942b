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

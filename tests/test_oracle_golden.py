"""
The oracle against the reference's own answers (CPU only).

Golden files under tests/golden/ were produced by oracle/make_golden.py by running the
REFERENCE's call_batch / normalise / find_signal_start_pos / combine_calls / print_output_header
(imported from /root/reference) around the oracle's predict, and checked there against every
assertion of the reference's tests/test_classify.py, test_combine_calls.py, test_load_fast5s.py.
"""
import os

import numpy as np
import pytest

from conftest import GOLD, PLAN
from oracle import classify_ref, network_ref


# reference tests/test_classify.py:115-121 and :134-140
EXPECTED_START = {'63c20e8e-9b10-4ede-9862-9a53eec3c512': '1',
                  '618f68a6-3a9a-45e1-afe0-845172b20349': '1',
                  '9bfcf22c-5654-4b4c-b8f7-d3cebd416338': '2',
                  '5ce8d6ab-8c24-43cc-808b-50fb336fda2f': '2',
                  '424bfd6b-576c-4e2c-bf86-604c771b5ec9': '3',
                  '177c3867-6812-4476-a6da-9e4d5c43b760': '3',
                  '2fbd86a4-029a-45cf-8f18-411d542572ba': '12'}
EXPECTED_END = dict(EXPECTED_START, **{'618f68a6-3a9a-45e1-afe0-845172b20349': 'none',
                                       '9bfcf22c-5654-4b4c-b8f7-d3cebd416338': 'none'})


def test_param_counts():
    # reference tests/test_network_architecture.py:36 and :46
    from deepbinner_amd.model_format import param_count
    assert param_count(13) == 107197
    assert param_count(25) == 107785


def test_loader_answers(gold):
    # reference tests/test_load_fast5s.py:46-49, 55-58, 69-72
    by_id = dict(zip(gold['read_ids'], gold['signals']))
    s = by_id['177c3867-6812-4476-a6da-9e4d5c43b760']
    assert (len(s), s[0], s[4950]) == (4971, 714, 396)
    s = by_id['9bfcf22c-5654-4b4c-b8f7-d3cebd416338']
    assert (len(s), s[0], s[4862]) == (4983, 493, 618)
    s = by_id['2fbd86a4-029a-45cf-8f18-411d542572ba']
    assert (len(s), s[0], s[5388]) == (5395, 505, 436)


@pytest.mark.parametrize('model_name,side', PLAN)
def test_oracle_reproduces_reference_call_batch(gold, all_signals, weights, model_name, side):
    w = weights[model_name]
    calls, probs = classify_ref.call_batch(
        lambda x: network_ref.forward(w, x.astype(np.float32), dtype=np.float32),
        all_signals, 1024, 6144, 0.5, side)
    assert calls == gold['calls']['%s/%s' % (model_name, side)]
    ref = np.load(os.path.join(GOLD, 'merged_%s_%s.npy' % (model_name, side)))
    assert np.abs(probs - ref).max() < 1e-6


def test_reference_test_assertions(gold):
    n = len(gold['read_ids'])
    start = dict(zip(gold['read_ids'], gold['calls']['EXP-NBD103_read_starts/start'][:n]))
    end = dict(zip(gold['read_ids'], gold['calls']['EXP-NBD103_read_ends/end'][:n]))
    assert start == EXPECTED_START
    assert end == EXPECTED_END
    either = {r: classify_ref.combine_calls(start[r], end[r], 'require_either') for r in start}
    both = {r: classify_ref.combine_calls(start[r], end[r], 'require_both') for r in start}
    assert either == EXPECTED_START      # tests/test_classify.py:154-160
    assert both == EXPECTED_END          # tests/test_classify.py:174-180


def test_windows_match_reference_normalise(gold, all_signals):
    for side in ('start', 'end'):
        ref = np.load(os.path.join(GOLD, 'windows_%s.npy' % side))    # float32 of float64
        mine = classify_ref.make_windows(all_signals, 1024, 6144, side).astype(np.float32)
        assert np.array_equal(mine, ref)


def test_trim_positions(gold):
    got = [classify_ref.find_signal_start_pos(s) for s in gold['signals']]
    assert got == gold['calls']['trim_start'] == [110, 235, 660, 85, 60, 285, 310]


def test_combine_table(gold):
    for key, want in gold['calls']['combine_table'].items():
        mode, s, e = key.split('|')
        assert classify_ref.combine_calls(s, e, mode) == want


def test_edge_semantics_matter():
    """The two TensorFlow edge rules the reference's own tests cannot see."""
    x = np.arange(1, 9, dtype=np.float64).reshape(1, 8, 1)
    # stride-2 SAME pads on the right: out[i] = x[2i] + x[2i+1] + x[2i+2]
    y = network_ref.conv1d(x, np.ones((3, 1, 1)), np.zeros(1), 2, 'same')
    assert y[0, :, 0].tolist() == [6.0, 12.0, 18.0, 15.0]
    a = network_ref.avg_pool3_same(x)
    assert a[0, 0, 0] == 1.5 and a[0, -1, 0] == 7.5 and a[0, 1, 0] == 2.0


def test_zero_window(gold, weights):
    for m, want in gold['calls']['zero_window_class0'].items():
        p = network_ref.forward(weights[m], np.zeros((1, 1024), np.float32), dtype=np.float64)
        assert abs(p[0, 0] - want) < 1e-12


def test_c_restatement_matches_numpy_oracle(gold, all_signals, weights):
    from oracle import dbref
    w = weights['EXP-NBD103_read_starts']
    m = dbref.CModel(w)
    x = np.load(os.path.join(GOLD, 'windows_start.npy')).reshape(-1, 1024)[::3]
    want = np.load(os.path.join(GOLD, 'window_probs_EXP-NBD103_read_starts_start.npy'))[::3]
    assert np.abs(m.predict(x) - want).max() < 5e-6
    offsets = np.zeros(len(all_signals) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(s) for s in all_signals])
    samples = np.concatenate(all_signals).astype(np.int16)
    for side in ('start', 'end'):
        ref = np.load(os.path.join(GOLD, 'windows_%s.npy' % side))   # [12, reads, 1024]
        got = m.windows(samples, offsets, side, 6144).reshape(len(all_signals), 12, 1024)
        assert np.abs(got - ref.transpose(1, 0, 2)).max() < 1e-6
    probs, calls = m.classify(samples, offsets, 'start', 6144, 0.5)
    want_calls = gold['calls']['EXP-NBD103_read_starts/start']
    assert ['none' if c == 0 else str(c) for c in calls] == want_calls
    ref = np.load(os.path.join(GOLD, 'merged_EXP-NBD103_read_starts_start.npy'))
    assert np.abs(probs - ref).max() < 1e-5


def test_the_reference_test_suite_passed_on_the_oracle():
    """tests/golden/reference_tests_report.json: the reference's OWN tests (its tests/ directory,
    unittest, run from its root) against the oracle's network behind a stand-in for Keras, with
    its classify.py / load_fast5s.py (h5py) untouched - produced by oracle/run_reference_tests.py in
    the build container.  Everything on the classify path passed; the one module that builds a
    real Keras graph could not be imported."""
    import json
    with open(os.path.join(GOLD, 'reference_tests_report.json')) as f:
        report = json.load(f)
    by_module = {}
    for test, outcome in report['outcomes'].items():
        by_module.setdefault(test.split('.')[1] if test.startswith('tests.') else test,
                             []).append(outcome)
    assert by_module['test_classify'] == ['ok'] * 14
    assert by_module['test_combine_calls'] == ['ok'] * 3
    assert by_module['test_load_fast5s'] == ['ok'] * 8
    others = {m: o for m, o in by_module.items()
              if m not in ('test_classify', 'test_combine_calls', 'test_load_fast5s')}
    assert all('test_network_architecture' in m for m in others), others


def test_the_reference_test_files_passed_on_this_package():
    """tests/golden/reference_tests_on_package.json: /root/reference/tests, unchanged, with the name
    ``deepbinner`` bound to deepbinner_amd and the oracle behind seam b1
    (oracle/run_reference_tests_on_package.py, build container): the 25 tests on the classify path
    pass - the drop-in claim checked by the reference's own assertions."""
    import json
    with open(os.path.join(GOLD, 'reference_tests_on_package.json')) as f:
        report = json.load(f)
    passed = [t for t, o in report['outcomes'].items() if o == 'ok']
    assert len(passed) == 25 and report['ran'] == 26
    assert sum(t.startswith('tests.test_classify.') for t in passed) == 14
    assert sum(t.startswith('tests.test_load_fast5s.') for t in passed) == 8
    assert sum(t.startswith('tests.test_combine_calls.') for t in passed) == 3
    assert list(report['details']) == ['unittest.loader._FailedTest.tests.test_network_architecture']

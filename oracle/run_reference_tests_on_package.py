#!/usr/bin/env python3
"""
ORACLE tooling - the reference's own test files, run against THIS package.

    python oracle/run_reference_tests_on_package.py [--backend oracle|hip]   (build container only)
The reference's tests import ``deepbinner.classify`` and ``deepbinner.load_fast5s``; here the name
``deepbinner`` is bound to ``deepbinner_amd`` (its modules have the reference's names), the working
directory is the reference's root (the tests use its ``tests/fast5_files`` and its Keras model
files, which this package reads with its own HDF5 reader) and unittest runs /root/reference/tests
unchanged.  With ``--backend oracle`` (default; no GPU here) seam b1 is the oracle's network, as in
this repository's CPU tests; ``hip`` uses the real backend where there is a GPU.
Writes tests/golden/reference_tests_on_package.json: every test id with its outcome - the drop-in
claim of SURVEY.md section 8b, checked by the reference's own assertions.
"""
import argparse
import importlib
import io
import json
import os
import sys
import unittest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
from run_reference_tests import Recorder      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', choices=('oracle', 'hip'), default='oracle')
    opts = ap.parse_args()
    import deepbinner_amd
    sys.modules['deepbinner'] = deepbinner_amd
    for name in ('classify', 'load_fast5s', 'trim_signal', 'misc', 'deepbinner', 'realtime', 'bin'):
        module = importlib.import_module('deepbinner_amd.' + name)
        sys.modules['deepbinner.' + name] = module
        setattr(deepbinner_amd, name, module)
    if opts.backend == 'oracle':
        from conftest import OracleModel
        import deepbinner_amd.classify as classify
        classify.build_model = lambda weights: OracleModel(weights)
    os.chdir('/root/reference')
    suite = unittest.defaultTestLoader.discover('tests', top_level_dir='/root/reference')
    runner = unittest.TextTestRunner(stream=io.StringIO(), resultclass=Recorder, verbosity=0)
    real_stdout, real_stderr = sys.stdout, sys.stderr
    sys.stdout, sys.stderr = io.StringIO(), io.StringIO()
    try:
        result = runner.run(suite)
    finally:
        sys.stdout, sys.stderr = real_stdout, real_stderr
    outcomes = dict(sorted(result.outcomes.items()))
    details = {t.id(): text.strip().splitlines()[-1] for t, text in result.errors + result.failures}
    report = {'backend': opts.backend, 'ran': result.testsRun,
              'ok': sum(v == 'ok' for v in outcomes.values()), 'outcomes': outcomes,
              'details': details}
    with open(os.path.join(REPO, 'tests', 'golden', 'reference_tests_on_package.json'), 'wt') as f:
        json.dump(report, f, indent=1)
    print(json.dumps({k: report[k] for k in ('backend', 'ran', 'ok', 'details')}, indent=1))


if __name__ == '__main__':
    main()

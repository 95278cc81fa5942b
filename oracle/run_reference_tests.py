#!/opt/conda/bin/python3.9
"""
ORACLE tooling - the reference's own test suite, run against the oracle's network.

    /opt/conda/bin/python3.9 oracle/run_reference_tests.py        (build container only)
Installs the same stand-in for Keras/TensorFlow as oracle/make_cli_golden.py (the model files are
read into oracle/network_ref.py; everything else - classify.py, load_fast5s.py on h5py,
network_architecture.py where it does not need Keras - is the reference's code) and runs
/root/reference/tests with unittest from the reference's root directory, as its README says.
Writes tests/golden/reference_tests_report.json: every test id with its outcome.  Tests that need
a real Keras graph (tests/test_network_architecture.py builds one) are expected to error and are
listed as such; the ones on the classify path must pass - that is the oracle's pin.
"""
import io
import json
import os
import sys
import unittest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'oracle'))
sys.path.insert(0, REPO)
import make_cli_golden      # noqa: E402  (for install_stand_ins)


class Recorder(unittest.TextTestResult):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.outcomes = {}

    def addSuccess(self, test):
        super().addSuccess(test)
        self.outcomes[test.id()] = 'ok'

    def addFailure(self, test, err):
        super().addFailure(test, err)
        self.outcomes[test.id()] = 'FAIL'

    def addError(self, test, err):
        super().addError(test, err)
        self.outcomes[test.id()] = 'error: ' + err[0].__name__

    def addSkip(self, test, reason):
        super().addSkip(test, reason)
        self.outcomes[test.id()] = 'skipped'


def main():
    make_cli_golden.install_stand_ins()
    os.chdir('/root/reference')
    sys.path.insert(0, '/root/reference')
    suite = unittest.defaultTestLoader.discover('tests', top_level_dir='/root/reference')
    stream = io.StringIO()
    runner = unittest.TextTestRunner(stream=stream, resultclass=Recorder, verbosity=0)
    real_stdout, real_stderr = sys.stdout, sys.stderr
    sys.stdout, sys.stderr = io.StringIO(), io.StringIO()      # the suite prints tables
    try:
        result = runner.run(suite)
    finally:
        sys.stdout, sys.stderr = real_stdout, real_stderr
    outcomes = dict(sorted(result.outcomes.items()))
    for test, _ in result.errors:          # import errors of whole modules have no test id run
        outcomes.setdefault(test.id(), 'error')
    report = {'ran': result.testsRun, 'ok': sum(v == 'ok' for v in outcomes.values()),
              'outcomes': outcomes}
    with open(os.path.join(REPO, 'tests', 'golden', 'reference_tests_report.json'), 'wt') as f:
        json.dump(report, f, indent=1)
    by_module = {}
    for test, outcome in outcomes.items():
        module = test.split('.')[1] if test.startswith('tests.') else test
        by_module.setdefault(module, []).append(outcome)
    for module, values in by_module.items():
        print(module, {v: values.count(v) for v in set(values)})


if __name__ == '__main__':
    main()

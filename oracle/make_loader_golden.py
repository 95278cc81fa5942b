#!/opt/conda/bin/python3.9
"""
ORACLE tooling - the reference's own loader run on every committed fast5 fixture.

Run with the image's conda interpreter (the only one with h5py), in the build container where
/root/reference is mounted:
    /opt/conda/bin/python3.9 oracle/make_loader_golden.py
It imports the reference's ``deepbinner/load_fast5s.py`` as it is (h5py + standard library) and
records, per file, what ``get_root_level_keys`` and ``get_read_id_and_signal`` return - the read
id, the signal length and a SHA-256 of the samples, or how the call ended (``multi`` for the
``sys.exit`` on multi-read files, ``none`` for (None, None), ``vlen`` where h5py 3 hands the
reference a ``str`` read id that it then tries to ``.decode()``: an ``AttributeError`` it does not
catch) - and what ``determine_single_or_multi_fast5s`` says about each fixture directory.
Output: tests/golden/loader_reference.json (data only).
"""
import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, '/root/reference')
from deepbinner import load_fast5s as ref      # noqa: E402

FAST5 = os.path.join(REPO, 'tests', 'golden', 'fast5')


def main():
    out = {'files': {}, 'directories': {}}
    for sub in sorted(os.listdir(FAST5)):
        folder = os.path.join(FAST5, sub)
        files = sorted(f for f in os.listdir(folder) if f.endswith('.fast5'))
        kinds = set()
        for name in files:
            path = os.path.join(folder, name)
            entry = {'root_keys': sorted(ref.get_root_level_keys(path))}
            try:
                read_id, signal = ref.get_read_id_and_signal(path)
                if read_id is None:
                    entry['result'] = 'none'
                else:
                    entry.update(result='read', read_id=read_id, n=int(len(signal)),
                                 dtype=str(signal.dtype),
                                 sha256=hashlib.sha256(np.ascontiguousarray(signal, dtype='<i2')
                                                       .tobytes()).hexdigest())
            except SystemExit as e:
                entry.update(result='multi', message=str(e))
            except AttributeError:
                entry['result'] = 'vlen'
            out['files'][sub + '/' + name] = entry
            kinds.add(ref.determine_single_or_multi_fast5s([path]))
        try:
            verdict = ref.determine_single_or_multi_fast5s(
                [os.path.join(folder, f) for f in files][:5])
        except SystemExit as e:
            verdict = 'exit: ' + str(e)
        out['directories'][sub] = {'first_five': verdict, 'per_file': sorted(kinds)}
    with open(os.path.join(REPO, 'tests', 'golden', 'loader_reference.json'), 'wt') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    results = [e['result'] for e in out['files'].values()]
    print({r: results.count(r) for r in set(results)}, out['directories'])


if __name__ == '__main__':
    main()

#!/opt/conda/bin/python3.9
"""
ORACLE tooling - what the real HDF5 library reads from the reference's Keras model files.

Run with the image's conda interpreter (h5py), in the build container where /root/reference is
mounted:   /opt/conda/bin/python3.9 oracle/make_model_golden.py
For each of ``models/EXP-NBD103_read_starts``, ``EXP-NBD103_read_ends``, ``SQK-RBK004_read_starts``
it records every dataset under ``model_weights`` as h5py returns it (path, shape, dtype, SHA-256
of the little-endian fp32 bytes), the ``keras_version`` / ``backend`` attributes, and a SHA-256 of
``model_config``.  Output: tests/golden/model_reference.json (data only).  The tests then check
that the shipped ``.dbw`` weight blobs (converted through this package's own HDF5 reader) hold
exactly those numbers, layer by layer.
"""
import hashlib
import json
import os

import h5py
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODELS = ['EXP-NBD103_read_starts', 'EXP-NBD103_read_ends', 'SQK-RBK004_read_starts']


def text(value):
    return value.decode() if isinstance(value, bytes) else str(value)


def main():
    out = {}
    for name in MODELS:
        entry = {'datasets': {}}
        with h5py.File(os.path.join('/root/reference/models', name), 'r') as f:
            entry['keras_version'] = text(f.attrs['keras_version'])
            entry['backend'] = text(f.attrs['backend'])
            config = text(f.attrs['model_config'])
            entry['model_config_sha256'] = hashlib.sha256(config.encode()).hexdigest()
            entry['layer_names'] = [l['name'] for l in json.loads(config)['config']['layers']]

            def visit(path, obj):
                if isinstance(obj, h5py.Dataset):
                    data = np.ascontiguousarray(obj[...], dtype='<f4')
                    entry['datasets'][path] = {
                        'shape': list(obj.shape), 'dtype': str(obj.dtype),
                        'sha256': hashlib.sha256(data.tobytes()).hexdigest()}
            f['model_weights'].visititems(visit)
        entry['n_parameters'] = int(sum(int(np.prod(d['shape'])) for d in entry['datasets'].values()))
        out[name] = entry
        print(name, len(entry['datasets']), 'datasets,', entry['n_parameters'], 'parameters')
    with open(os.path.join(REPO, 'tests', 'golden', 'model_reference.json'), 'wt') as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()

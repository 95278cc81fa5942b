"""
The Deepbinner network as data: the fixed layer table of ``build_network``
(reference ``deepbinner/network_architecture.py:18-95``), the flat fp32 weight blob the C-ABI
consumes (``include/deepbinner_hip.h``: ``dbh_model_create``), and import of the reference's
Keras-2.1.4 HDF5 model files (``models/*``, loaded by ``deepbinner/classify.py:90``) through
``hdf5_lite`` — no Keras/TensorFlow/h5py involved.

Flat blob layout (all little-endian fp32, "canonical order"):
    for conv i = 1..20:   kernel[k][C_in][C_out] (Keras layout), bias[C_out]
    for bn   i = 1..7:    gamma[C], beta[C], moving_mean[C], moving_variance[C]
For 13 classes that is 105,277 + 1,920 = 107,197 floats — the parameter count the reference pins
in ``tests/test_network_architecture.py:36``.

On-disk ``.dbw`` container (this project's own format, so the GPU box needs no HDF5 model file):
    b'DBW1' | u32 n_classes | u32 input_size | u32 n_floats | f32[n_floats]
"""

import json
import struct

import numpy as np

INPUT_SIZE = 1024
BN_EPSILON = 1e-3  # Keras BatchNormalization default, recorded in every model_config

# (name, kernel_size, C_in, C_out, stride, padding); C_out of conv1d_20 is the class count.
# network_architecture.py line numbers in the trailing comments.
CONV_LAYERS = [
    ('conv1d_1', 3, 1, 48, 2, 'same'),      # :28
    ('conv1d_2', 3, 48, 48, 1, 'same'),     # :34
    ('conv1d_3', 3, 48, 48, 1, 'same'),     # :35
    ('conv1d_4', 3, 48, 48, 1, 'same'),     # :36
    ('conv1d_5', 1, 48, 16, 1, 'valid'),    # :43
    ('conv1d_6', 3, 16, 48, 1, 'same'),     # :46
    ('conv1d_7', 3, 48, 48, 1, 'same'),     # :47
    ('conv1d_8', 3, 48, 48, 1, 'same'),     # :54
    ('conv1d_9', 3, 48, 48, 1, 'same'),     # :55
    ('conv1d_10', 1, 48, 48, 1, 'same'),    # :63  (after AveragePooling1D(3,1,same) :62)
    ('conv1d_11', 1, 48, 48, 1, 'same'),    # :64
    ('conv1d_12', 1, 48, 16, 1, 'same'),    # :65
    ('conv1d_13', 3, 16, 48, 1, 'same'),    # :66
    ('conv1d_14', 1, 48, 16, 1, 'same'),    # :67
    ('conv1d_15', 3, 16, 48, 1, 'same'),    # :68
    ('conv1d_16', 3, 48, 48, 1, 'same'),    # :69
    ('conv1d_17', 3, 192, 48, 2, 'same'),   # :77
    ('conv1d_18', 3, 48, 48, 1, 'same'),    # :83
    ('conv1d_19', 3, 48, 48, 1, 'same'),    # :84
    ('conv1d_20', 1, 48, None, 1, 'valid'),  # :91
]
BN_CHANNELS = [48, 48, 48, 48, 192, 48, 48]  # batch_normalization_1..7 (:30,39,50,58,73,79,87)

# Keras layer class sequence of the 45-layer graph (model_config order), used to recognise a
# genuine Deepbinner model file.
_EXPECTED_CLASSES = (
    ['InputLayer', 'GaussianNoise', 'Conv1D', 'BatchNormalization', 'Dropout'] +
    ['Conv1D'] * 3 + ['MaxPooling1D', 'BatchNormalization', 'Dropout'] +
    ['Conv1D'] * 3 + ['MaxPooling1D', 'BatchNormalization', 'Dropout'] +
    ['Conv1D'] * 2 + ['MaxPooling1D', 'BatchNormalization', 'Dropout'])


def conv_shapes(n_classes):
    out = []
    for name, k, cin, cout, stride, padding in CONV_LAYERS:
        out.append((name, k, cin, n_classes if cout is None else cout, stride, padding))
    return out


def param_count(n_classes):
    n = sum(k * cin * cout + cout for _, k, cin, cout, _, _ in conv_shapes(n_classes))
    return n + 4 * sum(BN_CHANNELS)


class ModelWeights:
    """Weights of one trained model in canonical order."""

    def __init__(self, n_classes, convs, bns, input_size=INPUT_SIZE):
        self.n_classes = int(n_classes)
        self.input_size = int(input_size)
        self.convs = convs  # list of (kernel[k,Cin,Cout] f32, bias[Cout] f32)
        self.bns = bns      # list of (gamma, beta, mean, var) f32
        shapes = conv_shapes(self.n_classes)
        if len(convs) != len(shapes) or len(bns) != len(BN_CHANNELS):
            raise ValueError('wrong number of layers for a Deepbinner model')
        for (kernel, bias), (name, k, cin, cout, _, _) in zip(convs, shapes):
            if kernel.shape != (k, cin, cout) or bias.shape != (cout,):
                raise ValueError('%s has shape %s, expected %s'
                                 % (name, kernel.shape, (k, cin, cout)))
        for bn, c in zip(bns, BN_CHANNELS):
            if any(a.shape != (c,) for a in bn):
                raise ValueError('batch-normalisation layer has the wrong channel count')

    def flat(self):
        parts = []
        for kernel, bias in self.convs:
            parts += [kernel.ravel(), bias.ravel()]
        for bn in self.bns:
            parts += [a.ravel() for a in bn]
        out = np.ascontiguousarray(np.concatenate(parts), dtype='<f4')
        assert out.size == param_count(self.n_classes)
        return out

    @classmethod
    def from_flat(cls, flat, n_classes, input_size=INPUT_SIZE):
        flat = np.asarray(flat, dtype='<f4').ravel()
        if flat.size != param_count(n_classes):
            raise ValueError('weight blob has %d floats, expected %d for %d classes'
                             % (flat.size, param_count(n_classes), n_classes))
        pos = 0
        convs, bns = [], []
        for _, k, cin, cout, _, _ in conv_shapes(n_classes):
            n = k * cin * cout
            kernel = flat[pos:pos + n].reshape(k, cin, cout).copy()
            pos += n
            bias = flat[pos:pos + cout].copy()
            pos += cout
            convs.append((kernel, bias))
        for c in BN_CHANNELS:
            bns.append(tuple(flat[pos + i * c:pos + (i + 1) * c].copy() for i in range(4)))
            pos += 4 * c
        return cls(n_classes, convs, bns, input_size)

    # -- .dbw container ----------------------------------------------------------------------
    def save(self, path):
        flat = self.flat()
        with open(path, 'wb') as f:
            f.write(b'DBW1')
            f.write(struct.pack('<III', self.n_classes, self.input_size, flat.size))
            f.write(flat.tobytes())

    @classmethod
    def load_dbw(cls, path):
        with open(path, 'rb') as f:
            head = f.read(16)
            if len(head) != 16 or head[:4] != b'DBW1':
                raise ValueError('not a .dbw weight file')
            n_classes, input_size, n_floats = struct.unpack('<III', head[4:])
            flat = np.frombuffer(f.read(4 * n_floats), dtype='<f4')
        if flat.size != n_floats:
            raise ValueError('truncated .dbw weight file')
        return cls.from_flat(flat, n_classes, input_size)

    # -- Keras HDF5 import ----------------------------------------------------------------------
    @classmethod
    def load_keras_hdf5(cls, path):
        """Read a Keras-2.1.4 model file as shipped in the reference's ``models/`` directory."""
        from . import hdf5_lite
        with hdf5_lite.File(path) as hf:
            if 'model_weights' not in hf.keys() or 'model_config' not in hf.attrs:
                raise ValueError('no model_weights/model_config in file')
            config = json.loads(hf.attrs['model_config'].decode('utf-8'))
            layers = config['config']['layers']
            input_shape = layers[0]['config'].get('batch_input_shape')
            check_architecture(layers)
            n_classes = [l for l in layers if l['name'] == 'conv1d_20'][0]['config']['filters']
            mw = hf['model_weights']
            convs, bns = [], []
            for i in range(1, 21):
                g = mw['conv1d_%d/conv1d_%d' % (i, i)]
                convs.append((g['kernel:0'][:].astype('<f4'), g['bias:0'][:].astype('<f4')))
            for i in range(1, 8):
                g = mw['batch_normalization_%d/batch_normalization_%d' % (i, i)]
                bns.append(tuple(g[n][:].astype('<f4') for n in
                                 ('gamma:0', 'beta:0', 'moving_mean:0', 'moving_variance:0')))
        return cls(n_classes, convs, bns, int(input_shape[1])), input_shape

    @classmethod
    def load(cls, path):
        """Load either container; returns (weights, input_shape) like the Keras loader."""
        with open(path, 'rb') as f:
            magic = f.read(4)
        if magic == b'DBW1':
            w = cls.load_dbw(path)
            return w, [None, w.input_size, 1]
        return cls.load_keras_hdf5(path)


def check_architecture(layers):
    """Raise ValueError unless ``layers`` (a Keras model_config layer list) is build_network()."""
    if len(layers) != 45:
        raise ValueError('expected 45 layers, found %d' % len(layers))
    classes = [l['class_name'] for l in layers]
    if classes[:len(_EXPECTED_CLASSES)] != _EXPECTED_CLASSES:
        raise ValueError('layer sequence does not match the Deepbinner architecture')
    by_name = {l['name']: l for l in layers}
    for name, k, cin, cout, stride, padding in CONV_LAYERS:
        cfg = by_name[name]['config']
        ok = (cfg['kernel_size'] == [k] and cfg['strides'] == [stride]
              and cfg['padding'] == padding and cfg['activation'] == 'relu'
              and cfg.get('use_bias', True) and cfg.get('dilation_rate', [1]) == [1]
              and (cout is None or cfg['filters'] == cout))
        if not ok:
            raise ValueError('%s does not match the Deepbinner architecture' % name)
    for i in range(1, 8):
        cfg = by_name['batch_normalization_%d' % i]['config']
        if abs(cfg['epsilon'] - BN_EPSILON) > 1e-12 or cfg['axis'] not in (-1, 2):
            raise ValueError('unexpected batch-normalisation settings')
    concat = by_name['concatenate_1']
    inbound = [x[0] for x in concat['inbound_nodes'][0]]
    if inbound != ['conv1d_10', 'conv1d_11', 'conv1d_13', 'conv1d_16']:
        raise ValueError('unexpected inception concatenation order')

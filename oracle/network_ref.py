"""
ORACLE — test infrastructure, not product code.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package; nothing under ``deepbinner_amd/``
does.

NumPy restatement of the inference graph that the reference builds in
``deepbinner/network_architecture.py:18-95`` and evaluates with ``model.predict``
(``deepbinner/classify.py:361``).  The arithmetic itself lives in un-vendored, unpinned
third-party packages (Keras 2.1.4 / TensorFlow 1.x per the model files' ``keras_version`` /
``backend`` attributes; ``requirements.txt:1-7``), which are not installed here, so this file
restates TensorFlow's published operator semantics:

* Conv1D is cross-correlation (no kernel flip), channels-last, kernel stored (k, C_in, C_out).
* ``padding='same'``: out = ceil(L/stride); pad_total = max((out-1)*stride + k - L, 0);
  pad_left = pad_total // 2, remainder on the RIGHT (so the two stride-2 convs pad right only).
* ``AveragePooling1D(3, strides=1, padding='same')`` divides by the number of VALID taps
  (TensorFlow avg_pool excludes padding), so edge outputs are (x0+x1)/2.
* ``MaxPooling1D(2)`` is 'valid' with stride 2.  BatchNormalization (inference):
  x * (gamma * rsqrt(var + eps)) + (beta - mean * gamma * rsqrt(var + eps)), eps = 1e-3.
* GaussianNoise and Dropout are identity at inference.

PINNING STATUS: barcode calls and 2-decimal probabilities are pinned against the reference's own
tests (``tests/test_classify.py:115-296``) — see ``tests/test_oracle_golden.py``.  At the 1e-4
probability level the reference holds no vector ("parity unpinned" there): the edge semantics above
rest on TensorFlow's documented SAME / avg_pool rules, cross-checked against an independent
torch-CPU evaluation of the same graph when the fixtures were generated (``make_golden.py``).
"""

import numpy as np

from deepbinner_amd.model_format import BN_EPSILON, conv_shapes


def conv1d(x, kernel, bias, stride, padding):
    """x [N, L, C_in] -> [N, L_out, C_out]; TensorFlow SAME/VALID rules."""
    n, length, _ = x.shape
    k = kernel.shape[0]
    if padding == 'same':
        out = -(-length // stride)
        pad_total = max((out - 1) * stride + k - length, 0)
        left = pad_total // 2
        right = pad_total - left
    else:
        out = (length - k) // stride + 1
        left = right = 0
    xp = np.pad(x, ((0, 0), (left, right), (0, 0))) if (left or right) else x
    y = np.empty((n, out, kernel.shape[2]), dtype=x.dtype)
    y[:] = bias
    span = (out - 1) * stride + 1
    for j in range(k):
        y += xp[:, j:j + span:stride, :] @ kernel[j]
    return y


def relu(x):
    return np.maximum(x, 0)


def max_pool2(x):
    n, length, c = x.shape
    half = length // 2
    return x[:, :2 * half, :].reshape(n, half, 2, c).max(axis=2)


def avg_pool3_same(x):
    """AveragePooling1D(pool_size=3, strides=1, padding='same'), TF valid-count divisor."""
    n, length, c = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (0, 0)))
    s = xp[:, 0:length] + xp[:, 1:length + 1] + xp[:, 2:length + 2]
    count = np.full((length, 1), 3.0, dtype=x.dtype)
    count[0] = 2.0
    count[-1] = 2.0
    if length == 1:
        count[0] = 1.0
    return s / count


def batch_norm(x, bn, dtype):
    gamma, beta, mean, var = (a.astype(dtype) for a in bn)
    scale = gamma / np.sqrt(var + dtype(BN_EPSILON))
    shift = beta - mean * scale
    return x * scale + shift


def softmax(x):
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def forward(weights, x, dtype=np.float64, return_stages=False):
    """
    weights: deepbinner_amd.model_format.ModelWeights;  x: [N, 1024] or [N, 1024, 1].
    Returns softmax probabilities [N, n_classes] (and, optionally, the activations after each
    stage A..H as named in DESIGN.md).  network_architecture.py line numbers in comments.
    """
    dtype = np.dtype(dtype).type
    x = np.asarray(x, dtype=dtype)
    if x.ndim == 2:
        x = x[:, :, None]
    shapes = conv_shapes(weights.n_classes)
    stages = {}

    def conv(i, t):
        kernel, bias = weights.convs[i - 1]
        _, _, _, _, stride, padding = shapes[i - 1]
        return relu(conv1d(t, kernel.astype(dtype), bias.astype(dtype), stride, padding))

    def bn(i, t):
        return batch_norm(t, weights.bns[i - 1], dtype)

    x = bn(1, conv(1, x))                                   # :28-31
    stages['A'] = x
    x = bn(2, max_pool2(conv(4, conv(3, conv(2, x)))))      # :34-40
    stages['B'] = x
    x = bn(3, max_pool2(conv(7, conv(6, conv(5, x)))))      # :43-51
    stages['C'] = x
    x = bn(4, max_pool2(conv(9, conv(8, x))))               # :54-59
    stages['D'] = x
    x1 = conv(10, avg_pool3_same(x))                        # :62-63
    x2 = conv(11, x)                                        # :64
    x3 = conv(13, conv(12, x))                              # :65-66
    x4 = conv(16, conv(15, conv(14, x)))                    # :67-69
    x = np.concatenate([x1, x2, x3, x4], axis=2)            # :70
    x = bn(5, max_pool2(x))                                 # :71-73
    stages['E'] = x
    x = bn(6, conv(17, x))                                  # :77-80
    stages['F'] = x
    x = bn(7, max_pool2(conv(19, conv(18, x))))             # :83-88
    stages['G'] = x
    x = conv(20, x)                                         # :91
    logits = x.mean(axis=1)                                 # :92
    stages['logits'] = logits
    probs = softmax(logits)                                 # :93
    stages['H'] = probs
    if return_stages:
        return probs, stages
    return probs

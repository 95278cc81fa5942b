"""
ORACLE — test infrastructure, not product code (see ``network_ref.py`` for the import rule).

Plain-Python/NumPy restatement of the reference's per-batch logic around ``model.predict``:

* ``normalise``                         ``deepbinner/trim_signal.py:61-69``
* window slicing / padding              ``deepbinner/classify.py:337-358``
* cross-step merge                      ``deepbinner/classify.py:363-377``
* ``make_sum_to_one``                   ``deepbinner/classify.py:387-393``
* ``get_barcode_call_from_probabilities``  ``deepbinner/classify.py:285-295``
* ``combine_calls``                     ``deepbinner/classify.py:298-322``
* ``find_signal_start_pos``             ``deepbinner/trim_signal.py:20-58``

Pinned by running the reference's own ``call_batch`` / ``normalise`` / ``find_signal_start_pos``
(imported from /root/reference with empty stand-in modules for h5py/keras/tensorflow, which those
functions never touch) on the same inputs: see ``make_golden.py`` and the committed fixtures under
``tests/golden/``.  Written as deliberately simple loops; use small cases.
"""

import numpy as np


class CannotTrim(IndexError):
    pass


def normalise(signal):
    signal = np.asarray(signal)
    if len(signal) == 0:
        return signal
    mean = np.mean(signal)
    stdev = np.std(signal)
    if stdev > 0.0:
        return (signal - mean) / stdev
    return signal - mean


def window_bounds(length, step, input_size, side):
    """[a, b) of the signal slice for scan step ``step`` (classify.py:337-349)."""
    half = input_size // 2
    sig_start = step * half
    sig_end = sig_start + input_size
    if side == 'start':
        return min(sig_start, length), min(sig_end, length)
    assert side == 'end'
    return max(length - sig_end, 0), max(length - sig_start, 0)


def make_windows(signals, input_size, scan_size, side):
    """-> float64 [steps, n_reads, input_size]: normalised, zero-padded windows."""
    steps = int(scan_size / (input_size // 2))
    out = np.zeros((steps, len(signals), input_size), dtype=np.float64)
    for s in range(steps):
        for i, signal in enumerate(signals):
            a, b = window_bounds(len(signal), s, input_size, side)
            w = normalise(np.asarray(signal)[a:b])
            n = len(w)
            if n == 0:
                continue
            if side == 'start':
                out[s, i, :n] = w          # right-padded (classify.py:355)
            else:
                out[s, i, input_size - n:] = w   # left-padded (classify.py:357)
    return out


def merge_steps(per_step_probs):
    """per_step_probs [steps, n_reads, C] -> merged [n_reads, C] (classify.py:363-374)."""
    p = np.array(per_step_probs[0], copy=True)
    for s in range(1, len(per_step_probs)):
        lab = per_step_probs[s]
        p[:, 0] = np.minimum(p[:, 0], lab[:, 0])
        p[:, 1:] = np.maximum(p[:, 1:], lab[:, 1:])
    return p


def make_sum_to_one(probabilities):
    """classify.py:387-393, evaluated in float64 as NumPy-1.x scalar promotion did."""
    p = np.asarray(probabilities, dtype=np.float64)
    none_p = p[0]
    factor = (1.0 - none_p) / p[1:].sum()
    out = p * factor
    out[0] = none_p
    return out


def barcode_call(probabilities, score_diff):
    order = sorted(range(len(probabilities)), key=lambda j: probabilities[j], reverse=True)
    best, second = order[0], order[1]
    if best == 0:
        return 'none'
    if probabilities[best] - probabilities[second] >= score_diff:
        return str(best)
    return 'none'


def combine_calls(start_call, end_call, mode):
    same = start_call == end_call
    if mode == 'require_both':
        return start_call if same else 'none'
    if mode == 'require_start':
        if same:
            return start_call
        if start_call == 'none':
            return 'none'
        return start_call if end_call == 'none' else 'none'
    assert mode == 'require_either'
    if same:
        return start_call
    if start_call == 'none':
        return end_call
    return start_call if end_call == 'none' else 'none'


def call_batch(predict, signals, input_size, scan_size, score_diff, side):
    """predict: callable float[N, input_size] -> float32[N, C].  -> (calls, probs float64)."""
    windows = make_windows(signals, input_size, scan_size, side)
    per_step = [np.asarray(predict(windows[s]), dtype=np.float32) for s in range(len(windows))]
    merged = merge_steps(per_step)
    probs = np.stack([make_sum_to_one(row) for row in merged]) if len(merged) else merged
    calls = [barcode_call(row, score_diff) for row in probs]
    return calls, probs


def find_signal_start_pos(signal):
    signal = np.asarray(signal)
    pos = 10

    def window_std(k):
        a = pos + 25 * k
        if a + 25 > len(signal):
            raise CannotTrim
        return np.std(signal[a:a + 25])

    while True:
        if window_std(0) > 20:
            if sum(1 for k in range(5) if window_std(k) > 20) >= 4:
                return pos
        pos += 25

"""
Signal normalisation and open-pore trimming — host-side mirror of the reference's
``deepbinner/trim_signal.py``.

``normalise`` (reference ``trim_signal.py:61-69``) is on the classify path; on the GPU path the
same arithmetic runs inside ``dbh_normalise_kernel`` (deepbinner_amd/csrc/dbh_forward.hip), this
function serves callers that enter at seam b1 with their own windows.
``find_signal_start_pos`` (reference ``trim_signal.py:20-58``) is a cheap host function used by
streaming front ends.
"""

import numpy as np


class CannotTrim(IndexError):
    pass


def normalise(signal):
    """(x - mean) / std in float64 with the population std; std == 0 -> x - mean; empty input
    is returned unchanged (reference trim_signal.py:61-69)."""
    if len(signal) == 0:
        return signal
    signal = np.asarray(signal)
    centred = signal - np.mean(signal)
    stdev = np.std(signal)
    return centred / stdev if stdev > 0.0 else centred


_INITIAL_TRIM = 10
_INCREMENT = 25
_STDEV_THRESHOLD = 20
_LOOK_FORWARD = 5
_NEEDED = 4


def get_window_stdev(signal, current_pos, window_num, increment):
    start = current_pos + window_num * increment
    end = start + increment
    if end > len(signal):
        raise CannotTrim
    return np.std(signal[start:end])


def find_signal_start_pos(signal):
    """Approximate position where open-pore signal ends (reference trim_signal.py:20-48):
    skip 10 samples, then step in 25-sample windows until the next window is noisy (std > 20)
    and at least 4 of the next 5 are.  Raises CannotTrim if the signal runs out first."""
    pos = _INITIAL_TRIM
    while True:
        if get_window_stdev(signal, pos, 0, _INCREMENT) > _STDEV_THRESHOLD:
            noisy = sum(1 for i in range(_LOOK_FORWARD)
                        if get_window_stdev(signal, pos, i, _INCREMENT) > _STDEV_THRESHOLD)
            if noisy >= _NEEDED:
                return pos
        pos += _INCREMENT

"""
ORACLE (test infrastructure): ctypes access to the two CPU versions of the semi-global DTW -
``restatement`` (oracle/dtwref.c, always built) and ``reference`` (oracle/_ref/dtw.so = the
reference's own deepbinner/dtw/dtw.cpp compiled where it lies, present when the build container
made it).  Same return convention as the reference's binding, dtw_semi_global.py:44-59:
``(distance, ref_start, ref_end, [(ref_index, query_index), ...] from start to end)``.
"""
import ctypes
import os

import numpy as np
from numpy.ctypeslib import ndpointer

HERE = os.path.dirname(os.path.abspath(__file__))
PATHS = {'restatement': (os.path.join(HERE, '_build', 'libdtwref.so'), 'dtwref_semi_global'),
         'reference': (os.path.join(HERE, '_ref', 'dtw.so'), 'semi_global_dtw')}
_loaded = {}


def available(kind):
    return os.path.isfile(PATHS[kind][0])


def _function(kind):
    if kind not in _loaded:
        path, symbol = PATHS[kind]
        fn = getattr(ctypes.CDLL(path), symbol)
        fn.restype = ctypes.c_double
        fn.argtypes = [ndpointer(ctypes.c_double, flags='C_CONTIGUOUS'),
                       ndpointer(ctypes.c_double, flags='C_CONTIGUOUS'), ctypes.c_int, ctypes.c_int,
                       ndpointer(ctypes.c_int, flags='C_CONTIGUOUS'),
                       ndpointer(ctypes.c_int, flags='C_CONTIGUOUS'),
                       ndpointer(ctypes.c_int, flags='C_CONTIGUOUS')]
        _loaded[kind] = fn
    return _loaded[kind]


def semi_global_dtw(ref, query, kind='restatement'):
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    query = np.ascontiguousarray(query, dtype=np.float64)
    alignment = np.empty((len(ref) + len(query)) * 2, dtype=np.int32)
    positions = np.empty(2, dtype=np.int32)
    path_length = np.empty(1, dtype=np.int32)
    distance = _function(kind)(ref, query, len(ref), len(query), alignment, positions, path_length)
    pairs = alignment[:2 * int(path_length[0])].reshape(-1, 2)[::-1]
    return distance, int(positions[0]), int(positions[1]), [(int(a), int(b)) for a, b in pairs]

"""
Semi-global dynamic time warping on the GPU: the host side of ``libdeepbinner_dtw.so`` (C ABI
``include/deepbinner_dtw.h``, kernel ``csrc/dtw_kernel.hip``), with the call surface of the
reference's ``deepbinner/dtw_semi_global.py``:

  ``semi_global_dtw(ref, query)``                  (:44-59)  one alignment
  ``semi_global_dtw_with_rescaling(ref, query)``   (:62-95)  two passes with a linear re-fit of the
                                                             query in between

and, new, the batched forms that give the device something to do - ``semi_global_dtw_batch`` and
``semi_global_dtw_with_rescaling_batch`` take lists of signals and run every pair of a pass in one
launch.  Results per pair are exactly those of the one-pair functions.

There is no CPU fallback: without the library or without a gfx950 device the calls raise.
"""

import ctypes
import os

import numpy as np
from numpy.ctypeslib import ndpointer

EXPORTED_SYMBOLS = ('dtw_version', 'dtw_status_string', 'dtw_last_error', 'semi_global_dtw',
                    'dtw_semi_global_batch', 'dtw_last_kernel_time')
LIB_NAME = 'libdeepbinner_dtw.so'
_lib = None


def library_path():
    return os.environ.get('DEEPBINNER_DTW_LIB') or os.path.join(
        os.path.dirname(os.path.abspath(__file__)), LIB_NAME)


def available():
    return os.path.isfile(library_path())


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.isfile(path):
        raise RuntimeError('{} is not built (run `make -C deepbinner_amd/csrc`): the DTW has no '
                           'CPU fallback'.format(path))
    lib = ctypes.CDLL(path)
    f64 = ndpointer(ctypes.c_double, flags='C_CONTIGUOUS')
    i64 = ndpointer(ctypes.c_int64, flags='C_CONTIGUOUS')
    i32 = ndpointer(ctypes.c_int32, flags='C_CONTIGUOUS')
    for name in ('dtw_version', 'dtw_last_error'):
        getattr(lib, name).restype = ctypes.c_char_p
        getattr(lib, name).argtypes = []
    lib.dtw_status_string.restype = ctypes.c_char_p
    lib.dtw_status_string.argtypes = [ctypes.c_int]
    # exactly the reference's binding (dtw_semi_global.py:33-41)
    lib.semi_global_dtw.restype = ctypes.c_double
    lib.semi_global_dtw.argtypes = [f64, f64, ctypes.c_int, ctypes.c_int, i32, i32, i32]
    lib.dtw_semi_global_batch.restype = ctypes.c_int
    lib.dtw_semi_global_batch.argtypes = [f64, i64, f64, i64, ctypes.c_int64, f64, i32, i32,
                                          ctypes.c_void_p]
    lib.dtw_last_kernel_time.restype = ctypes.c_int
    lib.dtw_last_kernel_time.argtypes = [ctypes.POINTER(ctypes.c_double),
                                         ctypes.POINTER(ctypes.c_int64)]
    _lib = lib
    return lib


def _failed(lib, what, status=None):
    text = lib.dtw_last_error().decode()
    if status is not None:
        text = '{} ({})'.format(lib.dtw_status_string(status).decode(), text)
    raise RuntimeError('{} failed: {}'.format(what, text))


def semi_global_dtw(ref, query):
    """(distance, ref_start, ref_end, [(ref_index, query_index), ...]) - the pairs from the start
    of the alignment to its end, as the reference returns them (:52-59)."""
    lib = load_library()
    ref = np.ascontiguousarray(ref, dtype=np.float64)
    query = np.ascontiguousarray(query, dtype=np.float64)
    alignment = np.empty((len(ref) + len(query)) * 2, dtype=np.int32)
    positions = np.empty(2, dtype=np.int32)
    path_length = np.zeros(1, dtype=np.int32)
    distance = lib.semi_global_dtw(ref, query, len(ref), len(query), alignment, positions,
                                   path_length)
    if path_length[0] == 0:
        _failed(lib, 'semi_global_dtw')
    pairs = alignment[:2 * int(path_length[0])].reshape(-1, 2)[::-1]
    return distance, positions[0], positions[1], [(a, b) for a, b in pairs]


def _pack(signals):
    arrays = [np.ascontiguousarray(s, dtype=np.float64).ravel() for s in signals]
    offsets = np.zeros(len(arrays) + 1, dtype=np.int64)
    np.cumsum([len(a) for a in arrays], out=offsets[1:])
    flat = np.concatenate(arrays) if arrays else np.zeros(0, dtype=np.float64)
    return np.ascontiguousarray(flat), offsets


def semi_global_dtw_batch(refs, queries, alignments=True):
    """One launch for all pairs (refs[k], queries[k]).  Returns a list of
    ``(distance, ref_start, ref_end, pairs)`` with ``pairs`` an ``int32[n, 2]`` array of
    (ref_index, query_index) from start to end, or ``None`` when ``alignments`` is false."""
    if len(refs) != len(queries):
        raise ValueError('as many queries as references, please')
    lib = load_library()
    n = len(refs)
    if n == 0:
        return []
    flat_refs, ref_offsets = _pack(refs)
    flat_queries, query_offsets = _pack(queries)
    distances = np.empty(n, dtype=np.float64)
    positions = np.empty(2 * n, dtype=np.int32)
    lengths = np.empty(n, dtype=np.int32)
    alignment = np.empty(2 * int(ref_offsets[-1] + query_offsets[-1]), dtype=np.int32) \
        if alignments else None
    status = lib.dtw_semi_global_batch(
        flat_refs, ref_offsets, flat_queries, query_offsets, n, distances, positions, lengths,
        alignment.ctypes.data if alignments else None)
    if status != 0:
        _failed(lib, 'dtw_semi_global_batch', status)
    results = []
    for k in range(n):
        pairs = None
        if alignments:
            at = 2 * int(ref_offsets[k] + query_offsets[k])
            pairs = alignment[at:at + 2 * int(lengths[k])].reshape(-1, 2)[::-1]
        results.append((float(distances[k]), int(positions[2 * k]), int(positions[2 * k + 1]),
                        pairs))
    return results


def last_kernel_time():
    """(milliseconds, cells) of the kernels of the last call."""
    lib = load_library()
    ms, cells = ctypes.c_double(0.0), ctypes.c_int64(0)
    lib.dtw_last_kernel_time(ctypes.byref(ms), ctypes.byref(cells))
    return ms.value, cells.value


ITERATIONS = 2                       # reference :70
SLOPE_RANGE = (0.75, 1.333)          # reference :91-93


def _refit(ref, query, pairs):
    """Least-squares line through (query value, reference value) over the aligned pairs; the
    query moved onto it (reference :81-89)."""
    x = query[pairs[:, 1]]
    y = ref[pairs[:, 0]]
    design = np.vstack([x, np.ones(len(x))]).T
    slope, intercept = np.linalg.lstsq(design, y, rcond=None)[0]
    return slope * query + intercept, slope


def semi_global_dtw_with_rescaling_batch(refs, queries):
    """``semi_global_dtw_with_rescaling`` for many pairs: each of the two passes is one launch.
    Returns a list of ``(distance, ref_start, ref_end, [(ref value, query value), ...])``."""
    refs = [np.ascontiguousarray(r, dtype=np.float64) for r in refs]
    queries = [np.array(q, dtype=np.float64) for q in queries]
    slopes = [1.0] * len(refs)
    results = []
    for iteration in range(ITERATIONS):
        results = semi_global_dtw_batch(refs, queries)
        if iteration == ITERATIONS - 1:
            break
        for k, (_, _, _, pairs) in enumerate(results):
            queries[k], slope = _refit(refs[k], queries[k], pairs)
            slopes[k] *= slope
    out = []
    for k, (distance, start, end, pairs) in enumerate(results):
        if slopes[k] < SLOPE_RANGE[0] or slopes[k] > SLOPE_RANGE[1]:
            distance = float('inf')
        values = list(zip(refs[k][pairs[:, 0]], queries[k][pairs[:, 1]]))
        out.append((distance, start, end, values))
    return out


def semi_global_dtw_with_rescaling(ref, query):
    """Reference :62-95 (after https://arxiv.org/abs/1705.01620): align, fit the query to the
    reference over the aligned pairs, align again; a total slope outside 0.75 .. 1.333 means the
    fit went wrong and the distance becomes inf."""
    return semi_global_dtw_with_rescaling_batch([ref], [query])[0]

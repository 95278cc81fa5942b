"""
ORACLE — ctypes binding of ``oracle/_build/libdbref.so`` (built from ``dbref.c`` by
``oracle/Makefile``).  Test infrastructure and CPU baseline only; see ``network_ref.py`` for the
import rule.
"""

import ctypes
import os
import subprocess

import numpy as np
from numpy.ctypeslib import ndpointer

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, '_build', 'libdbref.so')
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE])
    return _PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.isfile(_PATH):
            build()
        lib = ctypes.CDLL(_PATH)
        f32 = ndpointer(np.float32, flags='C_CONTIGUOUS')
        lib.dbref_create.restype = ctypes.c_void_p
        lib.dbref_create.argtypes = [f32, ctypes.c_int64, ctypes.c_int]
        lib.dbref_destroy.argtypes = [ctypes.c_void_p]
        lib.dbref_predict.restype = ctypes.c_int
        lib.dbref_predict.argtypes = [ctypes.c_void_p, f32, ctypes.c_int64, f32, ctypes.c_int]
        lib.dbref_windows.argtypes = [ndpointer(np.int16, flags='C_CONTIGUOUS'),
                                      ndpointer(np.int64, flags='C_CONTIGUOUS'), ctypes.c_int64,
                                      ctypes.c_int, ctypes.c_int, f32]
        lib.dbref_merge.argtypes = [f32, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_double, f32, ndpointer(np.int32, flags='C_CONTIGUOUS')]
        _lib = lib
    return _lib


class CModel:
    def __init__(self, weights):
        self.lib = load()
        flat = weights.flat()
        self.n_classes = weights.n_classes
        self.handle = self.lib.dbref_create(flat, flat.size, weights.n_classes)
        if not self.handle:
            raise ValueError('weight blob rejected by dbref_create')
        self.threads_used = 1

    def predict(self, x, threads=0):
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1, 1024))
        out = np.empty((x.shape[0], self.n_classes), dtype=np.float32)
        self.threads_used = self.lib.dbref_predict(self.handle, x, x.shape[0], out, threads)
        return out

    def windows(self, samples, offsets, side, scan_size):
        n = len(offsets) - 1
        steps = scan_size // 512
        out = np.empty((n * steps, 1024), dtype=np.float32)
        samples = np.ascontiguousarray(samples, dtype=np.int16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if n and (offsets[0] < 0 or offsets[-1] > samples.size or (np.diff(offsets) < 0).any()):
            raise ValueError('offsets do not describe reads inside the sample buffer')
        if samples.size == 0:
            samples = np.zeros(1, dtype=np.int16)
        self.lib.dbref_windows(samples, np.ascontiguousarray(offsets, dtype=np.int64), n,
                               0 if side == 'start' else 1, scan_size, out)
        return out

    def merge(self, wprobs, n_reads, steps, score_diff):
        wprobs = np.ascontiguousarray(wprobs, dtype=np.float32)
        probs = np.empty((n_reads, self.n_classes), dtype=np.float32)
        calls = np.empty(n_reads, dtype=np.int32)
        self.lib.dbref_merge(wprobs, n_reads, steps, self.n_classes, score_diff, probs, calls)
        return probs, calls

    def classify(self, samples, offsets, side, scan_size, score_diff, threads=0):
        w = self.windows(samples, offsets, side, scan_size)
        p = self.predict(w, threads)
        return self.merge(p, len(offsets) - 1, scan_size // 512, score_diff)

    def __del__(self):
        try:
            if self.handle:
                self.lib.dbref_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

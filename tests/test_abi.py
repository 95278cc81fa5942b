"""The C ABI: the library loads and exports every symbol include/deepbinner_hip.h declares.
No compute calls here (CPU-only box)."""
import ctypes
import os
import re

import numpy as np

from conftest import REPO
from deepbinner_amd import hip_backend


def declared_symbols():
    text = open(os.path.join(REPO, 'include', 'deepbinner_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dbh_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = hip_backend.load_library()
    names = declared_symbols()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(hip_backend.EXPORTED_SYMBOLS) == names


def test_status_strings_and_version():
    lib = hip_backend.load_library()
    assert b'gfx950' in lib.dbh_version()
    assert lib.dbh_status_string(0) == b'ok'
    for code in range(1, 7):
        assert lib.dbh_status_string(code) not in (b'ok', b'unknown status')
    assert lib.dbh_status_string(99) == b'unknown status'


def test_argument_validation_without_device():
    lib = hip_backend.load_library()
    handle = ctypes.c_void_p()
    blob = np.zeros(10, dtype=np.float32)
    # wrong geometry is rejected before any device work
    assert lib.dbh_model_create(blob, 10, 13, 512, ctypes.byref(handle)) == 5    # UNSUPPORTED
    assert lib.dbh_model_create(blob, 10, 40, 1024, ctypes.byref(handle)) == 5
    assert lib.dbh_model_create(blob, 10, 13, 1024, ctypes.byref(handle)) == 4   # BAD_WEIGHTS
    per = ctypes.c_int64()
    assert lib.dbh_stage_floats(0, ctypes.byref(per)) == 0 and per.value == 512 * 48
    assert lib.dbh_stage_floats(9, ctypes.byref(per)) == 1
    assert lib.dbh_model_destroy(None) == 0


def test_header_cites_reference_lines():
    text = open(os.path.join(REPO, 'include', 'deepbinner_hip.h')).read()
    for cite in ('classify.py:361', 'classify.py:325-384', 'classify.py:90',
                 'dtw_semi_global.py:30-41', 'trim_signal.py:61-69'):
        assert cite in text

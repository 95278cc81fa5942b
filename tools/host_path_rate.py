#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (dbh_classify_i16 / dbh_classify_pair_i16:
[staging copy ->] H2D -> kernels -> D2H, groups of 32k windows through three slots), from
pageable and from pinned caller buffers.  Not bench.py's `value` (that one keeps inputs resident
in HBM); reported in DESIGN.md next to it.
Usage: python tools/host_path_rate.py [n_reads]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepbinner_amd import hip_backend                      # noqa: E402
from deepbinner_amd.model_format import ModelWeights        # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    w, _ = ModelWeights.load(os.path.join(REPO, 'deepbinner_amd', 'models',
                                          'EXP-NBD103_read_starts.dbw'))
    model = hip_backend.HipModel(w, device=0)
    lib = hip_backend.load_library()
    rng = np.random.default_rng(1)
    out = {}
    end_w, _ = ModelWeights.load(os.path.join(REPO, 'deepbinner_amd', 'models',
                                              'EXP-NBD103_read_ends.dbw'))
    end_model = hip_backend.HipModel(end_w, device=0)
    if len(sys.argv) > 2:              # windows per model and group of the pipeline (A/B knob)
        model.set_host_group(int(sys.argv[2]))
        end_model.set_host_group(int(sys.argv[2]))
        out['host_group_windows'] = int(sys.argv[2])
    for label, length, scan in (('1024-sample reads, scan 512 (1 window/read)', 1024, 512),
                                ('6656-sample reads, scan 6144 (12 windows/read)', 6656, 6144),
                                ('13312-sample reads, scan 6144, start + end models + '
                                 'combine_calls in one call (24 windows/read)', 13312, 6144)):
        pair = 'start + end' in label
        reads = n if scan == 512 else (n // 16 if pair else n // 8)
        samples = rng.integers(300, 700, size=reads * length, dtype=np.int16)
        offsets = np.arange(reads + 1, dtype=np.int64) * length
        ptr = ctypes.c_void_p()
        hip_backend.check(lib.dbh_malloc_host(ctypes.byref(ptr), samples.nbytes))
        pinned = np.ctypeslib.as_array(ctypes.cast(ptr.value, ctypes.POINTER(ctypes.c_int16)),
                                       shape=samples.shape)
        pinned[:] = samples
        probs = np.empty((reads, 13), dtype=np.float32)
        calls = np.empty(reads, dtype=np.int32)
        row = {'reads': reads}
        for kind, buf in (('pageable', samples), ('pinned', pinned)):
            best = None
            for _ in range(4):
                t0 = time.perf_counter()
                if pair:
                    hip_backend.classify_pair(model, end_model, buf, offsets, scan, 0.5)
                else:
                    hip_backend.check(lib.dbh_classify_i16(model.handle, buf, offsets, reads, 0,
                                                           scan, 0.5, probs, calls))
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            windows = reads * (scan // 512) * (2 if pair else 1)
            row[kind] = {'seconds': round(best, 4), 'reads_per_s': round(reads / best),
                         'windows_per_s': round(windows / best),
                         'input_GB_per_s': round(samples.nbytes / best / 1e9, 2)}
        out[label] = row
        hip_backend.check(lib.dbh_free_host(ptr))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (dbh_classify_i16: pack -> H2D -> kernels ->
D2H, double-buffered on two streams).  Not bench.py's `value` (that one keeps inputs resident in
HBM); reported in DESIGN.md next to it.
Usage: python tools/host_path_rate.py [n_reads]"""
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
    for label, length, scan in (('1024-sample reads, scan 512 (1 window/read)', 1024, 512),
                                ('6656-sample reads, scan 6144 (12 windows/read)', 6656, 6144)):
        reads = n if scan == 512 else n // 8
        samples = rng.integers(300, 700, size=reads * length, dtype=np.int16)
        offsets = np.arange(reads + 1, dtype=np.int64) * length
        probs = np.empty((reads, 13), dtype=np.float32)
        calls = np.empty(reads, dtype=np.int32)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            hip_backend.check(lib.dbh_classify_i16(model.handle, samples, offsets, reads, 0, scan,
                                                   0.5, probs, calls))
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        windows = reads * (scan // 512)
        out[label] = {'reads': reads, 'seconds': round(best, 4),
                      'reads_per_s': round(reads / best), 'windows_per_s': round(windows / best),
                      'input_GB_per_s': round(samples.nbytes / best / 1e9, 2)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Rate of the semi-global DTW kernel on one GPU.

  python tools/dtw_rate.py [--pairs 4096] [--ref 4000] [--query 500]

Prints one JSON object: cell updates per second by HIP events around the kernel (GCUPS) and the
same including H2D of the signals and D2H of the alignments.  (The reference's own CPU code is
timed beside the kernel, and its answers compared, by the opt-in test
tests/test_dtw.py::test_gpu_rate_and_soak_beside_the_reference - the oracle is test infrastructure
and is not imported by tools.)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepbinner_amd import dtw_semi_global as dtw      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=4096)
    ap.add_argument('--ref', type=int, default=4000)
    ap.add_argument('--query', type=int, default=500)
    ap.add_argument('--cpu-pairs', type=int, default=0, help='(ignored; kept for old command lines)')
    ap.add_argument('--repeats', type=int, default=3)
    opts = ap.parse_args()
    rng = np.random.default_rng(3)
    refs = [rng.normal(size=opts.ref) for _ in range(opts.pairs)]
    queries = [rng.normal(size=opts.query) for _ in range(opts.pairs)]
    dtw.semi_global_dtw_batch(refs[:64], queries[:64])            # warm-up: buffers, code load
    best_wall, best_ms = None, None
    for _ in range(opts.repeats):
        t0 = time.perf_counter()
        dtw.semi_global_dtw_batch(refs, queries)
        wall = time.perf_counter() - t0
        ms, cells = dtw.last_kernel_time()
        if best_wall is None or wall < best_wall:
            best_wall, best_ms = wall, ms
    out = {'pairs': opts.pairs, 'ref_len': opts.ref, 'query_len': opts.query, 'cells': cells,
           'kernel_ms': best_ms, 'kernel_GCUPS': cells / (best_ms * 1e-3) / 1e9,
           'call_seconds_incl_transfers_and_python': best_wall,
           'call_GCUPS': cells / best_wall / 1e9,
           'direction_bytes_per_cell': 4.0 / (4 if opts.query <= 256 else 8 if opts.query <= 512
                                              else 16)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""
ORACLE tooling - golden vectors for the semi-global DTW, ``tests/golden/dtw.npz``.  Runs ONLY in
the build container: the answers come from ``oracle/_ref/dtw.so``, i.e. the reference's own
``deepbinner/dtw/dtw.cpp`` compiled where it lies (``make -C oracle``).  Committed: seeded inputs
and the reference's outputs for them (distance, positions, alignment) - data only.

Cases: random signals of assorted shapes (one-sample reference or query, queries up to 2,300
samples so that every lane width and the multi-panel path of the kernel are covered), a query cut out
of its reference (distance 0), and a squiggle-like pair (piecewise-constant levels + noise) with a
rescaled query as in ``semi_global_dtw_with_rescaling``.
"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import dtw_ref      # noqa: E402

SHAPES = [(1, 1), (1, 7), (9, 1), (2, 2), (50, 20), (300, 64), (257, 65), (700, 256), (640, 257),
          (1200, 512), (900, 513), (1500, 1024), (1100, 1025), (2500, 2048), (2600, 2300)]


def squiggle(rng, n_levels, dwell):
    levels = rng.normal(0.0, 1.0, size=n_levels)
    return np.repeat(levels, rng.integers(max(1, dwell - 3), dwell + 4, size=n_levels))


def main():
    rng = np.random.default_rng(20260928)
    cases = []
    for r, q in SHAPES:
        cases.append((rng.normal(size=r), rng.normal(size=q)))
    ref = rng.normal(size=800)
    cases.append((ref, ref[300:420].copy()))                      # an exact occurrence
    query = squiggle(rng, 40, 8)
    ref = np.concatenate([squiggle(rng, 60, 8), query + rng.normal(0, 0.05, len(query)),
                          squiggle(rng, 30, 8)])
    cases.append((ref, 0.9 * query + 0.2))
    out = {'n': np.int64(len(cases))}
    for k, (ref, query) in enumerate(cases):
        distance, start, end, pairs = dtw_ref.semi_global_dtw(ref, query, 'reference')
        again = dtw_ref.semi_global_dtw(ref, query, 'restatement')
        assert again == (distance, start, end, pairs), k
        out['ref_%d' % k] = ref
        out['query_%d' % k] = query
        out['answer_%d' % k] = np.array([distance, start, end], dtype=np.float64)
        out['pairs_%d' % k] = np.array(pairs, dtype=np.int32).reshape(-1, 2)
        print(k, len(ref), len(query), distance, start, end, len(pairs))
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'dtw.npz'), **out)


if __name__ == '__main__':
    main()

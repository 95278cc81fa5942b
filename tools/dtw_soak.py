#!/usr/bin/env python3
"""Soak of the DTW kernel against the reference's own code (oracle/_ref/dtw.so when it travelled
with the repository, else the oracle's restatement): N random pairs of assorted shapes and signal
kinds in batches, every distance, position and path compared bit for bit.
  python tools/dtw_soak.py [--pairs 4000] [--seed 1]"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepbinner_amd import dtw_semi_global as dtw      # noqa: E402
from oracle import dtw_ref                              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=4000)
    ap.add_argument('--seed', type=int, default=1)
    opts = ap.parse_args()
    rng = np.random.default_rng(opts.seed)
    kind = 'reference' if dtw_ref.available('reference') else 'restatement'
    refs, queries = [], []
    for k in range(opts.pairs):
        r = int(rng.integers(1, 4000)) if k % 11 else int(rng.integers(1, 70))
        q = int(rng.integers(1, 1500)) if k % 13 else int(rng.integers(1000, 3500))
        style = k % 4
        if style == 0:
            refs.append(rng.normal(size=r)); queries.append(rng.normal(size=q))
        elif style == 1:      # squiggle-like: levels held for a few samples, small noise
            refs.append(np.repeat(rng.normal(size=r // 6 + 1), 6)[:r] + rng.normal(0, .05, r))
            queries.append(np.repeat(rng.normal(size=q // 6 + 1), 6)[:q] + rng.normal(0, .05, q))
        elif style == 2:      # the query is a noisy, rescaled piece of the reference
            ref = rng.normal(size=max(r, 2))
            a = int(rng.integers(0, len(ref) - 1)); b = int(rng.integers(a + 1, len(ref)))
            refs.append(ref); queries.append(1.1 * ref[a:b][:q] + 0.1 + rng.normal(0, .1, len(ref[a:b][:q])))
        else:                 # large offsets and scales
            refs.append(rng.normal(400, 90, size=r)); queries.append(rng.normal(450, 60, size=q))
    t0 = time.perf_counter()
    got = dtw.semi_global_dtw_batch(refs, queries)
    gpu_seconds = time.perf_counter() - t0
    t0 = time.perf_counter()
    cells = 0
    for k, (ref, query) in enumerate(zip(refs, queries)):
        want = dtw_ref.semi_global_dtw(ref, query, kind)
        cells += len(ref) * len(query)
        assert got[k][0] == want[0] and got[k][1:3] == want[1:3], (k, len(ref), len(query))
        assert np.array_equal(got[k][3], np.array(want[3], dtype=np.int32).reshape(-1, 2)), k
    print(json.dumps({'pairs': opts.pairs, 'seed': opts.seed, 'cells': cells, 'checked_against': kind,
                      'identical': True, 'gpu_call_seconds': round(gpu_seconds, 3),
                      'cpu_seconds': round(time.perf_counter() - t0, 1)}))


if __name__ == '__main__':
    main()

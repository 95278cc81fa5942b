#!/usr/bin/env python3
"""A soak of the streaming path: the same containers classified over and over with the GPU
inflating a random share of their chunks, on a random number of queues, with and without CUs
set aside - every pass must give the calls of the CPU loader's path, read for read.

    python tools/stream_soak.py DIR [minutes=2]      (DIR: containers of gpu_inflate_split.py --write)
"""
import glob
import io
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    directory = sys.argv[1]
    minutes = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
    from deepbinner_amd import classify, fast5_native, hip_backend, realtime
    paths = sorted(glob.glob(os.path.join(directory, '*.fast5')))
    models = os.path.join(REPO, 'deepbinner_amd', 'models')
    sm, _, em, _, _, _ = classify.load_and_check_models(
        os.path.join(models, 'EXP-NBD103_read_starts.dbw'),
        os.path.join(models, 'EXP-NBD103_read_ends.dbw'), 6144, out_dest=io.StringIO())
    base = classify.device_replicas(sm, em)
    want = []
    for _, ids, samples, offsets, status in fast5_native.stream_reads(paths, keep=6656, threads=16):
        assert (status == 0).all()
        want.append(hip_backend.classify_pair(sm, em, samples, offsets, 6144, 0.5, 'require_either'))
    print('baseline: %d containers, %d reads, %d with a barcode' % (
        len(want), sum(len(w) for w in want), sum(int((w != 0).sum()) for w in want)), flush=True)

    def work(item, start_replica, end_replica):
        _, ids, offsets, _, comp, records = item
        calls, status = hip_backend.classify_pair_deflated(
            start_replica, end_replica, comp, records, offsets, 6144, 0.5)
        assert (status == 0).all()
        return calls

    rng = np.random.default_rng(int(time.time()))
    clones, passes, t_end = {}, 0, time.time() + 60 * minutes
    while time.time() < t_end:
        share = int(rng.choice([0, 5, 20, 40, 58, 75, 99]))
        os.environ['DEEPBINNER_INFLATE_QUEUES'] = str(int(rng.integers(1, 7)))
        os.environ['DEEPBINNER_INFLATE_CUS'] = str(int(rng.choice([0, 0, 16, 32, 200])))
        queues, held = realtime.inflate_queues(base, share)
        stream = fast5_native.stream_raw(paths, threads=int(rng.integers(2, 17)),
                                         depth=len(queues) + 2, host_inflate_above=-share)
        t0 = time.perf_counter()
        got = list(classify.dispatch_batches(stream, queues, work))
        dt = time.perf_counter() - t0
        for model in held:
            model.reserve_cus(0)
        assert len(got) == len(want)
        for k, (g, w) in enumerate(zip(got, want)):
            assert np.array_equal(g, w), (passes, k, share, len(queues))
        passes += 1
        print(json.dumps({'pass': passes, 'host_share': share, 'queues': len(queues),
                          'cus_left': os.environ['DEEPBINNER_INFLATE_CUS'],
                          'reads_per_s': round(sum(len(w) for w in want) / dt)}), flush=True)
    print('soak: %d passes, every call as the CPU loader\'s path gives it' % passes)


if __name__ == '__main__':
    main()

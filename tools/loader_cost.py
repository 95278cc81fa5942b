#!/usr/bin/env python3
"""What a read costs the host in the raw streaming loader (f5_stream_open_raw: parse the container,
resolve every read's chunk addresses, fetch the chunks as stored): process CPU time per read over
containers of 4,000 deflated reads written on the spot (bench.py's configs[4] containers), for a
few team sizes, with the byte buffers in malloc'd and - where a GPU is there - in pinned memory.
DEEPBINNER_FAST5_LIB=<another build> measures that build instead (an A/B of the loader alone);
DEEPBINNER_FAST5_TIMING=1 makes the library print its own parse / resolve / fetch split;
LOADER_COST_SAMPLES=lo,hi sets the reads' lengths (default 2000,9000: bench.py's containers),
LOADER_COST_DEPTH the containers in flight (default 4).
Usage: python tools/loader_cost.py [containers] [directory]  -> gpurun_out/loader_cost.json"""
import json
import os
import shutil
import sys
import tempfile
import time
import uuid
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepbinner_amd import fast5_native, hdf5_write          # noqa: E402


def write_containers(directory, count, reads_per_container=4000):
    lo, hi = (int(v) for v in os.environ.get('LOADER_COST_SAMPLES', '2000,9000').split(','))
    rng = np.random.default_rng(20260929)
    pool = []
    for _ in range(1000):
        n = int(rng.integers(lo, hi))
        levels = np.repeat(rng.normal(450, 80, n // 8 + 1), 8)[:n]
        pool.append(np.clip(np.rint(levels + rng.normal(0, 8, n)), 0, 2047).astype(np.int16))
    with ThreadPoolExecutor(16) as workers:
        deflated = list(workers.map(lambda sig: zlib.compress(sig.tobytes(), 1), pool))

    def write(job):
        path, seed = job
        r = np.random.default_rng(seed)
        reads = []
        for _ in range(reads_per_container):
            j = int(r.integers(0, len(pool)))
            reads.append((str(uuid.UUID(bytes=r.bytes(16), version=4)), pool[j], None, deflated[j]))
        with open(path, 'wb') as f:
            f.write(hdf5_write.multi_read_fast5_bytes(reads))

    paths = [os.path.join(directory, 'c%02d.fast5' % c) for c in range(count)]
    with ThreadPoolExecutor(8) as workers:
        list(workers.map(write, [(p, 7000 + i) for i, p in enumerate(paths)]))
    return paths


def measure(paths, threads, repeats=5):
    best = None
    for _ in range(repeats):
        t0, c0, reads, comp_bytes = time.perf_counter(), time.process_time(), 0, 0
        for item in fast5_native.stream_raw(paths, threads=threads, depth=int(os.environ.get('LOADER_COST_DEPTH', '4'))):
            reads += len(item[1])
            comp_bytes += len(item[4])
        cpu, wall = time.process_time() - c0, time.perf_counter() - t0
        if best is None or cpu < best[0]:
            best = (cpu, wall)
    return {'threads': threads, 'cpu_us_per_read': round(best[0] / reads * 1e6, 2),
            'wall_s': round(best[1], 4), 'reads_per_s': round(reads / best[1]),
            'stored_bytes_per_read': round(comp_bytes / reads)}


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    directory = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix='loader_cost_')
    os.makedirs(directory, exist_ok=True)
    paths = write_containers(directory, count)
    out = {'library': fast5_native.library_path(), 'containers': count, 'reads': 4000 * count,
           'cpus': len(os.sched_getaffinity(0)), 'malloc': [], 'pinned': None}
    for threads in (1, 4, 16):
        out['malloc'].append(measure(paths, threads))
        print('malloc', out['malloc'][-1], flush=True)
    try:
        from deepbinner_amd import hip_backend
        hip_backend.use_pinned_loader_buffers()
        out['pinned'] = []
        for threads in (1, 4, 16):
            out['pinned'].append(measure(paths, threads))
            print('pinned', out['pinned'][-1], flush=True)
    except Exception as e:                                   # no GPU here: the malloc rows stand
        out['pinned_unavailable'] = str(e)[:200]
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    name = 'loader_cost%s.json' % ('_' + os.environ['LOADER_COST_TAG'] if os.environ.get('LOADER_COST_TAG') else '')
    json.dump(out, open(os.path.join(REPO, 'gpurun_out', name), 'w'), indent=1)
    if len(sys.argv) <= 2:
        shutil.rmtree(directory, ignore_errors=True)


if __name__ == '__main__':
    main()

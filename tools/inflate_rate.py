#!/usr/bin/env python3
"""Rate of the GPU inflate (dbh_inflate: kernel 1 Huffman -> tokens, kernel 2 tokens -> bytes) on
containers' worth of zlib streams, beside zlib and libdeflate on one host core.
The streams: squiggles like tools/multi_read_rate.py's (log-normal read lengths, mean ~27 k
samples), deflated at level 1 (MinKNOW, h5py gzip=1).
Usage: python tools/inflate_rate.py [n_streams] [mean_samples]    (rocprofv3 --kernel-trace
--stats around it splits the time between the two kernels)"""
import json
import os
import sys
import time
import zlib

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepbinner_amd import hip_backend                      # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    mean = int(sys.argv[2]) if len(sys.argv) > 2 else 27000
    shape = sys.argv[3] if len(sys.argv) > 3 else 'lognormal'      # or: uniform, sorted
    rng = np.random.default_rng(5)
    pool = []
    for _ in range(300):
        k = mean if shape == 'uniform' else \
            int(np.clip(rng.lognormal(np.log(mean) - 0.32, 0.8), 2000, 400000))
        levels = np.repeat(rng.normal(450, 80, k // 8 + 1), 8)[:k]
        pool.append(np.clip(np.rint(levels + rng.normal(0, 8, k)), 0, 2047).astype('<i2').tobytes())
    deflated = [zlib.compress(p, 1) for p in pool]
    picks = rng.integers(0, len(pool), n)
    if shape == 'sorted':          # longest first: the lanes of a wave get streams of one length
        picks = np.array(sorted(picks, key=lambda j: -len(deflated[j])))
    records = np.zeros(n, dtype=hip_backend.INFLATE_STREAM)
    comp, at, out_at = bytearray(), 0, 0
    for k, j in enumerate(picks):
        records[k] = (at, len(deflated[j]), out_at, len(pool[j]), hip_backend.INFLATE_ZLIB, 0)
        comp += deflated[j]
        at += len(deflated[j])
        out_at += len(pool[j])
    comp = np.frombuffer(bytes(comp), dtype=np.uint8)
    best = None
    for _ in range(4):
        out, status, ms = hip_backend.inflate(comp, records, out_at)
        best = ms if best is None else min(best, ms)
    if os.environ.get('DEEPBINNER_INFLATE_NOCHECK') == '1':      # timing-only builds
        print(json.dumps({'gpu_kernels_ms': round(best, 2), 'failed_streams': int((status != 0).sum())}))
        return
    assert (status == 0).all()
    k = int(picks[7])
    assert out[records[7]['out_offset']:records[7]['out_offset'] + len(pool[k])].tobytes() == pool[k]
    t0 = time.perf_counter()
    for j in picks[:400]:
        zlib.decompress(deflated[j])
    cpu = (time.perf_counter() - t0) / 400
    print(json.dumps({
        'streams': n, 'lengths': shape, 'compressed_MB': round(len(comp) / 1e6, 1), 'output_MB': round(out_at / 1e6, 1),
        'mean_output_KB_per_stream': round(out_at / n / 1e3, 1),
        'gpu_kernels_ms': round(best, 2), 'gpu_streams_per_s': round(n / (best * 1e-3)),
        'gpu_output_GB_per_s': round(out_at / best / 1e6, 2),
        'zlib_one_core_us_per_stream': round(cpu * 1e6, 1),
        'zlib_one_core_streams_per_s': round(1 / cpu)}))


if __name__ == '__main__':
    main()

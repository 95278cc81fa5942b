#!/usr/bin/env python3
"""Rates of the other BASELINE.json configurations on one MI355X (they are parity-test cases, not
bench lines; bench.py measures configs[1]).  Inputs resident in HBM, one fused launch per batch:
  configs[2]: EXP-NBD103 start + end models, 100,000 synthetic signals, batch 512, combine_calls
              (require_either) on the host and on the device;
  configs[3]: SQK-RBK004_read_starts, this GPU's 125,000-read shard of 1,000,000, batch 256;
  default CLI geometry: 6,656-sample reads, scan_size 6144 (12 windows per read), batch 256.
Parity of these configurations is the business of tests/test_gpu_parity.py; this tool only times them
(the oracle is test infrastructure and is not imported here).
Usage: python tools/config_rates.py"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import synthetic_reads                            # noqa: E402
from deepbinner_amd import classify, hip_backend             # noqa: E402
from deepbinner_amd.model_format import ModelWeights         # noqa: E402


def load(name):
    w, _ = ModelWeights.load(os.path.join(REPO, 'deepbinner_amd', 'models', name + '.dbw'))
    return w, hip_backend.HipModel(w, device=0)


def run(model, d_samples, d_offsets, n, batch, side, scan, reps=3):
    d_probs = hip_backend.DeviceBuffer(n * model.n_classes * 4)
    d_calls = hip_backend.DeviceBuffer(n * 4)
    best = None
    for _ in range(reps + 1):
        hip_backend.synchronize()
        t0 = time.perf_counter()
        model.classify_batched_dev(d_samples.ptr, d_offsets.ptr, n, batch, side, scan, 0.5,
                                   d_probs.ptr, d_calls.ptr, None)
        hip_backend.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best, d_calls.download((n,), np.int32), d_probs.download((n, model.n_classes),
                                                                    np.float32)


def main():
    argparse.ArgumentParser(description=__doc__).parse_args()
    out = {}

    # ---- configs[2]: dual-model combine path, batch 512 ------------------------------------
    n = 100000
    reads = synthetic_reads(n, 20260927)
    d_samples = hip_backend.DeviceBuffer.from_array(reads)
    d_offsets = hip_backend.DeviceBuffer.from_array(np.arange(n + 1, dtype=np.int64) * 1024)
    ws, start = load('EXP-NBD103_read_starts')
    we, end = load('EXP-NBD103_read_ends')
    t_s, calls_s, probs_s = run(start, d_samples, d_offsets, n, 512, 'start', 512)
    t_e, calls_e, probs_e = run(end, d_samples, d_offsets, n, 512, 'end', 512)
    args = argparse.Namespace(require_both=False, require_start=False, require_either=True)
    t0 = time.perf_counter()
    names_s = ['none' if c == 0 else str(int(c)) for c in calls_s]
    names_e = ['none' if c == 0 else str(int(c)) for c in calls_e]
    final = [classify.combine_calls(a, b, args) for a, b in zip(names_s, names_e)]
    t_c = time.perf_counter() - t0
    # the same on the device: both models and combine_calls queued on one stream, one sync
    d_cs, d_ce = hip_backend.DeviceBuffer(n * 4), hip_backend.DeviceBuffer(n * 4)
    d_ps = hip_backend.DeviceBuffer(n * start.n_classes * 4)
    d_pe = hip_backend.DeviceBuffer(n * end.n_classes * 4)
    d_final = hip_backend.DeviceBuffer(n * 4)
    t_all = None
    for _ in range(4):
        hip_backend.synchronize()
        t0 = time.perf_counter()
        start.classify_batched_dev(d_samples.ptr, d_offsets.ptr, n, 512, 'start', 512, 0.5,
                                   d_ps.ptr, d_cs.ptr, None)
        end.classify_batched_dev(d_samples.ptr, d_offsets.ptr, n, 512, 'end', 512, 0.5,
                                 d_pe.ptr, d_ce.ptr, None)
        hip_backend.combine_calls_dev(d_cs.ptr, d_ce.ptr, n, 'require_either', d_final.ptr)
        hip_backend.synchronize()
        dt = time.perf_counter() - t0
        t_all = dt if t_all is None else min(t_all, dt)
    on_device = d_final.download((n,), np.int32)
    assert [('none' if c == 0 else str(int(c))) for c in on_device] == final
    out['configs[2] EXP-NBD103 start+end, 100k signals, batch 512'] = {
        'gpu_seconds_start_model': round(t_s, 5), 'gpu_seconds_end_model': round(t_e, 5),
        'host_combine_calls_seconds': round(t_c, 4),
        'reads_per_s_gpu_both_models': round(n / (t_s + t_e)),
        'windows_per_s_gpu': round(2 * n / (t_s + t_e)),
        'reads_per_s_incl_host_combine': round(n / (t_s + t_e + t_c)),
        'gpu_seconds_both_models_and_combine_on_device': round(t_all, 5),
        'reads_per_s_all_on_device': round(n / t_all),
        'called': int(sum(c != 'none' for c in final))}

    # ---- configs[3]: one GPU's shard of 1M reads, SQK-RBK004 --------------------------------
    n = 125000
    reads = synthetic_reads(n, 20260928)
    d_samples = hip_backend.DeviceBuffer.from_array(reads)
    d_offsets = hip_backend.DeviceBuffer.from_array(np.arange(n + 1, dtype=np.int64) * 1024)
    wr, rbk = load('SQK-RBK004_read_starts')
    t, calls, probs = run(rbk, d_samples, d_offsets, n, 256, 'start', 512)
    out['configs[3] SQK-RBK004_read_starts, 125k-read shard of 1M, batch 256'] = {
        'gpu_seconds': round(t, 5), 'reads_per_s': round(n / t)}

    # ---- the CLI's default geometry: 12 windows per read ------------------------------------
    n, length = 20000, 6656
    rng = np.random.default_rng(7)
    reads = np.clip(np.rint(np.repeat(rng.normal(450, 80, (n, length // 8)), 8, axis=1) +
                            rng.normal(0, 8, (n, length))), 0, 2047).astype(np.int16)
    d_samples = hip_backend.DeviceBuffer.from_array(reads)
    d_offsets = hip_backend.DeviceBuffer.from_array(np.arange(n + 1, dtype=np.int64) * length)
    t, calls, probs = run(start, d_samples, d_offsets, n, 256, 'start', 6144)
    out['default scan_size 6144: 20k reads of 6,656 samples, batch 256'] = {
        'gpu_seconds': round(t, 5), 'reads_per_s': round(n / t),
        'windows_per_s': round(12 * n / t)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

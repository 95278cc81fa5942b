#!/usr/bin/env python3
"""
bench.py — reads classified/sec on the Deepbinner classify hot path (BASELINE.json metric).

Workload (BASELINE.json configs[1]): EXP-NBD103_read_starts, 10,000 synthetic 1024-sample int16
signals, batch 256, per GPU.  One STEP = one pass of the whole hot path over those 10,000 reads in
batches of 256 through seam b2 (`dbh_classify_i16_batched_dev`: window slice + fp64 z-normalise + the
20-conv CNN + merge + renormalise + barcode call), with `--scan_size 512` so each 1024-sample read
is exactly one full window (classify.py:401-402 accepts it) — i.e. 1 read = 1 window = one
classification.  Inputs are resident in HBM before the timed region; per-read calls are gathered
over RCCL when N > 1 (weak scaling: every rank classifies its own 10,000 reads per step).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the forward kernel (dominant, compute-bound: 33,629,952 FLOP per window on the
                fp32 matrix pipe) — average launch duration measured live with HIP events on the
                launch stream inside the timed region;
  cpu_baseline  the oracle's C restatement (oracle/dbref.c, OpenMP) on a bounded sample of the same
                reads on this box's host cores.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from deepbinner_amd import hip_backend                       # noqa: E402
from deepbinner_amd.model_format import ModelWeights         # noqa: E402
from deepbinner_amd.sharding import env_world                # noqa: E402

FLOP_PER_WINDOW = 33629952          # 16,814,976 MAC in the 20 convolutions
PEAK_FP32_TFLOPS = 157.3            # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
MODEL = 'EXP-NBD103_read_starts'
N_READS = 10000
BATCH = 256
SCAN_SIZE = 512
SCORE_DIFF = 0.5
TIMING_STRIDE = 20          # an event bracket opens at every 20th forward launch ...
TIMING_SPAN = 4             # ... and covers 4 consecutive (full, 256-window) launches


def synthetic_reads(n, seed):
    """Seeded squiggle-like int16 signals: 70 % piecewise-constant levels
    N(450,80) held 8 samples + N(0,8) noise, 25 % pure Gaussian, 5 % flat; clipped to [0,2047]."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, 1024), dtype=np.int16)
    kind = rng.random(n)
    levels = np.repeat(rng.normal(450, 80, (n, 128)), 8, axis=1) + rng.normal(0, 8, (n, 1024))
    gauss = rng.normal(450, 60, (n, 1024))
    flat = np.repeat(rng.integers(300, 700, (n, 1)), 1024, axis=1)
    sig = np.where(kind[:, None] < 0.70, levels, np.where(kind[:, None] < 0.95, gauss, flat))
    out[:] = np.clip(np.rint(sig), 0, 2047).astype(np.int16)
    return out


def cpu_baseline(weights, reads, gpu_calls, gpu_probs):
    """Time the oracle's C port on a bounded sample (about 10-20 s of CPU work)."""
    from oracle import dbref
    model = dbref.CModel(weights)
    offsets = lambda k: np.arange(k + 1, dtype=np.int64) * 1024
    # calibrate on growing warm probes (the first call also spins up the OpenMP team), then size
    # the timed sample for about 15 s of CPU work, cycling through the 10,000 reads if needed
    probe_n, rate = 256, 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        model.classify(reads[:probe_n].ravel(), offsets(probe_n), 'start', SCAN_SIZE, SCORE_DIFF)
        rate = probe_n / max(time.perf_counter() - t0, 1e-6)
        probe_n = int(min(len(reads), max(probe_n, rate * 1.0)))
    sample = int(max(1024, rate * 15))
    idx = np.arange(sample) % len(reads)
    big = np.ascontiguousarray(reads[idx])
    t0 = time.perf_counter()
    probs, calls = model.classify(big.ravel(), offsets(sample), 'start', SCAN_SIZE, SCORE_DIFF)
    dt = time.perf_counter() - t0
    gpu_calls, gpu_probs = gpu_calls[idx], gpu_probs[idx]
    agree = bool(np.array_equal(calls, gpu_calls[:sample]))
    max_dp = float(np.abs(probs - gpu_probs[:sample]).max())
    return {'value': sample / dt, 'unit': 'reads/s', 'cores': int(model.threads_used),
            'kind': 'port',
            'sample': '{} reads (the {} synthetic reads, cycled), oracle/dbref.c (gcc -O3 -fopenmp), '
                      '{} host threads of {} cpus, {:.1f} s'.format(sample, len(reads),
                                                                   model.threads_used,
                                                                   os.cpu_count(), dt),
            'calls_match_gpu': agree, 'max_abs_dp_vs_gpu': max_dp}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true',
                    help='experiment: skip the per-launch HIP events (roofline object omitted)')
    args = ap.parse_args()

    rank, local_rank, world = env_world()
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus {} but WORLD_SIZE={}'.format(args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit('launch with: python -m torch.distributed.run --nnodes=1 '
                         '--nproc-per-node {} --master-addr 127.0.0.1 bench.py --gpus {}'
                         .format(args.gpus, args.gpus))

    # DEEPBINNER_BENCH_SHARE_GPU=1 is a TEST mode for boxes with one GPU: every rank uses
    # device 0 and the gather runs over gloo on host copies (RCCL refuses two ranks per device).
    share_gpu = os.environ.get('DEEPBINNER_BENCH_SHARE_GPU') == '1'
    # DEEPBINNER_BENCH_FORCE_DIST=1 is a TEST mode too: take the multi-rank code path (process
    # group, RCCL all-gather, MAX-over-ranks) even with a single rank, so that a one-GPU box
    # exercises exactly what the N > 1 runs execute.
    use_dist = world > 1 or os.environ.get('DEEPBINNER_BENCH_FORCE_DIST') == '1'
    device = 0 if share_gpu else local_rank
    dist = torch = None
    if use_dist:
        import torch
        from deepbinner_amd.sharding import init_process_group
        if share_gpu:
            dist = init_process_group('gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist = init_process_group('nccl', local_rank)
    hip_backend.set_device(device)

    weights, _ = ModelWeights.load(os.path.join(REPO, 'deepbinner_amd', 'models', MODEL + '.dbw'))
    model = hip_backend.HipModel(weights)

    # ---- inputs resident in HBM (rank-specific seed: every GPU has its own 10,000 reads) ------
    reads = synthetic_reads(N_READS, 20260927 + rank)
    d_samples = hip_backend.DeviceBuffer.from_array(reads)
    d_offsets = hip_backend.DeviceBuffer.from_array(np.arange(N_READS + 1, dtype=np.int64) * 1024)
    d_probs = hip_backend.DeviceBuffer(N_READS * model.n_classes * 4)
    if use_dist and not share_gpu:
        calls_t = torch.empty(N_READS, dtype=torch.int32, device='cuda')
        calls_ptr = calls_t.data_ptr()
        gathered = torch.empty(world * N_READS, dtype=torch.int32, device='cuda')
    else:
        d_calls = hip_backend.DeviceBuffer(N_READS * 4)
        calls_ptr = d_calls.ptr
    # the synthetic reads are all 1,024 samples long: say so (checked on the device per read)
    if os.environ.get('DEEPBINNER_BENCH_NO_HINT') != '1':      # (A/B knob)
        model.set_read_length_hint(1024, N_READS * 1024)
    # One C-ABI call per step: the library walks the 10,000 reads in batches of 256, one fused
    # kernel launch per batch, back to back on one stream - which is also where the HIP events
    # that time every launch are recorded.
    def step():
        model.classify_batched_dev(d_samples.ptr, d_offsets.ptr, N_READS, BATCH, 'start',
                                   SCAN_SIZE, SCORE_DIFF, d_probs.ptr, calls_ptr, None)
        if use_dist and share_gpu:
            host = torch.from_numpy(d_calls.download((N_READS,), np.int32))
            out = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(out, host)
        elif use_dist:
            # In line with the classification, not overlapped with the next step's: a launch is
            # exactly one wave of 256 workgroups on 256 CUs, and a collective kernel running
            # beside it pushes some of them into a second wave (measured: +6 % per step).
            dist.all_gather_into_tensor(gathered, calls_t)

    def sync():
        hip_backend.synchronize()
        if use_dist and not share_gpu:
            torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync()
    barrier()
    sync()
    # One HIP event pair around launches 20k .. 20k+3 of the 40 launches of a step (all full,
    # 256-window launches; the 16-window tail launch is never inside a bracket).  A pair around
    # EVERY launch costs ~7 us of queue time per batch and slows what is being measured, and a
    # pair around a single launch includes ~2.5 us of dispatch latency that back-to-back launches
    # do not pay (rocprofv3's per-kernel duration is that much shorter).
    model.timing_enable(0 if args.no_kernel_timing else TIMING_STRIDE, TIMING_SPAN)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches, windows = model.timing_read()
    model.timing_enable(False)

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if share_gpu else 'cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_reads = N_READS * world * args.steps
    value = total_reads / elapsed
    result = {
        'metric': 'reads classified/sec (1024-sample windows, batch 256)',
        'value': value, 'unit': 'reads/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': 1000.0 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {'workload': 'BASELINE.json configs[1]: {} model, {} synthetic 1024-sample int16 '
                               'signals per GPU per step, batch {}, seam b2 (slice + normalise + '
                               'CNN + renormalise + call fused in one launch per batch), '
                               'scan_size {} => 1 window per read, inputs resident in HBM, uniform read '
                               'length declared (dbh_model_set_read_length_hint)'.format(MODEL, N_READS, BATCH, SCAN_SIZE),
                   'reads_per_step_per_gpu': N_READS, 'batch': BATCH, 'windows_per_read': 1,
                   'launches_per_batch': 1,
                   'parallelism': 'reads sharded, {} rank(s), RCCL all_gather of calls'.format(world)
                   if world > 1 else 'single GPU'},
    }
    if rank == 0 and args.no_kernel_timing:
        print(json.dumps(result))
    elif rank == 0:
        avg_ms = kernel_ms / max(launches, 1)
        windows_per_launch = windows / max(launches, 1)
        achieved = FLOP_PER_WINDOW * windows_per_launch / (avg_ms * 1e-3) / 1e12 if launches else 0.0
        traffic = None
        pmc_path = os.path.join(REPO, 'profiles', 'pmc_traffic.json')
        if os.path.isfile(pmc_path):
            with open(pmc_path) as f:
                traffic = json.load(f).get('hbm_bytes_per_launch')
        result['roofline'] = {
            'bound': 'mfma', 'kernel': 'dbh_forward_kernel', 'achieved': achieved,
            'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP32_TFLOPS,
            'traffic': traffic, 'avg_launch_ms': avg_ms, 'launches_timed': launches,
            'timed_every_nth_launch': TIMING_STRIDE, 'launches_per_event_bracket': TIMING_SPAN,
            'windows_per_launch': windows_per_launch,
            'algorithmic_flop_per_window': FLOP_PER_WINDOW,
            # fused seam b2: int16 samples in, fp32 probabilities + int32 call out
            'algorithmic_hbm_bytes_per_window': 1024 * 2 + model.n_classes * 4 + 4,
            'hbm_frac_at_algorithmic_bytes': (value * (1024 * 2 + model.n_classes * 4 + 4) / world)
                                             / 8.0e12,
        }
        if world == 1 and not args.no_cpu_baseline:
            if use_dist and not share_gpu:       # (forced single-rank RCCL test mode)
                gpu_calls = calls_t.cpu().numpy()
            else:
                gpu_calls = d_calls.download((N_READS,), np.int32)
            gpu_probs = d_probs.download((N_READS, model.n_classes), np.float32)
            result['cpu_baseline'] = cpu_baseline(weights, reads, gpu_calls, gpu_probs)
        result['device'] = hip_backend.device_name(device)
        print(json.dumps(result))
    barrier()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""
bench.py — reads classified/sec on the Deepbinner classify hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,3}] [--steps-per-launch S]

--config 1 (default; the configuration the metric is quoted on): BASELINE.json configs[1],
    EXP-NBD103_read_starts, 10,000 synthetic 1024-sample int16 signals per GPU, batch 256.
--config 2: configs[2], EXP-NBD103 start + end models over 100,000 signals per GPU, batch 512,
    combine_calls (require_either) on the device.
--config 3: configs[3], SQK-RBK004_read_starts, 1,000,000 signals sharded over the N GPUs
    (strong scaling: the total is fixed), batch 256, all-gather of the calls.

One STEP = one pass of the whole hot path over the configuration's reads through seam b2
(`dbh_classify_i16_batched_dev`: window slice + fp64 z-normalise + the 20-conv CNN + renormalise +
barcode call), with `--scan_size 512` so that each 1024-sample read is exactly one window
(classify.py:401-402 accepts it): 1 read = 1 window = one classification.  Inputs are resident in
HBM before the timed region; inside it every step ends with the all-gather of the per-read calls
(RCCL between GPUs when N > 1) and a copy of the gathered calls to pinned host memory.

--steps-per-launch S (alias --gather-every): S consecutive steps are queued as ONE persistent
launch (and, N > 1, one exchange of S steps' calls): the reads of every step lie in HBM S times
over, each step reads its own copy and writes its own results.  The forward kernel hands windows
to its 256 workgroups off a counter; 10,000 windows are 39.06 rounds, so a launch of ONE
configs[1] step ends with 2.3 % of the GPU idle - a boundary effect of the launch, not of the
kernel (configs[2] / [3], whose steps are 100,000 / 1,000,000 windows, do not see it).  Default:
10 for --config 1 (cut down to a divisor of --steps), 1 otherwise; `--steps-per-launch 1` is the
one-launch-per-step form of rounds 1-4 (`value_one_launch_per_step` in the line is its rate).  With
N > 1 the same switch separates what a straggling rank costs from what the collective costs: the
exchange is paid once per S steps.

N > 1 runs either way, with no torch anywhere:
  * `python bench.py --gpus N`: ONE process drives the N devices (a thread per device,
    ncclCommInitAll behind the C ABI);
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (how the driver
    launches it): one process per GPU; RANK / LOCAL_RANK / WORLD_SIZE from the environment, RCCL's
    unique id and the barrier / MAX-over-ranks of the timing over deepbinner_amd.sharding's own
    socket rendezvous.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the forward kernel (dominant, compute-bound: 33,629,952 algorithmic FLOP per
                window on the fp32 matrix pipe), average launch duration measured live with HIP
                events on the launch stream inside the timed region; `achieved` / `frac` count
                the FLOP of the MFMAs the kernel really issues (the Winograd layers issue fewer
                than the direct convolutions) = how busy the matrix pipe is, <= 1;
                `frac_algorithmic_equivalent` counts the direct convolutions' FLOP instead;
  cpu_baseline  the oracle's C restatement (oracle/dbref.c, OpenMP) on a bounded sample of the same
                reads on this box's host cores (N = 1 only): best of 3 with one thread per usable
                CPU (cgroup quota), and with the reference's default of 12 threads.
"""

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from deepbinner_amd import hip_backend                       # noqa: E402
from deepbinner_amd import sharding                          # noqa: E402
from deepbinner_amd.model_format import ModelWeights         # noqa: E402

FLOP_PER_WINDOW = 33629952          # 16,814,976 MAC in the 20 convolutions
PEAK_FP32_TFLOPS = 157.3            # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
PEAK_HBM_BYTES_PER_S = 8.0e12
SCAN_SIZE = 512
SCORE_DIFF = 0.5
# The forward kernel is persistent: one launch carries a whole step's reads through the batches
# (DEEPBINNER_LAUNCH_PER_BATCH=1: the library launches once per batch instead, for comparison).
LAUNCH_PER_BATCH = os.environ.get('DEEPBINNER_LAUNCH_PER_BATCH') == '1'
# HIP event brackets on the launch stream: every launch when there is one per step; with a launch
# per batch, one bracket at every 20th launch covering 4 consecutive (full) launches - a pair
# around EVERY short launch costs ~7 us of queue time per batch and slows what it measures, a pair
# around a single one includes ~2.5 us of dispatch latency that back-to-back launches do not pay.
TIMING_STRIDE, TIMING_SPAN = (20, 4) if LAUNCH_PER_BATCH else (1, 1)
# The headline run takes every read's length from the offsets array, as a real caller does.
# DEEPBINNER_BENCH_HINT=1 declares the uniform length (dbh_model_set_read_length_hint) instead;
# `value_with_hint` in the line is that variant's rate, a side figure.
USE_HINT = os.environ.get('DEEPBINNER_BENCH_HINT') == '1'
# README.md:213 of the reference: "about 15 reads/sec using 12 threads" (its TensorFlow CPU path,
# start + end models, on its author's laptop) - quoted beside the timed port, never a ratio's base
PUBLISHED_CPU = {'value': 15, 'unit': 'reads/s', 'threads': 12, 'source': 'README.md:213'}

CONFIGS = {
    1: {'name': 'BASELINE.json configs[1]', 'models': ['EXP-NBD103_read_starts'],
        'sides': ['start'], 'reads': 10000, 'batch': 256, 'scaling': 'weak',
        'steps_per_launch': 10},
    2: {'name': 'BASELINE.json configs[2]',
        'models': ['EXP-NBD103_read_starts', 'EXP-NBD103_read_ends'], 'sides': ['start', 'end'],
        'reads': 100000, 'batch': 512, 'scaling': 'weak', 'combine': 'require_either'},
    3: {'name': 'BASELINE.json configs[3]', 'models': ['SQK-RBK004_read_starts'],
        'sides': ['start'], 'reads': 1000000, 'batch': 256, 'scaling': 'strong'},
}


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


def real_windows(limit):
    """1024-sample windows cut from the 37 real reads of the reference's test set
    (tests/golden/reads.npz) at 64-sample shifts over their first 6,144 samples: unlike the
    synthetic squiggles (all 'none') these exercise the barcode classes."""
    path = os.path.join(REPO, 'tests', 'golden', 'reads.npz')
    if not os.path.isfile(path):
        return np.zeros((0, 1024), dtype=np.int16)
    data = np.load(path)
    cut = []
    for samples, offsets in ((data['samples'], data['offsets']),
                             (data['multi_samples'], data['multi_offsets'])):
        for i in range(len(offsets) - 1):
            read = samples[offsets[i]:offsets[i + 1]]
            for s in range(0, min(len(read) - 1024, 5120) + 1, 64):
                cut.append(read[s:s + 1024])
    cut = np.asarray(cut, dtype=np.int16)
    if len(cut) > limit:
        cut = cut[np.linspace(0, len(cut) - 1, limit).astype(np.int64)]
    return cut


def config_reads(n, seed):
    """n reads of 1,024 samples: seeded synthetic signals in chunks of 10,000 (the first 50,000
    are distinct; beyond that they repeat, rolled by the tile number), every tenth read of the
    first 10,000 replaced by a window of a real read."""
    unique = min(n, 50000)
    base = np.concatenate([synthetic_reads(min(10000, unique - a), seed + a // 10000)
                           for a in range(0, unique, 10000)])
    real = real_windows(min(unique, 10000) // 10)
    base[0:10 * len(real):10] = real
    if n <= unique:
        return base
    out = np.empty((n, 1024), dtype=np.int16)
    for tile, a in enumerate(range(0, n, unique)):
        b = min(a + unique, n)
        out[a:b] = np.roll(base[:b - a], 17 * tile, axis=1) if tile else base[:b - a]
    return out


class ShardJob:
    """One device's share of the benchmark, living on that device's thread (or on the rank's
    main thread): resident reads, per-model outputs, and the step that queues the launches."""

    def __init__(self, shard, cfg, weights, reads, block, timing_model, steps_per_launch=1):
        self.shard, self.cfg = shard, cfg
        self.models = [shard.model] + [shard.add_model(w) for w in weights[1:]]
        self.n_step = len(reads)
        # one launch carries `steps_per_launch` steps: every step has its own copy of the reads
        # in HBM (and its own results), step by step
        if steps_per_launch > 1:
            reads = np.tile(reads, (steps_per_launch, 1))
        n = len(reads)
        shard.upload(reads.reshape(-1), np.arange(n + 1, dtype=np.int64) * 1024,
                     block * steps_per_launch)
        self.n = n
        dual = len(self.models) == 2
        self.probs = [hip_backend.DeviceBuffer(max(n, 1) * m.n_classes * 4) for m in self.models]
        self.side_calls = [hip_backend.DeviceBuffer(max(n, 1) * 4) for _ in self.models] if dual else []
        if USE_HINT:                   # (A/B knob; the headline run does not declare anything)
            for m in self.models:      # all reads are 1,024 samples long: say so (checked per read)
                m.set_read_length_hint(1024, n * 1024)
        self.timing_model = self.models[0] if timing_model else None

    calls_out = None      # where the final calls go instead of the shard's device buffer
    side = None           # sharding.SideGather: the exchange on a stream of its own
    pinned = None         # lead only: where the gathered calls land on the host
    pending = None        # slot whose exchange was queued since this job's last step
    timed = None          # lead only: [(Event before the exchange, Event behind the copy)] per step

    def enable_side_gather(self, pinned):
        self.side = sharding.SideGather(self.shard)
        self.pinned = pinned

    def finish_pending(self):
        """Behind the exchange of the step before (queued on the side stream by the caller since
        this job's last step): the copy of the gathered calls to the host (lead), the slot's
        release - and the end of that step's gather bracket."""
        if self.side is None or self.pending is None:
            return
        slot, self.pending = self.pending, None
        if self.pinned is not None:
            self.pinned.fetch(self.side.gathered[slot].ptr, self.side.stream.ptr)
        if self.timed is not None:
            self.timed[-1][1].record(self.side.stream.ptr)
        self.side.release(slot)

    def step(self, slot=0):
        s, cfg = self.shard, self.cfg
        if self.side is not None:
            self.finish_pending()
            self.side.before_classify(slot)
            final = self.side.calls[slot].ptr
        else:
            final = ctypes.c_void_p(self.calls_out) if self.calls_out else s.calls.ptr
        self._classify(final)
        if self.side is not None:
            self.side.after_classify(slot)
            if self.timed is not None:
                pair = self.spare.pop() if self.spare else (hip_backend.Event(), hip_backend.Event())
                pair[0].record(self.side.stream.ptr)
                self.timed.append(pair)
            self.pending = slot

    spare = ()
    xw_spare, xw_pairs = (), None     # event pairs around this rank's exchange alone (in stream)

    def time_exchanges(self, launches):
        """From now on exchange_begin / exchange_end bracket this device's share of every
        exchange on its classification stream: from 'own calls final' to 'all-gather done' -
        what the collective costs plus what waiting for the slowest rank costs."""
        self.xw_spare = [(hip_backend.Event(), hip_backend.Event()) for _ in range(launches)]
        self.xw_pairs = []

    def exchange_begin(self):
        if self.xw_pairs is not None and self.xw_spare:
            pair = self.xw_spare.pop()
            pair[0].record(self.shard.stream.ptr)
            self.xw_pairs.append(pair)

    def exchange_end(self):
        if self.xw_pairs:
            self.xw_pairs[-1][1].record(self.shard.stream.ptr)

    def launch_stats(self):
        """This device's forward launches of the timed region (HIP events on its stream) and its
        exchange brackets: what a SCALE record needs to say which rank was the slow one."""
        ms, launches, windows = self.models[0].timing_read()
        waits = [a.elapsed_ms(b) for a, b in (self.xw_pairs or [])]
        if not waits and self.timed:
            waits = [a.elapsed_ms(b) for a, b in self.timed]
        return {'avg_launch_ms': ms / launches if launches else None, 'launches': launches,
                'total_ms': ms, 'windows': windows,
                'windows_per_launch': windows / launches if launches else None,
                'exchange_ms': sum(waits) / len(waits) if waits else None}

    def time_gathers(self, steps):
        """From now on every step's exchange + copy is bracketed by two events on the side stream
        (created here, outside the timed region)."""
        self.spare = [(hip_backend.Event(), hip_backend.Event()) for _ in range(steps)]
        self.timed = []

    def gather_ms_per_step(self):
        done = [a.elapsed_ms(b) for a, b in (self.timed or [])]
        return sum(done) / len(done) if done else None

    def _classify(self, final):
        s, cfg = self.shard, self.cfg
        if not self.side_calls:
            self.models[0].classify_batched_dev(s.samples.ptr, s.offsets.ptr, self.n, cfg['batch'],
                                                cfg['sides'][0], SCAN_SIZE, SCORE_DIFF,
                                                self.probs[0].ptr, final, s.stream.ptr)
            return
        for m, side, probs, calls in zip(self.models, cfg['sides'], self.probs, self.side_calls):
            m.classify_batched_dev(s.samples.ptr, s.offsets.ptr, self.n, cfg['batch'], side,
                                   SCAN_SIZE, SCORE_DIFF, probs.ptr, calls.ptr, s.stream.ptr)
        hip_backend.combine_calls_dev(self.side_calls[0].ptr, self.side_calls[1].ptr, self.n,
                                      cfg['combine'], final, s.stream.ptr)


class PinnedCalls:
    """Pinned host landing buffer for the gathered calls (async D2H inside the timed region)."""

    def __init__(self, count):
        self.lib = hip_backend.load_library()
        self.count = int(count)
        ptr = ctypes.c_void_p()
        hip_backend.check(self.lib.dbh_malloc_host(ctypes.byref(ptr), max(self.count, 1) * 4))
        self.ptr = ptr.value

    def device_pointer(self):
        """The address the GPU stores into this buffer at (dbh_host_device_pointer)."""
        dev = ctypes.c_void_p()
        hip_backend.check(self.lib.dbh_host_device_pointer(ctypes.c_void_p(self.ptr),
                                                           ctypes.byref(dev)),
                          'dbh_host_device_pointer')
        return dev.value

    def fetch(self, dev_ptr, stream):
        hip_backend.check(self.lib.dbh_memcpy_d2h(self.ptr, dev_ptr, self.count * 4, stream),
                          'dbh_memcpy_d2h')

    def array(self):
        return np.ctypeslib.as_array(ctypes.cast(self.ptr, ctypes.POINTER(ctypes.c_int32)),
                                     shape=(self.count,)).copy()


def cpu_baseline(cfg, weights, reads, gpu_calls, gpu_probs):
    """Time the oracle's C port on a bounded sample of the workload (about 20-25 s of CPU work in
    all): a warm-up, then the best of three runs with one thread per CPU this process may use
    (misc.usable_cpus(): the online CPUs cut down to the affinity mask and the cgroup quota - a
    container that shows 256 CPUs may be allowed 16, and a team of 256 then runs slower than a
    team of 16), and the best of two runs with 12 threads, the reference's default
    (deepbinner.py:149-156: --intra_op_parallelism_threads 12)."""
    from oracle import dbref
    from oracle import classify_ref
    from deepbinner_amd import misc
    models = [dbref.CModel(w) for w in weights]
    offsets = lambda k: np.arange(k + 1, dtype=np.int64) * 1024
    usable = misc.usable_cpus()

    def run(sample_reads, threads):
        out = [m.classify(sample_reads.ravel(), offsets(len(sample_reads)), side, SCAN_SIZE,
                          SCORE_DIFF, threads) for m, side in zip(models, cfg['sides'])]
        calls = out[0][1]
        if len(out) == 2:
            names = [['none' if c == 0 else str(int(c)) for c in o[1]] for o in out]
            final = [classify_ref.combine_calls(a, b, cfg['combine']) for a, b in zip(*names)]
            calls = np.array([0 if c == 'none' else int(c) for c in final], dtype=np.int32)
        return out[0][0], calls

    def best_of(sample_reads, threads, repeats):
        best, result = None, None
        for _ in range(repeats):
            t0 = time.perf_counter()
            result = run(sample_reads, threads)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return best, result

    # warm-up and calibration on growing probes (the first call also spins up the OpenMP team),
    # then samples sized for about 5 s per run, cycling through the reads if needed
    probe_n, rate = 256, 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        run(reads[:probe_n], usable)
        rate = probe_n / max(time.perf_counter() - t0, 1e-6)
        probe_n = int(min(len(reads), max(probe_n, rate * 1.0)))
    sample = int(max(1024, rate * 5))
    idx = np.arange(sample) % len(reads)
    big = np.ascontiguousarray(reads[idx])
    dt, (probs, calls) = best_of(big, usable, 3)
    threads = int(models[0].threads_used)
    ref_threads = 12
    sample12 = int(max(512, min(sample, sample * ref_threads // max(usable, 1))))
    dt12, _ = best_of(big[:sample12], ref_threads, 2)
    return {'value': sample / dt, 'unit': 'reads/s', 'cores': threads, 'kind': 'port',
            'cpus_online': os.cpu_count(), 'cpus_usable': usable,
            'value_12_threads': sample12 / dt12,
            'published': PUBLISHED_CPU,
            'sample': '{} reads (the first {} reads of the workload, cycled), oracle/dbref.c '
                      '(gcc -O3 -fopenmp), best of 3 runs after a warm-up with {} threads = the '
                      'CPUs this process may use (cgroup quota / affinity; {} online), {:.1f} s '
                      'per run; value_12_threads: {} reads, best of 2 with the reference\'s '
                      'default of 12 threads, {:.1f} s per run'
                      .format(sample, len(reads), threads, os.cpu_count(), dt, sample12, dt12),
            'calls_match_gpu': bool(np.array_equal(calls, gpu_calls[idx])),
            'calls_not_none_in_sample': int((calls != 0).sum()),
            'max_abs_dp_vs_gpu': float(np.abs(probs - gpu_probs[idx]).max())}


def side_rates(weights, reads):
    """Two rates of configs[1] that are NOT `value` (N = 1 only): with the uniform-read-length
    hint declared, and PCIe-inclusive through the host-buffer entry point dbh_classify_i16
    ([staging copy ->] H2D -> kernels -> D2H, groups through three slots) over 20 copies of the
    reads, from pageable and from pinned memory."""
    n = len(reads)
    model = hip_backend.HipModel(weights)
    d_samples = hip_backend.DeviceBuffer.from_array(reads)
    d_offsets = hip_backend.DeviceBuffer.from_array(np.arange(n + 1, dtype=np.int64) * 1024)
    d_probs = hip_backend.DeviceBuffer(n * model.n_classes * 4)
    d_calls = hip_backend.DeviceBuffer(n * 4)
    out = {}
    # one launch per step (the form `value` had in rounds 1-4): every launch of 10,000 windows
    # ends with its 40th round of 256 workgroups 6 % full
    best = None
    for _ in range(4):
        hip_backend.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model.classify_batched_dev(d_samples.ptr, d_offsets.ptr, n, 256, 'start', SCAN_SIZE,
                                       SCORE_DIFF, d_probs.ptr, d_calls.ptr, None)
        hip_backend.synchronize()
        dt = (time.perf_counter() - t0) / 10
        best = dt if best is None else min(best, dt)
    out['value_one_launch_per_step'] = n / best
    best = None
    model.set_read_length_hint(1024, n * 1024)
    for _ in range(4):
        hip_backend.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            model.classify_batched_dev(d_samples.ptr, d_offsets.ptr, n, 256, 'start', SCAN_SIZE,
                                       SCORE_DIFF, d_probs.ptr, d_calls.ptr, None)
        hip_backend.synchronize()
        dt = (time.perf_counter() - t0) / 10
        best = dt if best is None else min(best, dt)
    out['value_with_hint'] = n / best
    model.set_read_length_hint(0, 0)
    # Consecutive steps on TWO streams (two sets of outputs): a launch's last round - 10,000
    # windows are 39.06 rounds of 256 workgroups - and the next launch's first windows overlap.
    # Not `value`: the launches of the headline run follow each other on one stream, so that a
    # launch's duration means something (roofline.avg_launch_ms).
    streams = [hip_backend.Stream(), hip_backend.Stream()]
    outs = [(d_probs, d_calls), (hip_backend.DeviceBuffer(n * model.n_classes * 4),
                                 hip_backend.DeviceBuffer(n * 4))]
    best = None
    for _ in range(4):
        hip_backend.synchronize()
        t0 = time.perf_counter()
        for k in range(20):
            probs, calls = outs[k & 1]
            model.classify_batched_dev(d_samples.ptr, d_offsets.ptr, n, 256, 'start', SCAN_SIZE,
                                       SCORE_DIFF, probs.ptr, calls.ptr, streams[k & 1].ptr)
        for st in streams:
            st.synchronize()
        dt = (time.perf_counter() - t0) / 20
        best = dt if best is None else min(best, dt)
    out['value_two_streams'] = n / best
    out['two_streams_note'] = ('steps queued alternately on two streams with their own outputs: '
                               'the end of one launch overlaps the start of the next')
    for st in streams:
        st.close()
    tiles = 20
    big = np.ascontiguousarray(np.tile(reads, (tiles, 1))).reshape(-1)
    offsets = np.arange(n * tiles + 1, dtype=np.int64) * 1024
    lib = hip_backend.load_library()
    ptr = ctypes.c_void_p()
    hip_backend.check(lib.dbh_malloc_host(ctypes.byref(ptr), big.nbytes), 'dbh_malloc_host')
    pinned = np.ctypeslib.as_array(ctypes.cast(ptr.value, ctypes.POINTER(ctypes.c_int16)),
                                   shape=big.shape)
    pinned[:] = big
    for key, buf, where in (('value_pcie_inclusive', big, 'pageable host memory (staged through '
                             'pinned slots by a small thread team)'),
                            ('value_pcie_inclusive_pinned', pinned, 'pinned host memory (read by '
                             'the DMA engine in place: what the native loader hands over)')):
        best = None
        for _ in range(4):
            t0 = time.perf_counter()
            model.classify_packed(buf, offsets, 'start', SCAN_SIZE, SCORE_DIFF)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[key] = n * tiles / best
        out[key.replace('value_', '') + '_note'] = (
            '{} reads per dbh_classify_i16 call from {}, results back in host arrays; '
            '{:.2f} GB/s of int16 over PCIe'.format(n * tiles, where, big.nbytes / best / 1e9))
    hip_backend.check(lib.dbh_free_host(ptr), 'dbh_free_host')
    model.close()
    return out


def other_configs():
    """The other BASELINE.json configurations on this one GPU, inside the default run's budget, so that
    every configuration has a figure the driver observed (N = 1, --config 1 only):
      "2", "3": this script again as a child (`--config K --steps 3 --warmup 1`, no CPU leg): reads/s and
                roofline.frac of its line;
      "4":     the realtime stream over multi-read containers written on the spot by this package's
                own writer (4,000 reads each, deflated, 2,000-9,000 samples): `deepbinner realtime`
                (table only) over 2 and over 26 of them - the difference of the two runs takes the
                fixed costs (model loading, thread teams) out: reads/s of the 24 containers between."""
    import shutil
    import subprocess
    import tempfile
    out = {}
    for k in (2, 3):
        try:
            t0 = time.perf_counter()
            p = subprocess.run([sys.executable, os.path.abspath(__file__), '--config', str(k), '--steps', '3',
                                '--warmup', '1', '--no-cpu-baseline', '--no-side-rates', '--no-other-configs'],
                               capture_output=True, text=True, timeout=120)
            line = [l for l in p.stdout.splitlines() if l.startswith('{')][-1]
            d = json.loads(line)
            out[str(k)] = {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'],
                           'frac': d['roofline']['frac'], 'windows_per_s': d.get('windows_per_s'),
                           'workload': d['config']['workload'].split(':')[0],
                           'seconds': round(time.perf_counter() - t0, 1)}
        except Exception as e:                      # noqa: BLE001
            out[str(k)] = {'skipped': repr(e)[:200]}
    directory = tempfile.mkdtemp(prefix='dbh_bench_stream_')
    try:
        t0 = time.perf_counter()
        out['4'] = stream_rate(directory)
        out['4']['seconds'] = round(time.perf_counter() - t0, 1)
    except Exception as e:                          # noqa: BLE001
        out['4'] = {'skipped': repr(e)[:200]}
    finally:
        shutil.rmtree(directory, ignore_errors=True)
    return {'other_configs': out}


def stream_rate(directory, small=2, large=26, reads_per_container=4000):
    import contextlib
    import io
    import uuid
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    from deepbinner_amd import deepbinner as cli
    from deepbinner_amd import hdf5_write
    import deepbinner_amd.realtime as realtime
    rng = np.random.default_rng(20260929)
    pool = []
    for _ in range(1000):
        n = int(rng.integers(2000, 9000))
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

    dirs = {}
    jobs = []
    for name, count in (('small', small), ('large', large)):
        dirs[name] = os.path.join(directory, name)
        os.makedirs(dirs[name])
        jobs += [(os.path.join(dirs[name], 'stream_%02d.fast5' % c), 1000 * count + c) for c in range(count)]
    with ThreadPoolExecutor(8) as workers:
        list(workers.map(write, jobs))
    models = os.path.join(REPO, 'deepbinner_amd', 'models')
    realtime.POLL_SECONDS = 0
    os.environ['DEEPBINNER_REALTIME_TABLE_ONLY'] = '1'
    seconds = {}
    for name in ('small', 'large', 'small', 'large'):        # (the second pair is the measurement)
        argv = ['realtime', '--in_dir', dirs[name], '--out_dir', os.path.join(directory, 'out_' + name), '--stop',
                '-s', os.path.join(models, 'EXP-NBD103_read_starts.dbw'),
                '-e', os.path.join(models, 'EXP-NBD103_read_ends.dbw')]
        shutil_out = os.path.join(directory, 'out_' + name)
        if os.path.isdir(shutil_out):
            import shutil
            shutil.rmtree(shutil_out)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(argv)
        seconds[name] = time.perf_counter() - t0
        with open(os.path.join(shutil_out, 'multi_read_classifications.tsv')) as f:
            rows = sum(1 for _ in f)
        if rows != (small if name == 'small' else large) * reads_per_container:
            raise RuntimeError('%s: %d rows' % (name, rows))
    reads = (large - small) * reads_per_container
    return {'reads_per_s': reads / (seconds['large'] - seconds['small']), 'unit': 'reads/s',
            'reads': reads, 'host_share_percent': realtime.host_inflate_share(1),
            'workload': 'BASELINE.json configs[4] on one GPU: `deepbinner realtime` (start + end models, table '
                        'only) over multi-read containers of 4,000 deflated reads of 2,000-9,000 samples; the '
                        'rate of the %d containers by which two runs differ' % (large - small)}


def workload_string(cfg, direct=False, spl=1):
    """`config.workload` of the JSON line: names the BASELINE.json configuration first."""
    n_models = len(cfg['models'])
    return ('{name}: {models} model{plural}, {reads} synthetic 1024-sample int16 signals {share} '
            'per step, batch {batch}, seam b2 (slice + normalise + CNN + renormalise + call '
            'fused in one kernel, {launches}{combine}), scan_size {scan} => 1 window per read '
            'and model, inputs resident in HBM, {hint}, {calls}').format(
                calls=('calls stored by the kernel straight into pinned host memory every step '
                       '(one GPU: nothing to gather)' if direct else
                       'gathered calls copied to pinned host memory every step'),
                name=cfg['name'], models=' + '.join(cfg['models']),
                plural='s' if n_models > 1 else '', reads=cfg['reads'],
                share='in total' if cfg['scaling'] == 'strong' else 'per GPU',
                batch=cfg['batch'],
                launches=('one launch per batch' if LAUNCH_PER_BATCH else
                          'the batches of a step walked by ONE persistent launch per model'
                          if spl == 1 else
                          '{} consecutive steps - each on its own copy of the reads, with its own '
                          'results - queued as ONE persistent launch per model'.format(spl)),
                combine=', combine_calls on the device' if n_models > 1 else '',
                scan=SCAN_SIZE,
                hint=('uniform read length declared (dbh_model_set_read_length_hint)'
                      if USE_HINT else 'read lengths taken from the offsets (no hint)'))


def pmc_constants():
    """Static counters of the committed rocprofv3 PMC runs of this build (profiles/pmc_traffic.json,
    written by tools/summarise_profile.py): HBM bytes per launch and matrix-pipe busy fraction."""
    path = os.path.join(REPO, 'profiles', 'pmc_traffic.json')
    if os.path.isfile(path):
        with open(path) as f:
            return json.load(f)
    return {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', type=int, default=1, choices=sorted(CONFIGS))
    ap.add_argument('--steps-per-launch', '--gather-every', type=int, default=0,
                    dest='steps_per_launch',
                    help='steps queued as one persistent launch (and one exchange); 0 = the '
                         'configuration\'s default (10 for --config 1, else 1), cut down to a '
                         'divisor of --steps')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-side-rates', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the short runs of configs[2], [3] and [4] beside --config 1 (other_configs)')
    ap.add_argument('--no-kernel-timing', action='store_true',
                    help='experiment: skip the per-launch HIP events (roofline object omitted)')
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    rank, local_rank, env_size = sharding.env_world()
    # a launcher started one process per GPU (DEEPBINNER_BENCH_FORCE_RANKS=1: TEST mode that
    # takes this path - rendezvous, ncclCommInitRank, RCCL all-gather - with a single rank, so
    # that a one-GPU box executes exactly what the N > 1 runs execute, minus the peers)
    per_rank = env_size > 1 or os.environ.get('DEEPBINNER_BENCH_FORCE_RANKS') == '1'
    if per_rank and env_size != args.gpus:
        raise SystemExit('--gpus {} but WORLD_SIZE={}'.format(args.gpus, env_size))
    world = args.gpus
    if cfg['scaling'] == 'strong':
        bounds = [sharding.shard_bounds(cfg['reads'], world, r) for r in range(world)]
    else:
        bounds = [(0, cfg['reads'])] * world
    shard_sizes = [b - a for a, b in bounds]
    block = max(shard_sizes)
    want_spl = args.steps_per_launch if args.steps_per_launch > 0 else cfg.get('steps_per_launch', 1)
    spl = max(d for d in range(1, max(1, min(want_spl, args.steps)) + 1) if args.steps % d == 0)
    launches_timed = args.steps // spl
    launches_warmup = -(-args.warmup // spl)
    launch_block = block * spl                   # calls one device hands over per exchange
    weights = [ModelWeights.load(os.path.join(REPO, 'deepbinner_amd', 'models', m + '.dbw'))[0]
               for m in cfg['models']]

    def reads_of(r):
        if cfg['scaling'] == 'strong':      # one global set, this rank's contiguous block
            a, b = bounds[r]
            return config_reads(cfg['reads'], 20260927)[a:b] if world == 1 else \
                config_reads(b - a, 20260927 + 1000 * r)
        return config_reads(cfg['reads'], 20260927 + 1000 * r)   # weak: own reads per GPU

    rdzv = None
    if per_rank:
        rdzv = sharding.Rendezvous(rank, world)
        group = sharding.RankGroup(weights[0], rdzv)
        my_reads = reads_of(rank)
        jobs = [ShardJob(group.shard, cfg, weights, my_reads, block, True, spl)]
        group.shard_sizes = [n * spl for n in shard_sizes]
        run_all = lambda fn: [fn(jobs[0])]
        lead = jobs[0]
        all_reads0 = my_reads if rank == 0 else None
        transport = group.transport
    else:
        group = sharding.DeviceGroup(weights[0], world)
        group.shard_sizes = [n * spl for n in shard_sizes]
        all_reads0 = reads_of(0)
        jobs = group.run_indexed(lambda i: ShardJob(group.shards[i], cfg, weights,
                                                    all_reads0 if i == 0 else reads_of(i), block,
                                                    True, spl))
        run_all = lambda fn: group.run_indexed(lambda i: fn(jobs[i]))
        lead = jobs[0]
        transport = group.transport
    is_lead = rank == 0
    pinned = PinnedCalls(launch_block * world) if is_lead else None

    def fetch():
        pinned.fetch(lead.shard.gathered.ptr, lead.shard.stream.ptr)

    # One GPU has nothing to gather: its kernels store the calls straight into the pinned host
    # buffer (40 KB per 10,000 reads) and a step is nothing but its launches - a copy command
    # between two launches of a stream costs ~20 us of idle GPU (tools/step_gap.py), 1 % of a step.
    # DEEPBINNER_BENCH_COPY_CALLS=1 brings the copy back for comparison.
    direct = (world == 1 and group.comm is None and
              os.environ.get('DEEPBINNER_BENCH_COPY_CALLS') != '1')
    if direct:
        lead.calls_out = pinned.device_pointer()

    # More than one GPU (or a communicator to exercise): the exchange and the copy of the gathered
    # calls run on a SIDE stream against two sets of call arrays used in turn, so that step k + 1's
    # launch follows step k's on the classification stream with nothing in between
    # (sharding.SideGather).  DEEPBINNER_BENCH_GATHER=instream queues them on the classification
    # stream as rounds 1-3 did, =side forces the side stream; the host transport (no communicator)
    # always uses the classification stream.
    #   Measured on one GPU (tools/gather_ab.sh, profiles/r04_gather_ab.txt): device copies on
    # the side stream overlap the next launch (+0.7 % with two shards, +1.9 % with eight); an RCCL
    # all-gather there takes 1.2-1.3 ms instead of ~20 us - its kernel does not get to run beside
    # the persistent forward kernel, CUs left free for it or not - and the step grows by 8 %.  So
    # the default is the side stream for the copy transport and the classification stream for RCCL.
    where = os.environ.get('DEEPBINNER_BENCH_GATHER', 'auto')
    side_gather = (not direct and group.comm is not None and
                   (where == 'side' or (where == 'auto' and transport == 'copy')))
    if side_gather:
        run_all(lambda j: j.enable_side_gather(pinned if (is_lead and j is lead) else None))
    # (experiment: CUs left out of the forward launches, for a collective's kernel to run beside them)
    reserve = int(os.environ.get('DEEPBINNER_BENCH_RESERVE_CUS', '0') or 0)
    if reserve:
        run_all(lambda j: [m.reserve_cus(reserve) for m in j.models])
    steps_queued = [0]
    instream_pairs, instream_spare = None, []      # event brackets around exchange + copy, in stream

    def step():
        slot = steps_queued[0] & 1
        steps_queued[0] += 1
        if side_gather:
            run_all(lambda j: j.step(slot))
            group.all_gather_side([j.side for j in jobs], slot)
            return
        run_all(lambda j: j.step())
        if direct:
            return
        run_all(lambda j: j.exchange_begin())
        bracket = None
        if instream_pairs is not None and instream_spare:      # (rank 0, one process per GPU)
            bracket = instream_spare.pop()
            bracket[0].record(lead.shard.stream.ptr)
        group.all_gather()
        run_all(lambda j: j.exchange_end())
        if is_lead:       # the gathered calls reach the host inside the timed region
            fetch() if per_rank else group.run_on(0, fetch)
        if bracket is not None:
            bracket[1].record(lead.shard.stream.ptr)
            instream_pairs.append(bracket)

    def sync():
        def drain(j):
            j.finish_pending()
            j.shard.synchronize()
            if j.side is not None:
                j.side.synchronize()
        run_all(drain)

    def barrier():
        if rdzv is not None:
            rdzv.barrier()

    for _ in range(launches_warmup):
        step()
    sync()
    barrier()
    sync()
    if side_gather:
        run_all(lambda j: j.time_gathers(launches_timed))
    elif not direct:
        run_all(lambda j: j.time_exchanges(launches_timed))
        if per_rank and is_lead and group.comm is not None:
            instream_spare.extend((hip_backend.Event(), hip_backend.Event())
                                  for _ in range(launches_timed))
            instream_pairs = []
    # every device times its own forward launches (events on its stream): the line says which
    # rank was the slow one
    run_all(lambda j: j.models[0].timing_enable(0 if args.no_kernel_timing else TIMING_STRIDE,
                                                TIMING_SPAN))
    if lead.timing_model is not None:
        # two stores per workgroup and launch: the shader clock against the 100 MHz wall clock
        lead.timing_model.clock_enable(not args.no_kernel_timing)
    t0 = time.perf_counter()
    for _ in range(launches_timed):
        step()
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0
    kernel_ms = launches = windows = 0
    shader_ghz = None
    phase_cycles, phase_groups = None, 0
    per_device = run_all(lambda j: j.launch_stats())
    own = per_device[0]         # (the lead's: device 0 of this process)
    if rdzv is not None:        # one process per GPU: every rank's figures travel to all
        per_device = [json.loads(p.decode())[0]
                      for p in rdzv.all_gather(json.dumps(per_device).encode())]
    if lead.timing_model is not None:
        kernel_ms, launches, windows = own['total_ms'], own['launches'], own['windows']
        run_all(lambda j: j.models[0].timing_enable(False))
        if not args.no_kernel_timing:
            shader_ghz = lead.timing_model.clock_read()      # of the timed region's last launch
            try:
                # the phases of a group of four windows (dbh_forward_phases_read): ONE more launch, behind
                # the timed region - the stamps cost about 1 % of the kernel's time
                lead.timing_model.phases_enable(True)
                step()
                sync()
                phase_cycles, phase_groups = lead.timing_model.phases_read()
            except Exception:                                   # noqa: BLE001
                phase_cycles, phase_groups = None, 0
            lead.timing_model.phases_enable(False)
            lead.timing_model.clock_enable(False)
    if rdzv is not None:
        elapsed = rdzv.max_float(elapsed)

    n_models = len(cfg['models'])
    reads_per_step = sum(shard_sizes)
    value = reads_per_step * args.steps / elapsed
    result = {
        'metric': 'reads classified/sec (1024-sample windows, batch 256)',
        'value': value, 'unit': 'reads/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'warmup_steps_run': launches_warmup * spl,
        'ms_per_step': 1000.0 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': cfg['scaling'], 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': workload_string(cfg, direct, spl),
            'reads_per_step': reads_per_step, 'reads_per_step_per_gpu': shard_sizes,
            'batch': cfg['batch'], 'windows_per_read': n_models,
            'steps_per_launch': spl, 'launches_in_timed_region': launches_timed,
            'forward_launches_per_step': (n_models * -(-max(shard_sizes) // cfg['batch'])
                                          if LAUNCH_PER_BATCH else n_models),
            'real_read_windows_in_first_10000': int(len(real_windows(1000))),
            'parallelism': ('reads sharded over {} GPUs, {}, all-gather of int32 calls over {}'
                            .format(world, 'one process per GPU' if per_rank
                                    else 'one process, a thread per device', transport)
                            if world > 1 else 'single GPU')},
        'windows_per_s': value * n_models,
        'gather': {'transport': transport if (world > 1 or group.comm is not None) else 'none',
                   'fallback_reason': group.fallback_reason},
    }
    if side_gather or not direct:
        # where the exchange is queued, and (side stream, rank 0's GPU) how long one step's
        # exchange + copy of the gathered calls to the host takes from the moment that step's
        # calls are final - it overlaps the next step's launch
        result['gather']['queued_on'] = ('side stream, call arrays double-buffered' if side_gather
                                         else 'classification stream')
        per_exchange = None
        if side_gather and is_lead:
            per_exchange = run_all(lambda j: j.gather_ms_per_step() if j is lead else None)[0]
        elif instream_pairs:
            per_exchange = (sum(a.elapsed_ms(b) for a, b in instream_pairs) / len(instream_pairs))
        # (rank 0: exchange + copy of the gathered calls to the host; one exchange per launch)
        result['gather']['steps_per_exchange'] = spl
        result['gather']['ms_per_exchange'] = per_exchange
        result['gather']['ms_per_step'] = per_exchange / spl if per_exchange is not None else None
        # every rank's own bracket from 'own calls final' to 'all-gather done' on its stream: the
        # collective itself plus the wait for the slowest rank (with one exchange per S steps,
        # --steps-per-launch S, the two separate: the wait grows with S, the collective does not)
        waits = [d.get('exchange_ms') for d in per_device]
        if all(w is not None for w in waits) and waits:
            result['gather']['wait_ms_per_exchange'] = {
                'per_rank': waits, 'min': min(waits), 'max': max(waits),
                'mean': sum(waits) / len(waits), 'longest_on_rank': int(np.argmax(waits))}
            result['gather']['wait_ms_per_step'] = max(waits) / spl
    if is_lead:
        calls_host = pinned.array()
        # (every device's block holds steps_per_launch passes; the first pass is enough here)
        gathered = np.concatenate([calls_host[r * launch_block:r * launch_block + shard_sizes[r]]
                                   for r in range(world)])
        if spl > 1:       # every step of a launch must have given the same calls
            agree = True
            for r in range(world):
                passes = calls_host[r * launch_block:r * launch_block + spl * shard_sizes[r]]
                passes = passes.reshape(spl, shard_sizes[r])
                agree = agree and bool((passes == passes[0]).all())
            result['steps_of_a_launch_agree'] = agree
        result['calls_not_none_rank0'] = int((gathered[:shard_sizes[0]] != 0).sum())
    launch_ms_per_rank = None
    per_rank_ms = [d.get('avg_launch_ms') for d in per_device]
    if per_rank_ms and all(v is not None for v in per_rank_ms):
        launch_ms_per_rank = {'per_rank': per_rank_ms, 'min': min(per_rank_ms),
                              'max': max(per_rank_ms), 'slowest_rank': int(np.argmax(per_rank_ms)),
                              'fastest_rank': int(np.argmin(per_rank_ms))}
    if is_lead and not args.no_kernel_timing:
        avg_ms = kernel_ms / max(launches, 1)
        windows_per_launch = windows / max(launches, 1)
        rate = windows_per_launch / (avg_ms * 1e-3) if launches else 0.0
        achieved = FLOP_PER_WINDOW * rate / 1e12
        mfmas, executed_flop = hip_backend.forward_executed_mfmas(lead.models[0].n_classes)
        bytes_per_window = 1024 * 2 + lead.models[0].n_classes * 4 + 4
        pmc = pmc_constants()
        executed = executed_flop * rate / 1e12
        result['roofline'] = {
            # `achieved` / `frac`: the FLOP of the MFMAs the kernel really issues (9,588 per window
            # x 2,048: the Winograd layers issue fewer than the direct convolutions would) over the
            # fp32 matrix peak = how busy the matrix pipe is.  The direct convolutions' 33.6 MFLOP
            # per window over the same time are the *_algorithmic_equivalent figures (> peak).
            'bound': 'mfma', 'kernel': 'dbh_forward_kernel', 'achieved': executed,
            'peak': PEAK_FP32_TFLOPS, 'unit': 'TFLOP/s', 'frac': executed / PEAK_FP32_TFLOPS,
            'frac_counts': 'executed MFMA FLOP (executed_mfma_per_window x 2,048) per launch over '
                           'the launch time',
            'achieved_algorithmic_equivalent': achieved,
            'frac_algorithmic_equivalent': achieved / PEAK_FP32_TFLOPS,
            'executed_mfma_per_window': mfmas, 'executed_flop_per_window': executed_flop,
            'achieved_executed': executed,
            'frac_executed': executed / PEAK_FP32_TFLOPS,
            # (how busy the matrix pipe is = frac_executed: an MFMA of this kind holds a SIMD's pipe for
            # 32 cycles and does 2,048 FLOP, which is what the peak counts; rounds 2-5 also printed a
            # figure scaled from a committed PMC run - a constant of the build, not of the run: dropped)
            # The peak above is the data sheet's, at 2.4 GHz.  Under this kernel the shader clock
            # runs lower (power management); measured inside the timed region's last launch by
            # every workgroup (s_memtime against the 100 MHz s_memrealtime, median).  In CYCLES -
            # which is what "how busy is the pipe" means - the matrix pipe is this busy:
            'shader_clock_ghz': shader_ghz,
            # shader cycles per GROUP of four windows and phase, steady-state groups of one more launch
            # behind the timed region, measured by the shipped kernel itself (one lane's clock reads
            # behind barriers; with the stamps on a launch runs ~1 % slower): stages A-C of its windows, the stage D-E chain, stage F of its windows,
            # the batched tail (every second group), between two groups
            'phase_cycles_per_group': (
                dict(zip(['stages_a_c', 'chain_d_e', 'stage_f', 'tail', 'between', 'f_mfma', 'f_barrier', 'f_reduce', 'f_end_barrier',
                          'ac_top', 'ac_chain', 'ac_conv7_first_three', 'conv7_last', 'conv7_last_to_chain'],
                         [round(c, 1) for c in phase_cycles]), groups=phase_groups)
                if phase_groups else None),
            'peak_at_shader_clock': PEAK_FP32_TFLOPS * shader_ghz / 2.4 if shader_ghz else None,
            'frac_executed_at_shader_clock': (
                executed_flop * rate / 1e12 / (PEAK_FP32_TFLOPS * shader_ghz / 2.4)
                if shader_ghz else None),
            # HBM bytes per launch from the PMC passes, scaled to this run's windows per launch
            'traffic': (pmc['hbm_bytes_per_launch'] * windows_per_launch / pmc['windows_per_launch']
                        if 'hbm_bytes_per_launch' in pmc and windows_per_launch else None),
            'traffic_algorithmic': bytes_per_window * windows_per_launch,
            # (a launch carries steps_per_launch steps: avg_launch_ms / steps_per_launch is the
            # kernel time of ONE step, to hold against ms_per_step)
            'steps_per_launch': spl, 'avg_launch_ms_per_step': avg_ms / spl,
            'avg_launch_ms': avg_ms, 'launches_timed': launches,
            'avg_launch_ms_per_rank': launch_ms_per_rank,
            'timed_every_nth_launch': TIMING_STRIDE, 'launches_per_event_bracket': TIMING_SPAN,
            'windows_per_launch': windows_per_launch,
            'algorithmic_flop_per_window': FLOP_PER_WINDOW,
            # fused seam b2: int16 samples in, fp32 probabilities + int32 call out
            'algorithmic_hbm_bytes_per_window': bytes_per_window,
            'hbm_frac_at_algorithmic_bytes': rate * bytes_per_window / PEAK_HBM_BYTES_PER_S,
        }
    if is_lead and world == 1:
        if args.config == 1 and not args.no_side_rates:
            result.update(side_rates(weights[0], all_reads0))
        if args.config == 1 and not args.no_other_configs and not args.no_side_rates:
            result['_other_configs_pending'] = True
        if not args.no_cpu_baseline:
            n0 = shard_sizes[0]
            gpu_probs = lead.probs[0].download((n0, lead.models[0].n_classes), np.float32)
            result['cpu_baseline'] = cpu_baseline(cfg, weights, all_reads0[:min(n0, 50000)],
                                                  gathered[:n0], gpu_probs)
    pending = bool(is_lead and result.pop('_other_configs_pending', False))
    if is_lead:
        result['device'] = hip_backend.device_name(lead.shard.device)
    barrier()
    group.close()
    if rdzv is not None:
        rdzv.close()
    if pending:
        # (behind everything of this configuration's own: its models and buffers are released)
        result.update(other_configs())
    if is_lead:
        print(json.dumps(result))
        sys.stdout.flush()


if __name__ == '__main__':
    try:
        main()
    except (hip_backend.HipBackendError, sharding.RendezvousError) as e:
        # (e.g. `--gpus 8` on a box with fewer devices: one line, not a traceback per thread)
        raise SystemExit('bench.py: {}'.format(e))

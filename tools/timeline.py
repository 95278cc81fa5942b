#!/usr/bin/env python3
"""Cycle-accurate phase timeline of the forward kernel (dbh_forward_timeline): average, over all
workgroups of one 256-window launch, of the per-phase durations seen by the slowest wave.
Usage: python tools/timeline.py [n_windows]      (DEEPBINNER_TIMELINE_FUSED=1: the fused seam-b2
mode - int16 reads in - instead of normalised fp32 windows)"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepbinner_amd import hip_backend                      # noqa: E402
from deepbinner_amd.model_format import ModelWeights        # noqa: E402

NAMES = {0: 'start', 1: 'A done (samples normalised; conv1 runs inside conv2 tile 0)'}
# stage B (dbh_forward.hip: stage_b_chain): nine tiles back to back, T = 3 layer + tile
NAMES.update({2: 'conv2 tile 0 done (conv1 inside)', 3: 'conv2 tile 1 may start', 4: 'conv2 tile 1 done',
              5: 'conv2 tile 2 done', 6: 'conv3 entry waits passed', 7: 'conv3 tile 0 done',
              8: 'conv3 tile 1 done', 9: 'conv3 tile 2 done', 10: 'conv4 entry waits passed',
              11: 'conv4 tile 0 done', 12: 'conv4 tile 1 done',
              13: 'conv4 end (tile 2, last epilogue, conv5 weights known)'})
LAYERS = ['conv5', 'conv6', 'conv7', 'conv8', 'conv9']
for i, l in enumerate(LAYERS):
    for j, what in enumerate(['mfma done', 'barrier1', 'epilogue done', 'barrier2']):
        NAMES[14 + 4 * i + j] = '%s %s' % (l, what)
# conv5 and conv6 run at the end of stage B's chain, in registers
NAMES.update({14: 'conv6: weights asked for, edge rows posted', 15: 'conv6 first 24 MFMAs (no halo) issued',
              16: 'conv6 halo rows arrived', 17: 'conv6 all MFMAs issued', 18: '-', 19: 'conv6 rows stored',
              20: '-', 21: 'conv6 barrier passed'})
NAMES.update({22: 'conv7 own N tile done', 23: 'conv7 barrier', 24: 'conv7 shared tile done, partials published', 25: 'conv7 end',
              34: 'E top: requests, zero rows (no barrier)', 35: 'E1 compute', 36: 'E1 barrier', 37: 'E2 compute',
              38: 'E2 barrier', 39: 'E3 compute', 40: 'E3 barrier'})
for j, what in enumerate(['partial done', 'barrier', 'reduce+epilogue (to global)', 'end']):
    NAMES[41 + j] = 'conv17 %s' % what
# the batched tail (every 8th window of a workgroup, or its last): one wave per window
TAIL = {45: 'tail: barrier (conv17 out)', 46: 'tail: X loaded, weights landed', 47: 'tail: conv18 done',
        49: 'tail: conv19 done', 53: 'tail: conv20+softmax+call', 55: 'tail: end barrier'}
ORDER = list(range(0, 45))
EXTRA = {25: 'conv7 end', 26: 'conv8 exchange stored', 27: 'conv8 barrier1',
         51: 'A: first barrier passed', 54: 'A: window normalised',
         57: 'conv4 tile 1 waits passed', 58: 'conv4 tile 2 MFMAs done'}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    w, _ = ModelWeights.load(os.path.join(REPO, 'deepbinner_amd', 'models',
                                          'EXP-NBD103_read_starts.dbw'))
    model = hip_backend.HipModel(w, device=0)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, 1024)).astype(np.float32)
    fused = os.environ.get('DEEPBINNER_TIMELINE_FUSED') == '1'
    if fused:   # seam-b2 mode: int16 reads of 1,024 samples in, slice + normalise fused in stage A
        x = np.clip(np.rint(x * 60 + 450), 0, 2047).astype(np.int16)
        if os.environ.get('DEEPBINNER_TIMELINE_HINT') == '1':
            model.set_read_length_hint(1024, x.size)
    run = model.timeline_i16 if fused else model.timeline
    run(x)                                  # warm-up
    st = run(x)                             # [n, 8 waves, 64]
    # stamps are the low 32 bits of the counters (0 = not stamped): rebase them on the run's first
    # stamp modulo 2^32 (+ 1, so that 0 still means "not stamped")
    st = st.astype(np.int64)
    for lo, hi in ((0, 62), (62, 64)):
        blk = st[:, :, lo:hi]
        base = st[0, 0, lo]
        blk[...] = np.where(blk != 0, ((blk - base + (1 << 31)) & 0xFFFFFFFF) + 1, 0)
    ids = ORDER
    # shader clock while the kernel runs: stamps 0 / 44 are s_memtime (shader clock), 62 / 63
    # s_memrealtime (constant 100 MHz) at the same two places of every window
    d_shader = (st[:, :, 44] - st[:, :, 0]).astype(np.float64)
    d_real = (st[:, :, 63] - st[:, :, 62]).astype(np.float64)
    ok = (d_real > 0) & (d_shader > 0)
    if ok.any():
        print('shader clock during the run: %.3f GHz (s_memtime against the 100 MHz s_memrealtime, '
              'median over %d wave-windows)' % (np.median(d_shader[ok] / d_real[ok]) * 0.1, ok.sum()))
    if n > 256:
        # persistent launch: window w is the (w // grid)-th of workgroup w % grid; leave out every
        # workgroup's first window (cold) and report the steady state, plus the period from one
        # window's start to the next one's on the same workgroup
        grid = 256
        starts = st[:, :, 0].min(axis=1)
        period = (starts[grid:] - starts[:-grid])
        print('steady-state period per window (start to start, same workgroup): %.0f cycles '
              '(first windows: %.0f)' % (np.median(period[grid:]), np.median(period[:grid])))
        st = st[grid:]
    t0 = st[:, :, 0].min(axis=1, keepdims=True)          # block start
    rel = st[:, :, ids] - t0[:, :, None]                 # cycles since block start
    last = rel.max(axis=1)                               # slowest wave reaches each mark
    first = rel.min(axis=1)
    mean_last = last.mean(axis=0)
    mean_first = first.mean(axis=0)
    prev = 0.0
    rows = []
    for k, i in enumerate(ids):
        rows.append((NAMES[i], mean_last[k], mean_last[k] - prev, mean_last[k] - mean_first[k]))
        prev = mean_last[k]
    total = mean_last[-1]
    print('%-26s %10s %9s %7s %10s' % ('mark', 'cum_cycles', 'delta', 'pct', 'wave_skew'))
    for name, cum, d, skew in rows:
        print('%-26s %10.0f %9.0f %6.1f%% %10.0f' % (name, cum, d, 100 * d / total, skew))
    if os.environ.get('DEEPBINNER_TIMELINE_WAVES') == '1':
        # per wave (waves w and w + 4 share a SIMD): when each reaches the marks of stage B
        print('%-26s' % 'mark (mean cycle per wave)' + ''.join('%9s' % ('w%d' % w) for w in range(8)))
        for k, i in enumerate(ids):
            if 1 <= i <= 44:
                print('%-26s' % NAMES[i] + ''.join('%9.0f' % rel[:, w, k].mean() for w in range(8)))
    # the batched tail: windows whose workgroup ran it right after them (mark 55 stamped)
    ran = st[:, :, 55].max(axis=1) > 0
    if ran.any():
        tl = st[ran]
        base = tl[:, :, 44].max(axis=1)                   # slowest wave leaves conv17
        prev = base
        print('batched tail (%d of %d windows end a batch), cycles after conv17:' % (ran.sum(), len(st)))
        for i in sorted(TAIL):
            stamps = np.where(tl[:, :, i] > 0, tl[:, :, i], 0).max(axis=1)
            print('  %-34s %8.0f  (+%.0f)' % (TAIL[i], (stamps - base).mean(), (stamps - prev).mean()))
            prev = stamps
    for i, name in EXTRA.items():
        r = st[:, :, i] - t0
        print('%-26s first wave %8.0f  last wave %8.0f   waves 0-3 %8.0f  waves 4-7 %8.0f' % (
            name, r.min(axis=1).mean(), r.max(axis=1).mean(), r[:, :4].mean(), r[:, 4:].mean()))
    print(json.dumps({'total_cycles': float(total), 'n_windows': n}))


if __name__ == '__main__':
    main()

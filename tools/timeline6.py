#!/usr/bin/env python3
"""Phase timeline of the round-6 forward kernel (groups of four windows): a persistent launch of
the stamped build (dbh_forward_timeline_i16, debug_stage 301: 256 workgroups, fixed shares, the
batched tail behind every group), averaged over the steady-state groups of every workgroup.
For every phase: cycles from the phase's first stamp to its last, as seen by the LAST wave to get
there (max over the waves of the stamp, then differences).
Usage: python tools/timeline6.py [groups_per_workgroup]"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepbinner_amd import hip_backend                      # noqa: E402
from deepbinner_amd.model_format import ModelWeights        # noqa: E402

# stamp ids (dbh_forward.hip: mark)
AC = [(0, 'A start'), (1, 'A: samples normalised'), (2, 'conv2 tile 0 (conv1 inside)'), (5, 'conv2 done'),
      (9, 'conv3 done'), (13, 'conv4 done (+conv5)'), (21, 'conv6 done, barrier passed'),
      (23, 'conv7 mid barrier'), (25, 'conv7 end (park written, barrier)')]
D = [(26, 'D: operands loaded, edge rows posted'), (27, 'conv8 tile 0'), (28, 'conv8 tile 1'), (29, 'conv8 tile 2'),
     (30, 'conv9 tile 0'), (31, 'conv9 tile 1'), (32, 'conv9 tile 2 (X in registers)'),
     (34, 'E: conv12, conv14'), (35, 'E: conv10 products'), (36, 'E: conv13'), (37, 'E: conv15'),
     (38, 'E: conv10 average, out'), (39, 'E: conv11'), (40, 'E: conv16'), (33, 'D+E closing barrier')]
EF = [(42, 'F: conv17 MFMAs + barrier'), (43, 'F: conv17 reduce + store')]
TAIL = [(45, 'tail: first barrier'), (46, 'tail: X loaded, weights landed'), (49, 'tail: conv18, conv19'),
        (53, 'tail: conv20, softmax, call'), (55, 'tail: end barrier')]


def main():
    groups = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    w, _ = ModelWeights.load(os.path.join(REPO, 'deepbinner_amd', 'models', 'EXP-NBD103_read_starts.dbw'))
    model = hip_backend.HipModel(w, device=0)
    grid = 256
    n = grid * 4 * groups
    rng = np.random.default_rng(0)
    x = np.clip(np.rint(rng.standard_normal((n, 1024)) * 60 + 450), 0, 2047).astype(np.int16)
    model.set_read_length_hint(1024, x.size)
    model.timeline_i16(x)
    st = model.timeline_i16(x).astype(np.int64)          # [n, 8 waves, 64 ids], low 32 bits of the counter
    out = {'windows': n, 'groups_per_workgroup': groups}

    def at(win, ident, how):
        v = st[win, :, ident]
        v = v[v != 0]
        if v.size == 0:
            return None
        return int(v.max() if how == 'max' else v.min())

    def diff(a, b):                                     # 32-bit wrap-around
        return None if a is None or b is None else int((b - a) & 0xFFFFFFFF)

    rows = {}
    def add(name, val):
        if val is not None and val < (1 << 30):
            rows.setdefault(name, []).append(val)

    group_cycles = []
    for b in range(grid):
        for g in range(1, groups - 1):                   # steady state: not the first, not the last group
            w0 = (g * grid + b) * 4
            # stages A-C of each window (the last window's conv7 is stamped by stage D's waves: its
            # stamps 23, 25 lie in the rows of all four windows)
            def group_at(ident, ref):
                vals = [at(w0 + k, ident, 'max') for k in range(4)]
                vals = [v for v in vals if v is not None]
                return max(vals, key=lambda v: (v - ref) & 0xFFFFFFFF) if vals else None
            d_start = None
            for k in range(4):
                win = w0 + k
                prev = at(win, 0, 'min')
                start = prev
                for ident, name in AC[1:]:
                    if k == 3 and ident in (23, 25):      # (the last window's conv7: stamps 60, 61)
                        cur = group_at(60 if ident == 23 else 61, start)
                    else:
                        cur = at(win, ident, 'max')
                    add('AC: ' + name, diff(prev, cur))
                    prev = cur
                add('AC total (window)', diff(start, prev))
                d_start = prev
            stamps = {}
            for ident, _ in D:
                vals = [at(w0 + k, ident, 'max') for k in range(4)]
                vals = [v for v in vals if v is not None]
                stamps[ident] = max(vals, key=lambda v: (v - d_start) & 0xFFFFFFFF) if vals else None
            prev = d_start
            for ident, name in D:
                add('D: ' + name, diff(prev, stamps[ident]))
                prev = stamps[ident]
            add('D+E total (group)', diff(d_start, stamps[33]))
            # stages E, F of each window
            prev_end = stamps[33]
            for k in range(4):
                win = w0 + k
                prev = prev_end
                for ident, name in EF:
                    cur = at(win, ident, 'max')
                    add('EF: ' + name, diff(prev, cur))
                    prev = cur
                add('F total (window)', diff(prev_end, prev))
                prev_end = prev
            # the batched tail (behind every group in this mode)
            prev = prev_end
            for ident, name in TAIL:
                vals = [at(w0 + k, ident, 'max') for k in range(4)]
                vals = [v for v in vals if v is not None]
                cur = max(vals, key=lambda v: (v - prev_end) & 0xFFFFFFFF) if vals else None
                add('tail: ' + name, diff(prev, cur))
                prev = cur
            add('tail total (group)', diff(prev_end, prev))
            nxt = at(((g + 1) * grid + b) * 4, 0, 'min')
            add('group: first stamp to the next group\'s first', diff(at(w0, 0, 'min'), nxt))
    for name, vals in rows.items():
        a = np.array(vals, dtype=np.float64)
        out[name] = {'mean': round(float(a.mean()), 1), 'p10': round(float(np.percentile(a, 10)), 1),
                     'p90': round(float(np.percentile(a, 90)), 1), 'n': int(a.size)}
        print('%-52s mean %9.1f   p10 %9.1f   p90 %9.1f   (%d)' % (name, a.mean(), np.percentile(a, 10),
                                                                 np.percentile(a, 90), a.size))
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(REPO, 'gpurun_out', 'timeline6.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Per-stage hardware counters of the forward kernel (GPU box, under rocprofv3 --pmc).

`run`: launches the kernel truncated after stage A, B, C, D, E (dbh_forward_truncated_dev) and
whole (dbh_predict_dev), REPS launches each, in that order, on N windows.  Under
`rocprofv3 --pmc <counters> --kernel-trace` every launch is one row per counter of the CSV.
`report <pmc_counter_collection.csv> [...]`: groups the forward kernel's dispatches by that order
(the first launch of a group is dropped), averages, and differences consecutive groups: what each
stage adds per window.  tools/stage_pmc.sh drives both.  (What a truncated launch contains: "A" =
the window's top + conv2's tile 0 with conv1 inside; "A+B" ends behind conv6 - conv5 and conv6
run on the end of stage B's chain since round 5 -, so "C" is conv7 alone.)
Usage: python tools/stage_pmc.py run [n_windows] | report <csv>..."""
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REPS = 5
GROUPS = ['A', 'A+B', 'A-C', 'A-D', 'A-E', 'whole']
N_DEFAULT = 5120


def run(n):
    import numpy as np
    from deepbinner_amd import hip_backend
    from deepbinner_amd.model_format import ModelWeights
    w, _ = ModelWeights.load(os.path.join(REPO, 'deepbinner_amd', 'models',
                                          'EXP-NBD103_read_starts.dbw'))
    model = hip_backend.HipModel(w, device=0)
    rng = np.random.default_rng(0)
    x = hip_backend.DeviceBuffer.from_array(rng.standard_normal((n, 1024)).astype(np.float32))
    probs = hip_backend.DeviceBuffer(n * 13 * 4)
    for stage in (0, 1, 2, 3, 4, None):
        for _ in range(REPS):
            if stage is None:
                model.predict_dev(x.ptr, n, probs.ptr)
            else:
                model.forward_truncated_dev(x.ptr, n, stage)
            hip_backend.synchronize()
    print('launched', len(GROUPS), 'groups of', REPS, 'on', n, 'windows')


def report(paths, n):
    out = {}
    for path in paths:
        per = {}
        with open(path) as f:
            for row in csv.DictReader(f):
                if 'dbh_forward_kernel' not in row['Kernel_Name'] or 'timeline' in row['Kernel_Name']:
                    continue
                per.setdefault(row['Counter_Name'], []).append(
                    (int(row['Dispatch_Id']), float(row['Counter_Value'])))
        for name, rows in per.items():
            rows.sort()
            values = [v for _, v in rows]
            if len(values) != REPS * len(GROUPS):
                out[name] = {'error': '%d dispatches, expected %d' % (len(values), REPS * len(GROUPS))}
                continue
            means = [sum(values[g * REPS + 1:(g + 1) * REPS]) / (REPS - 1) / n
                     for g in range(len(GROUPS))]
            added = [means[0]] + [means[g] - means[g - 1] for g in range(1, len(GROUPS))]
            out[name] = {'cumulative_per_window': dict(zip(GROUPS, [round(m, 1) for m in means])),
                         'added_per_window': dict(zip(['A', 'B', 'C', 'D', 'E', 'F-H'],
                                                      [round(a, 1) for a in added]))}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(int(sys.argv[2]) if len(sys.argv) > 2 else N_DEFAULT)
    else:
        report(sys.argv[2:], N_DEFAULT)

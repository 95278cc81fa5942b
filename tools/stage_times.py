#!/usr/bin/env python3
"""Attribute forward-kernel time to network stages on a real MI355X: run the kernel truncated
after each stage (dbh_forward_truncated_dev) and difference the HIP-event times.
(Since round 4 conv1 runs inside conv2's first tile: a kernel truncated "after stage A" ends behind
that tile, so A carries a third of conv2's MFMAs and B lacks them.  Since round 5 conv5 and conv6
run on the end of stage B's chain in registers: a kernel truncated "after stage B" ends behind
conv6, so B carries their MFMAs and C is conv7 alone.)
Usage: python tools/stage_times.py [n_windows] [repeats]"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepbinner_amd import hip_backend                      # noqa: E402
from deepbinner_amd.model_format import ModelWeights        # noqa: E402

# MACs per window per stage
STAGE_MACS = {'A': 73728, 'B': 3 * 3538944, 'C': 196608 + 589824 + 1769472, 'D': 2 * 884736,
              'E': 147456 * 4 + 49152 * 2 + 442368, 'F': 442368, 'G': 2 * 110592, 'H': 4992}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    w, _ = ModelWeights.load(os.path.join(REPO, 'deepbinner_amd', 'models',
                                          'EXP-NBD103_read_starts.dbw'))
    model = hip_backend.HipModel(w, device=0)
    rng = np.random.default_rng(0)
    x = hip_backend.DeviceBuffer.from_array(rng.standard_normal((n, 1024)).astype(np.float32))
    probs = hip_backend.DeviceBuffer(n * 13 * 4)
    out = {}
    cum = []
    for stage in list(range(7)) + [None]:
        for timed in (False, True):
            model.timing_enable(timed)
            for _ in range(reps):
                if stage is None:
                    model.predict_dev(x.ptr, n, probs.ptr)
                else:
                    model.forward_truncated_dev(x.ptr, n, stage)
            hip_backend.synchronize()
        ms, launches, _ = model.timing_read()
        cum.append(1000.0 * ms / launches)
    names = 'ABCDEFGH'
    prev = 0.0
    total = cum[-1]
    for name, c in zip(names, cum):
        us = c - prev
        flops = 2 * STAGE_MACS[name] * n
        out[name] = {'us': round(us, 2), 'cum_us': round(c, 2), 'pct': round(100 * us / total, 1),
                     'tflops': round(flops / (us * 1e-6) / 1e12, 1) if us > 0 else None}
        prev = c
    out['total_us'] = round(total, 2)
    out['n_windows'] = n
    print(json.dumps(out))
    for name in names:
        print(name, out[name])


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Per-phase cycle totals of the persistent forward kernel, production-like: the timeline build's
profile mode (debug_stage 302) reads the clock at seven phase boundaries per window and keeps the
sums in registers - no stores, no extra waits at barriers - so the phases cost what they cost in a
real launch (the per-mark stamps of tools/timeline.py inflate every full barrier).
Usage (GPU box): python tools/phase_profile.py [n_windows=10000]"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ['DEEPBINNER_TIMELINE_PROFILE'] = '1'
from deepbinner_amd import hip_backend                      # noqa: E402
from deepbinner_amd.model_format import ModelWeights        # noqa: E402

NAMES = ['between windows', 'stage A', 'stage B', 'stage C', 'stage D (pair)', 'stages E+F',
         'between passes + batched tail', '-']


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    w, _ = ModelWeights.load(os.path.join(REPO, 'deepbinner_amd', 'models',
                                          'EXP-NBD103_read_starts.dbw'))
    model = hip_backend.HipModel(w, device=0)
    rng = np.random.default_rng(0)
    x = np.clip(np.rint(rng.standard_normal((n, 1024)) * 60 + 450), 0, 2047).astype(np.int16)
    model.set_read_length_hint(1024, x.size)
    model.timeline_i16(x)
    st = model.timeline_i16(x)[:256].astype(np.float64)         # [workgroup, wave, counter]
    per_wg = st[:, :, :8].mean(axis=1)                           # mean over the waves
    windows = np.array([len(range(b, n, 256)) for b in range(256)], dtype=np.float64)
    total = per_wg.sum(axis=1)
    print('windows per workgroup %d..%d; cycles per window (mean over workgroups): %.0f' % (
        windows.min(), windows.max(), (total / windows).mean()))
    out = {}
    for k, name in enumerate(NAMES[:7]):
        per_window = (per_wg[:, k] / windows).mean()
        out[name] = per_window
        print('%-32s %9.0f cycles per window  %5.1f%%' % (name, per_window,
                                                         100 * per_window / (total / windows).mean()))
    print(json.dumps(out))


if __name__ == '__main__':
    main()

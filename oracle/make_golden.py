#!/usr/bin/env python3
"""
ORACLE tooling — generates the committed fixtures under ``tests/golden/``.  Runs ONLY in the
build container, where /root/reference is mounted; nothing here travels to the GPU box except the
data files it writes.

What it does
1. Reads the reference's own test inputs (``tests/fast5_files/*.fast5``,
   ``tests/multi_read_fast5_files/*.fast5``) and copies them — data files, not source — to
   ``tests/golden/fast5/`` so the loader and end-to-end tests run anywhere.
2. Imports the reference's ``deepbinner/classify.py`` and ``deepbinner/trim_signal.py`` FROM
   /root/reference (with empty stand-in modules registered for h5py / keras / tensorflow, which
   ``call_batch``, ``make_sum_to_one``, ``get_barcode_call_from_probabilities``, ``combine_calls``,
   ``normalise`` and ``find_signal_start_pos`` never touch) and runs those reference functions
   themselves, around the oracle's ``predict``, to produce:
     * normalised windows for every (read, step, side)         -> windows_*.npy
     * merged probabilities + calls per model/side             -> calls.json / merged_*.npy
     * find_signal_start_pos per read                          -> calls.json["trim_start"]
3. Checks the results against every assertion the reference's tests make on this path
   (``tests/test_classify.py:115-121,134-140,154-160,174-180``; verbose rows ``:213-217``;
   ``tests/test_load_fast5s.py:46-72``) and aborts if any differs.
4. Cross-checks the NumPy graph against an independent torch-CPU evaluation (different conv /
   pooling implementation) and records the maximum difference.
5. Stores per-stage fp64 activations for a few windows for per-stage GPU parity tests.
"""

import io
import json
import os
import shutil
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
GOLD = os.path.join(REPO, 'tests', 'golden')
sys.path.insert(0, REPO)

from deepbinner_amd import hdf5_lite                      # noqa: E402
from deepbinner_amd.model_format import ModelWeights      # noqa: E402
from oracle import network_ref, classify_ref              # noqa: E402

MODELS = ['EXP-NBD103_read_starts', 'EXP-NBD103_read_ends', 'SQK-RBK004_read_starts']

# tests/test_classify.py:115-121 / 134-140 (require_either == start column, require_both == end)
EXPECTED_START = {'63c20e8e-9b10-4ede-9862-9a53eec3c512': '1',
                  '618f68a6-3a9a-45e1-afe0-845172b20349': '1',
                  '9bfcf22c-5654-4b4c-b8f7-d3cebd416338': '2',
                  '5ce8d6ab-8c24-43cc-808b-50fb336fda2f': '2',
                  '424bfd6b-576c-4e2c-bf86-604c771b5ec9': '3',
                  '177c3867-6812-4476-a6da-9e4d5c43b760': '3',
                  '2fbd86a4-029a-45cf-8f18-411d542572ba': '12'}
EXPECTED_END = dict(EXPECTED_START, **{'618f68a6-3a9a-45e1-afe0-845172b20349': 'none',
                                       '9bfcf22c-5654-4b4c-b8f7-d3cebd416338': 'none'})


def import_reference():
    for name in ('h5py', 'keras', 'keras.models', 'keras.backend', 'tensorflow'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['keras.models'].load_model = None
    sys.modules['keras'].backend = sys.modules['keras.backend']
    sys.path.insert(0, REF)
    import deepbinner.classify as ref_classify
    import deepbinner.trim_signal as ref_trim
    return ref_classify, ref_trim


def read_fast5(path):
    with hdf5_lite.File(path) as hf:
        if 'Raw' in hf.keys():
            group = list(hf['Raw/Reads/'].values())[0]
        else:
            name = [k for k in hf.keys() if k.startswith('read_')][0]
            group = hf[name + '/Raw/']
        return group.attrs['read_id'].decode(), group['Signal'][:]


class OracleModel:
    """Stands in for the Keras model at seam b1 (classify.py:361)."""

    def __init__(self, weights, dtype=np.float32):
        self.weights = weights
        self.dtype = dtype
        self.calls = []

    def predict(self, x, batch_size=None):
        x = np.asarray(x)
        self.calls.append(x[:, :, 0].copy())
        return network_ref.forward(self.weights, x.astype(np.float32),
                                   dtype=self.dtype).astype(np.float32)


def torch_forward(weights, x):
    """Independent evaluation with torch CPU operators (float64)."""
    import torch
    import torch.nn.functional as F
    from deepbinner_amd.model_format import conv_shapes, BN_EPSILON
    shapes = conv_shapes(weights.n_classes)
    t = torch.from_numpy(np.asarray(x, dtype=np.float64))[:, None, :]   # N, C, L

    def conv(i, t):
        kernel, bias = weights.convs[i - 1]
        _, k, _, _, stride, padding = shapes[i - 1]
        w = torch.from_numpy(kernel.astype(np.float64)).permute(2, 1, 0).contiguous()
        b = torch.from_numpy(bias.astype(np.float64))
        if padding == 'same':
            length = t.shape[2]
            out = -(-length // stride)
            total = max((out - 1) * stride + k - length, 0)
            t = F.pad(t, (total // 2, total - total // 2))
        return F.relu(F.conv1d(t, w, b, stride=stride))

    def bn(i, t):
        g, b, m, v = (torch.from_numpy(a.astype(np.float64)) for a in weights.bns[i - 1])
        return F.batch_norm(t, m, v, g, b, training=False, eps=BN_EPSILON)

    def avg(t):
        return F.avg_pool1d(t, 3, stride=1, padding=1, count_include_pad=False)

    t = bn(1, conv(1, t))
    t = bn(2, F.max_pool1d(conv(4, conv(3, conv(2, t))), 2))
    t = bn(3, F.max_pool1d(conv(7, conv(6, conv(5, t))), 2))
    t = bn(4, F.max_pool1d(conv(9, conv(8, t)), 2))
    t = torch.cat([conv(10, avg(t)), conv(11, t), conv(13, conv(12, t)),
                   conv(16, conv(15, conv(14, t)))], dim=1)
    t = bn(5, F.max_pool1d(t, 2))
    t = bn(6, conv(17, t))
    t = bn(7, F.max_pool1d(conv(19, conv(18, t)), 2))
    t = conv(20, t)
    return torch.softmax(t.mean(dim=2), dim=1).numpy()


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref_classify, ref_trim = import_reference()
    report = {}

    # ---- 1. inputs -------------------------------------------------------------------------
    single_dir = os.path.join(REF, 'tests', 'fast5_files')
    multi_dir = os.path.join(REF, 'tests', 'multi_read_fast5_files')
    for src, dst in ((single_dir, 'fast5/single'), (multi_dir, 'fast5/multi')):
        os.makedirs(os.path.join(GOLD, dst), exist_ok=True)
        for name in sorted(os.listdir(src)):
            if name.endswith('.fast5'):
                shutil.copyfile(os.path.join(src, name), os.path.join(GOLD, dst, name))
                os.chmod(os.path.join(GOLD, dst, name), 0o644)

    files = sorted(f for f in os.listdir(single_dir) if f.endswith('.fast5'))
    read_ids, signals = [], []
    for name in files:
        rid, sig = read_fast5(os.path.join(single_dir, name))
        read_ids.append(rid)
        signals.append(sig)
    # tests/test_load_fast5s.py:46-49,55-58,69-72
    by_id = dict(zip(read_ids, signals))
    s = by_id['177c3867-6812-4476-a6da-9e4d5c43b760']
    assert (len(s), s[0], s[4950]) == (4971, 714, 396)
    s = by_id['9bfcf22c-5654-4b4c-b8f7-d3cebd416338']
    assert (len(s), s[0], s[4862]) == (4983, 493, 618)
    s = by_id['2fbd86a4-029a-45cf-8f18-411d542572ba']
    assert (len(s), s[0], s[5388]) == (5395, 505, 436)

    # multi-read files: every read of the three files (30 reads) for wider coverage
    multi_ids, multi_signals = [], []
    for name in sorted(os.listdir(multi_dir)):
        with hdf5_lite.File(os.path.join(multi_dir, name)) as hf:
            for key in hf.keys():
                g = hf[key + '/Raw']
                multi_ids.append(g.attrs['read_id'].decode())
                multi_signals.append(g['Signal'][:])

    def pack(sigs):
        offsets = np.zeros(len(sigs) + 1, dtype=np.int64)
        offsets[1:] = np.cumsum([len(x) for x in sigs])
        return np.concatenate(sigs).astype(np.int16), offsets

    samples, offsets = pack(signals)
    msamples, moffsets = pack(multi_signals)
    np.savez_compressed(os.path.join(GOLD, 'reads.npz'), files=np.array(files),
                        read_ids=np.array(read_ids), samples=samples, offsets=offsets,
                        multi_read_ids=np.array(multi_ids), multi_samples=msamples,
                        multi_offsets=moffsets)

    # ---- 2+3. reference call_batch around the oracle predict -------------------------------
    import argparse
    args = argparse.Namespace(scan_size=6144, batch_size=128, score_diff=0.5,
                              require_either=False, require_start=False, require_both=False)
    weights = {m: ModelWeights.load_keras_hdf5(os.path.join(REF, 'models', m))[0] for m in MODELS}
    calls_out = {'read_ids': read_ids, 'files': files, 'multi_read_ids': multi_ids}
    all_ids = read_ids + multi_ids
    all_signals = signals + multi_signals
    plan = [('EXP-NBD103_read_starts', 'start'), ('EXP-NBD103_read_ends', 'end'),
            ('SQK-RBK004_read_starts', 'start')]
    for model_name, side in plan:
        model = OracleModel(weights[model_name])
        ref_calls, ref_probs = ref_classify.call_batch(1024, 13, all_ids, all_signals, model,
                                                       args, side)
        ref_probs = np.array([[float(v) for v in row] for row in ref_probs])
        windows = np.stack(model.calls)                       # [12, n_reads, 1024] float64
        # the oracle's own restatement of windowing + merge must agree with the reference code
        mine = classify_ref.make_windows(all_signals, 1024, 6144, side)
        assert np.array_equal(mine, windows), 'window restatement differs from reference'
        o_calls, o_probs = classify_ref.call_batch(
            lambda w: network_ref.forward(weights[model_name], w.astype(np.float32),
                                          dtype=np.float32),
            all_signals, 1024, 6144, 0.5, side)
        assert o_calls == ref_calls
        assert np.abs(o_probs - ref_probs).max() < 1e-6
        key = '%s/%s' % (model_name, side)
        calls_out[key] = ref_calls
        np.save(os.path.join(GOLD, 'merged_%s_%s.npy' % (model_name, side)), ref_probs)
        np.save(os.path.join(GOLD, 'windows_%s.npy' % side), windows.astype(np.float32))
        # per-window probabilities (pre-merge), fp64 graph
        flat = windows.reshape(-1, 1024).astype(np.float32)
        p64 = network_ref.forward(weights[model_name], flat, dtype=np.float64)
        p32 = network_ref.forward(weights[model_name], flat, dtype=np.float32)
        np.save(os.path.join(GOLD, 'window_probs_%s_%s.npy' % (model_name, side)), p64)
        report['fp32_vs_fp64_%s' % key] = float(np.abs(p64 - p32).max())
        pt = torch_forward(weights[model_name], flat)
        report['numpy_vs_torch_%s' % key] = float(np.abs(p64 - pt).max())
        assert report['numpy_vs_torch_%s' % key] < 1e-9

    # reference test assertions (28 calls)
    n = len(read_ids)
    start = dict(zip(read_ids, calls_out['EXP-NBD103_read_starts/start'][:n]))
    end = dict(zip(read_ids, calls_out['EXP-NBD103_read_ends/end'][:n]))
    assert start == EXPECTED_START, start
    assert end == EXPECTED_END, end
    for mode, want in (('require_either', EXPECTED_START), ('require_both', EXPECTED_END)):
        a = argparse.Namespace(require_either=mode == 'require_either', require_start=False,
                               require_both=mode == 'require_both')
        got = {r: ref_classify.combine_calls(start[r], end[r], a) for r in read_ids}
        assert got == want, (mode, got)
        assert got == {r: classify_ref.combine_calls(start[r], end[r], mode) for r in read_ids}
    # verbose row, tests/test_classify.py:213-217 and :249-253
    i = read_ids.index('177c3867-6812-4476-a6da-9e4d5c43b760')
    for name in ('merged_EXP-NBD103_read_starts_start.npy', 'merged_EXP-NBD103_read_ends_end.npy'):
        row = np.load(os.path.join(GOLD, name))[i]
        assert ['%.2f' % v for v in row] == ['0.00'] * 3 + ['1.00'] + ['0.00'] * 9

    # trim_signal
    calls_out['trim_start'] = [int(ref_trim.find_signal_start_pos(s)) for s in signals]
    assert calls_out['trim_start'] == [int(classify_ref.find_signal_start_pos(s)) for s in signals]
    trims = []
    for sgl in multi_signals:
        try:
            trims.append(int(ref_trim.find_signal_start_pos(sgl)))
        except ref_trim.CannotTrim:
            trims.append(-1)
    calls_out['multi_trim_start'] = trims
    # normalise edge cases straight from the reference function
    edge = {'empty': ref_trim.normalise(np.array([], dtype=np.int16)).tolist(),
            'flat': ref_trim.normalise(np.array([7, 7, 7], dtype=np.int16)).tolist(),
            'ramp': ref_trim.normalise(np.array([1, 2, 3, 4], dtype=np.int16)).tolist()}
    calls_out['normalise_edge'] = edge

    # combine_calls truth table (tests/test_combine_calls.py:29-51), from the reference function
    table = {}
    for mode in ('require_either', 'require_start', 'require_both'):
        a = argparse.Namespace(require_either=mode == 'require_either',
                               require_start=mode == 'require_start',
                               require_both=mode == 'require_both')
        for s_, e_ in (('4', '4'), ('none', 'none'), ('5', 'none'), ('none', '7'), ('1', '2')):
            table['%s|%s|%s' % (mode, s_, e_)] = ref_classify.combine_calls(s_, e_, a)
    calls_out['combine_table'] = table

    # header strings (tests/test_classify.py:195,213,287-292) from the reference function
    headers = {}
    for verbose, st, en in ((False, True, False), (True, True, False), (True, False, True),
                            (True, True, True)):
        buf = io.StringIO()
        old = sys.stdout
        sys.stdout = buf
        try:
            ref_classify.print_output_header(verbose, st, en, 13)
        finally:
            sys.stdout = old
        headers['%d%d%d' % (verbose, st, en)] = buf.getvalue()
    calls_out['headers'] = headers

    # ---- 5. per-stage activations for a few windows (NBD103 starts) ------------------------
    wstart = np.load(os.path.join(GOLD, 'windows_start.npy'))
    # pick informative windows: step 0 of four real reads, a short/padded one, an all-zero one
    picks = [(0, 0), (0, 2), (0, 3), (0, 6), (9, 0), (11, 1)]
    sel = np.stack([wstart[s, i] for s, i in picks]).astype(np.float32)
    zero_and_synth = np.zeros((2, 1024), dtype=np.float32)
    rng = np.random.default_rng(20260927)
    zero_and_synth[1] = rng.standard_normal(1024).astype(np.float32)
    sel = np.concatenate([sel, zero_and_synth])
    _, stages = network_ref.forward(weights['EXP-NBD103_read_starts'], sel, dtype=np.float64,
                                    return_stages=True)
    np.savez_compressed(os.path.join(GOLD, 'stages_EXP-NBD103_read_starts.npz'), x=sel,
                        **{k: v.astype(np.float32 if k not in ('H', 'logits') else np.float64)
                           for k, v in stages.items()})
    zero_probs = {}
    for m in MODELS:
        p = network_ref.forward(weights[m], np.zeros((1, 1024), np.float32), dtype=np.float64)
        zero_probs[m] = float(p[0, 0])
    calls_out['zero_window_class0'] = zero_probs
    calls_out['report'] = report
    with open(os.path.join(GOLD, 'calls.json'), 'w') as f:
        json.dump(calls_out, f, indent=1, sort_keys=True)
    print(json.dumps(report, indent=1))
    print('zero-window class-0 probabilities', zero_probs)
    print('trim', calls_out['trim_start'])


if __name__ == '__main__':
    main()

"""Semi-global DTW: the oracle against the reference's own answers (CPU), and the HIP kernel
against the oracle through the C ABI (GPU).  Integer outputs (positions, alignment) and the fp64
distance are compared bit for bit - the kernel does the reference's operations in its order."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLD, REPO
from deepbinner_amd import dtw_semi_global as dtw
from oracle import dtw_ref


def golden_cases():
    data = np.load(os.path.join(GOLD, 'dtw.npz'))
    for k in range(int(data['n'])):
        distance, start, end = data['answer_%d' % k]
        yield (data['ref_%d' % k], data['query_%d' % k], float(distance), int(start), int(end),
               data['pairs_%d' % k])


def squiggle(rng, n_levels, dwell=8):
    levels = rng.normal(0.0, 1.0, size=n_levels)
    return np.repeat(levels, rng.integers(max(1, dwell - 3), dwell + 4, size=n_levels))


def path_cost(ref, query, pairs):
    """Accumulated cost along an alignment by the reference's rules: column 0 is free."""
    return sum((ref[i] - query[j]) ** 2 for i, j in pairs if j > 0)


def check_path(ref, query, start, end, pairs):
    pairs = [tuple(int(v) for v in p) for p in pairs]
    assert pairs[0] == (start, 0) and pairs[-1] == (end, len(query) - 1)
    for (i0, j0), (i1, j1) in zip(pairs, pairs[1:]):
        assert (i1 - i0, j1 - j0) in ((1, 1), (0, 1), (1, 0))


# ---------------------------------------------------------------------------------- CPU
def test_restatement_reproduces_the_reference_answers():
    """tests/golden/dtw.npz holds what the compiled dtw.cpp returned (oracle/make_dtw_golden.py)."""
    n = 0
    for ref, query, distance, start, end, pairs in golden_cases():
        got = dtw_ref.semi_global_dtw(ref, query, 'restatement')
        assert got[0] == distance and got[1:3] == (start, end)
        assert np.array_equal(np.array(got[3], dtype=np.int32).reshape(-1, 2), pairs)
        n += 1
    assert n >= 15


@pytest.mark.skipif(not dtw_ref.available('reference'), reason='oracle/_ref/dtw.so not built')
def test_restatement_against_the_compiled_reference():
    rng = np.random.default_rng(5)
    for trial in range(150):
        ref = rng.normal(size=int(rng.integers(1, 400)))
        query = rng.normal(size=int(rng.integers(1, 120)))
        assert dtw_ref.semi_global_dtw(ref, query, 'restatement') == \
            dtw_ref.semi_global_dtw(ref, query, 'reference'), trial
    # integer-valued signals tie all the time: the reference draws rand() between LEFT and UP
    # there, so only the distance (which cannot depend on it) and the path's validity compare
    for trial in range(50):
        ref = rng.integers(0, 4, size=int(rng.integers(2, 200))).astype(np.float64)
        query = rng.integers(0, 4, size=int(rng.integers(1, 60))).astype(np.float64)
        ours = dtw_ref.semi_global_dtw(ref, query, 'restatement')
        theirs = dtw_ref.semi_global_dtw(ref, query, 'reference')
        assert ours[0] == theirs[0]
        for distance, start, end, pairs in (ours, theirs):
            check_path(ref, query, start, end, pairs)
            assert path_cost(ref, query, pairs) == distance        # small integers: exact


@pytest.mark.skipif(not dtw.available(), reason='libdeepbinner_dtw.so not built')
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, 'include', 'deepbinner_dtw.h')).read()
    declared = set(re.findall(r'\b(dtw_[a-z_0-9]+|semi_global_dtw)\s*\(', header))
    assert declared == set(dtw.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(dtw.library_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert dtw.load_library().dtw_version().startswith(b'deepbinner_dtw')


@pytest.mark.skipif(not dtw.available(), reason='libdeepbinner_dtw.so not built')
def test_no_device_is_an_error_not_a_fallback():
    from deepbinner_amd import hip_backend
    try:
        n_devices = hip_backend.device_count()
    except Exception:
        n_devices = 0
    if n_devices > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(RuntimeError):
        dtw.semi_global_dtw(np.zeros(10), np.zeros(4))
    with pytest.raises(RuntimeError):
        dtw.semi_global_dtw_batch([np.zeros(10)], [np.zeros(4)])


# ---------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_golden_cases_one_by_one_and_as_a_batch():
    cases = list(golden_cases())
    for ref, query, distance, start, end, pairs in cases:
        got = dtw.semi_global_dtw(ref, query)
        assert got[0] == distance and (got[1], got[2]) == (start, end)
        assert np.array_equal(np.array(got[3], dtype=np.int32).reshape(-1, 2), pairs)
    batch = dtw.semi_global_dtw_batch([c[0] for c in cases], [c[1] for c in cases])
    for (ref, query, distance, start, end, pairs), got in zip(cases, batch):
        assert got[0] == distance and got[1:3] == (start, end)
        assert np.array_equal(got[3], pairs)
    without = dtw.semi_global_dtw_batch([c[0] for c in cases], [c[1] for c in cases],
                                        alignments=False)
    assert [w[:3] for w in without] == [b[:3] for b in batch]


@pytest.mark.gpu
def test_gpu_random_batch_matches_oracle():
    """300 pairs of assorted shapes in one call: every lane width (4/8/16 columns per lane),
    multi-panel queries (> 1,024 samples), one-sample signals, references shorter than a wave."""
    rng = np.random.default_rng(11)
    refs, queries = [], []
    for trial in range(300):
        q = int(rng.choice([1, 2, 3, 63, 64, 65, 200, 256, 257, 500, 512, 513, 1000, 1024, 1025,
                            2047, 2048, 2049, 4100])) if trial % 3 == 0 \
            else int(rng.integers(1, 1400))
        r = int(rng.integers(1, 60)) if trial % 7 == 0 else int(rng.integers(1, 3000))
        if trial % 5 == 0:
            refs.append(squiggle(rng, r // 8 + 1)[:r])
            queries.append(squiggle(rng, q // 8 + 1)[:q])
        else:
            refs.append(rng.normal(size=r))
            queries.append(rng.normal(size=q))
    got = dtw.semi_global_dtw_batch(refs, queries)
    kind = 'reference' if dtw_ref.available('reference') else 'restatement'
    for k, (ref, query) in enumerate(zip(refs, queries)):
        want = dtw_ref.semi_global_dtw(ref, query, 'restatement')
        assert got[k][0] == want[0] and got[k][1:3] == want[1:3], (k, len(ref), len(query))
        assert np.array_equal(got[k][3], np.array(want[3], dtype=np.int32).reshape(-1, 2)), k
        if kind == 'reference' and k % 10 == 0:
            assert dtw_ref.semi_global_dtw(ref, query, 'reference') == want


@pytest.mark.gpu
def test_gpu_ties_give_the_same_distance_and_a_valid_path():
    rng = np.random.default_rng(12)
    refs = [rng.integers(0, 4, size=int(rng.integers(2, 600))).astype(np.float64)
            for _ in range(60)]
    queries = [rng.integers(0, 4, size=int(rng.integers(1, 300))).astype(np.float64)
               for _ in range(60)]
    for ref, query, (distance, start, end, pairs) in zip(refs, queries,
                                                        dtw.semi_global_dtw_batch(refs, queries)):
        want = dtw_ref.semi_global_dtw(ref, query, 'restatement')
        assert (distance, start, end) == want[:3]          # same tie rule as the restatement
        assert np.array_equal(pairs, np.array(want[3], dtype=np.int32).reshape(-1, 2))
        check_path(ref, query, start, end, pairs)
        assert path_cost(ref, query, pairs) == distance


@pytest.mark.gpu
def test_gpu_properties_at_sizes_the_oracle_does_not_reach():
    """A query cut out of a long reference is found where it was cut, at distance 0, on the
    diagonal; a shifted copy of the reference region costs len * shift^2 at most."""
    rng = np.random.default_rng(13)
    refs, queries, where = [], [], []
    for _ in range(64):
        ref = rng.normal(size=int(rng.integers(20000, 40000)))
        n = int(rng.integers(500, 5000))
        at = int(rng.integers(1, len(ref) - n))
        refs.append(ref)
        queries.append(ref[at:at + n].copy())
        where.append((at, n))
    for (distance, start, end, pairs), (at, n) in zip(dtw.semi_global_dtw_batch(refs, queries),
                                                      where):
        assert distance == 0.0 and end == at + n - 1
        # column 0 is free, so the path may enter anywhere on it; from column 1 on it is the diagonal
        assert np.array_equal(pairs[1:, 0] - pairs[1:, 1], np.full(len(pairs) - 1, at))
    shifted = dtw.semi_global_dtw_batch(refs[:8], [q + 0.25 for q in queries[:8]],
                                        alignments=False)
    for (distance, _, _, _), (at, n) in zip(shifted, where):
        assert 0.0 < distance <= n * 0.0625 + 1e-9


@pytest.mark.gpu
def test_gpu_small_path_budget_splits_the_batch(monkeypatch):
    rng = np.random.default_rng(14)
    refs = [rng.normal(size=int(rng.integers(100, 900))) for _ in range(40)]
    queries = [rng.normal(size=int(rng.integers(10, 700))) for _ in range(40)]
    whole = dtw.semi_global_dtw_batch(refs, queries)
    monkeypatch.setenv('DEEPBINNER_DTW_PATH_BYTES', str(300 * 1024))
    split = dtw.semi_global_dtw_batch(refs, queries)
    for a, b in zip(whole, split):
        assert a[:3] == b[:3] and np.array_equal(a[3], b[3])


@pytest.mark.gpu
def test_gpu_rescaling_as_the_reference_does_it():
    """semi_global_dtw_with_rescaling (dtw_semi_global.py:62-95) replayed with the oracle's DTW:
    align, least-squares fit of the query onto the reference over the aligned pairs, align again."""
    rng = np.random.default_rng(15)
    for trial in range(6):
        query = squiggle(rng, 40)
        ref = np.concatenate([squiggle(rng, 50), query + rng.normal(0, 0.05, len(query)),
                              squiggle(rng, 30)])
        scaled = (0.85 + 0.05 * trial) * query + 0.3
        distance, start, end, values = dtw.semi_global_dtw_with_rescaling(ref, scaled)

        q = np.array(scaled)
        d1, _, _, pairs = dtw_ref.semi_global_dtw(ref, q)
        x = [q[j] for _, j in pairs]
        y = [ref[i] for i, _ in pairs]
        m, b = np.linalg.lstsq(np.vstack([x, np.ones(len(x))]).T, y, rcond=None)[0]
        q = m * q + b
        d2, s2, e2, pairs = dtw_ref.semi_global_dtw(ref, q)
        if m < 0.75 or m > 1.333:
            d2 = float('inf')
        assert (distance, start, end) == (d2, s2, e2)
        assert values == [(ref[i], q[j]) for i, j in pairs]
        assert d2 < d1
    # a slope far from 1 is flagged
    assert dtw.semi_global_dtw_with_rescaling(ref, 3.0 * query)[0] == float('inf')
    many = dtw.semi_global_dtw_with_rescaling_batch([ref, ref], [scaled, 3.0 * query])
    assert many[0][:3] == (distance, start, end) and many[1][0] == float('inf')


@pytest.mark.gpu
def test_gpu_bad_arguments():
    with pytest.raises(RuntimeError):
        dtw.semi_global_dtw(np.zeros(0), np.zeros(4))
    with pytest.raises(RuntimeError):
        dtw.semi_global_dtw_batch([np.zeros(5), np.zeros(0)], [np.zeros(4), np.zeros(4)])
    with pytest.raises(ValueError):
        dtw.semi_global_dtw_batch([np.zeros(5)], [])
    assert dtw.semi_global_dtw_batch([], []) == []


@pytest.mark.gpu
@pytest.mark.skipif(os.environ.get('DEEPBINNER_SOAK') != '1',
                    reason='opt-in: DEEPBINNER_SOAK=1 (about two minutes of CPU for the reference)')
def test_gpu_rate_and_soak_beside_the_reference(capsys):
    """6,000 random pairs of four signal kinds in one batched call, every distance, position and
    path compared with the reference's own code (oracle/_ref/dtw.so when it travelled with the
    repository, else the restatement); then the kernel rate at four query lengths with the CPU
    code timed beside it on one core.  Prints JSON lines (kept in profiles/r01_dtw/)."""
    import json
    import time
    kind = 'reference' if dtw_ref.available('reference') else 'restatement'
    rng = np.random.default_rng(7)
    refs, queries = [], []
    for k in range(6000):
        r = int(rng.integers(1, 4000)) if k % 11 else int(rng.integers(1, 70))
        q = int(rng.integers(1, 1500)) if k % 13 else int(rng.integers(1000, 3500))
        style = k % 4
        if style == 0:
            refs.append(rng.normal(size=r))
            queries.append(rng.normal(size=q))
        elif style == 1:      # squiggle-like: levels held for a few samples, small noise
            refs.append(np.repeat(rng.normal(size=r // 6 + 1), 6)[:r] + rng.normal(0, .05, r))
            queries.append(np.repeat(rng.normal(size=q // 6 + 1), 6)[:q] + rng.normal(0, .05, q))
        elif style == 2:      # the query is a noisy, rescaled piece of the reference
            ref = rng.normal(size=max(r, 2))
            a = int(rng.integers(0, len(ref) - 1))
            piece = ref[a:int(rng.integers(a + 1, len(ref) + 1))][:q]
            refs.append(ref)
            queries.append(1.1 * piece + 0.1 + rng.normal(0, .1, len(piece)))
        else:                 # large offsets and scales
            refs.append(rng.normal(400, 90, size=r))
            queries.append(rng.normal(450, 60, size=q))
    t0 = time.perf_counter()
    got = dtw.semi_global_dtw_batch(refs, queries)
    gpu_seconds = time.perf_counter() - t0
    t0 = time.perf_counter()
    cells = 0
    for k, (ref, query) in enumerate(zip(refs, queries)):
        want = dtw_ref.semi_global_dtw(ref, query, kind)
        cells += len(ref) * len(query)
        assert got[k][0] == want[0] and got[k][1:3] == want[1:3], (k, len(ref), len(query))
        assert np.array_equal(got[k][3], np.array(want[3], dtype=np.int32).reshape(-1, 2)), k
    lines = [{'soak_pairs': 6000, 'cells': cells, 'checked_against': kind, 'identical': True,
              'gpu_call_seconds': round(gpu_seconds, 3),
              'cpu_seconds': round(time.perf_counter() - t0, 1)}]
    for q in (200, 500, 1000, 2000):
        refs = [rng.normal(size=4000) for _ in range(4096)]
        queries = [rng.normal(size=q) for _ in range(4096)]
        dtw.semi_global_dtw_batch(refs[:64], queries[:64])
        results = dtw.semi_global_dtw_batch(refs, queries)
        ms, n_cells = dtw.last_kernel_time()
        t0 = time.perf_counter()
        for k in range(12):
            want = dtw_ref.semi_global_dtw(refs[k], queries[k], kind)
            assert results[k][0] == want[0] and results[k][1:3] == want[1:3]
        cpu = 12 * 4000 * q / (time.perf_counter() - t0) / 1e9
        lines.append({'query_len': q, 'pairs': 4096, 'kernel_ms': ms,
                      'kernel_GCUPS': n_cells / (ms * 1e-3) / 1e9,
                      'cpu': {'kind': kind, 'cores': 1, 'GCUPS': cpu}})
    with capsys.disabled():
        for line in lines:
            print(json.dumps(line))

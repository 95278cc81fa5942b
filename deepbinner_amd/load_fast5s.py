"""
fast5 discovery and signal loading — mirror of the reference's ``deepbinner/load_fast5s.py``
with h5py replaced by this package's own reader (``hdf5_lite``).

Beyond the reference: ``iter_reads`` walks every read of a multi-read fast5 directly, which is
what lets ``realtime`` skip the reference's ``multi_to_single_fast5`` subprocess
(reference ``realtime.py:183-190``).
"""

import os
import random
import sys

from . import hdf5_lite


def _read_group(hdf5_file):
    """The group holding ``read_id`` and ``Signal`` for a one-read file, or None
    (reference load_fast5s.py:29-43)."""
    keys = list(hdf5_file.keys())
    if 'Raw' in keys:   # older format: exactly one read under /Raw/Reads
        return list(hdf5_file['Raw/Reads/'].values())[0]
    reads = [k for k in keys if k.startswith('read_')]
    if len(reads) > 1:
        sys.exit('Error: Deepbinner does not (yet) support multi-read fast5 files')
    if not reads:
        return None
    return hdf5_file[reads[0] + '/Raw/']


def reader_kind():
    """Which fast5 reader serves this process: 'native' (libdeepbinner_fast5.so, C++) or 'python'
    (hdf5_lite).  DEEPBINNER_FAST5_READER=native|python forces one; the default takes the native
    library when it has been built.  Both implement the same slice of HDF5 and are tested against
    each other (tests/test_fast5_native.py)."""
    want = os.environ.get('DEEPBINNER_FAST5_READER', 'auto')
    if want == 'python':
        return 'python'
    from . import fast5_native
    if want == 'native':
        fast5_native.load_library()         # fail loudly if it was asked for and is not there
        return 'native'
    return 'native' if fast5_native.available() else 'python'


def get_read_id_and_signal(fast5_file):
    """-> (read_id str, int16 ndarray); (None, None) for unreadable files
    (reference load_fast5s.py:25-49)."""
    if reader_kind() == 'native':
        from . import fast5_native
        return fast5_native.get_read_id_and_signal(fast5_file)
    return _python_get_read_id_and_signal(fast5_file)


def _python_get_read_id_and_signal(fast5_file):
    try:
        with hdf5_lite.File(str(fast5_file), 'r') as hdf5_file:
            group = _read_group(hdf5_file)
            if group is None:
                return None, None
            read_id = group.attrs['read_id'].decode()
            signal = group['Signal'][:]
        return read_id, signal
    except Exception:      # h5py: OSError / KeyError; a damaged file can trip anything else
        return None, None


def iter_reads(fast5_file):
    """Yield (read_id, signal) for every read of a single- or multi-read fast5."""
    if reader_kind() == 'native':
        from . import fast5_native
        yield from fast5_native.iter_reads(fast5_file)
        return
    yield from _python_iter_reads(fast5_file)


def _python_iter_reads(fast5_file):
    try:
        with hdf5_lite.File(str(fast5_file), 'r') as hdf5_file:
            keys = list(hdf5_file.keys())
            if 'Raw' in keys:
                groups = [list(hdf5_file['Raw/Reads/'].values())[0]]
            else:
                groups = [hdf5_file[k + '/Raw/'] for k in keys if k.startswith('read_')]
            for group in groups:
                yield group.attrs['read_id'].decode(), group['Signal'][:]
    except Exception:      # h5py: OSError / KeyError; a damaged file can trip anything else
        return


def keep_ends(signal, keep):
    """``signal`` if it is at most 2*keep samples long, else its first and last ``keep`` samples
    joined.  Windows are cut from the first (start model) or last (end model) scan_size + half a
    window samples only (reference classify.py:337-349), so with keep >= that the joined array
    classifies exactly like the whole read."""
    if keep is None or signal is None or len(signal) <= 2 * keep:
        return signal
    import numpy as np
    return np.concatenate((signal[:keep], signal[-keep:]))


def _load_files(paths, keep=None):
    """Worker of LoaderPool: [(read_id, signal)] for a run of one-read files."""
    out = []
    for path in paths:
        read_id, signal = get_read_id_and_signal(path)
        out.append((read_id, keep_ends(signal, keep)))
    return out


class LoaderPool:
    """Ordered, prefetching pool of loader processes for one-read fast5 files.

    The reference loads its files one by one on the thread that also drives the model
    (classify.py:141-150); with the network on the GPU that loop is the bottleneck (~1 ms per
    file here against ~3 us per read on the device), so the files of the coming batches are
    parsed and inflated by ``procs`` worker processes while the current batch is classified.
    Results come back in file order; at most ``ahead`` runs of ``run`` files are in flight, so
    memory stays bounded however far the loaders could get ahead.  Workers are spawned, not
    forked: the parent may already hold a HIP context."""

    def __init__(self, procs, run=8, ahead=None):
        import multiprocessing
        self.procs = int(procs)
        self.run = int(run)
        self.ahead = int(ahead) if ahead else 4 * self.procs
        self._pool = multiprocessing.get_context('spawn').Pool(self.procs)

    def load(self, fast5_files, keep=None):
        """Yield (fast5_file, read_id, signal) for every file, in order.  ``keep``: ship only the
        first and last ``keep`` samples of longer reads (see keep_ends) - the pipe back from the
        workers is what limits this pool."""
        from collections import deque
        runs = [fast5_files[i:i + self.run] for i in range(0, len(fast5_files), self.run)]
        pending, submitted = deque(), 0
        while pending or submitted < len(runs):
            while submitted < len(runs) and len(pending) < self.ahead:
                pending.append((runs[submitted],
                                self._pool.apply_async(_load_files, (runs[submitted], keep))))
                submitted += 1
            paths, job = pending.popleft()
            for path, (read_id, signal) in zip(paths, job.get()):
                yield path, read_id, signal

    def close(self):
        self._pool.terminate()
        self._pool.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def choose_loader_procs(requested, n_files):
    """Number of loader processes: the ``--loader_procs`` request, or by default up to 8 (more
    only queue behind the parent's end of the result pipe: tools/loader_rate.py) once a job is
    big enough to repay starting them (>= 512 files)."""
    if requested is not None and int(requested) > 0:
        return int(requested)
    if n_files < 512:
        return 1
    from .misc import usable_cpus
    return max(1, min(8, usable_cpus() // 2))


def find_all_fast5s(directory, verbose=False):
    if verbose:
        print('Looking for fast5 files in {}... '.format(directory), file=sys.stderr, end='',
              flush=True)
    fast5s = [os.path.join(root, name)
              for root, _, names in os.walk(str(directory))
              for name in names if name.endswith('.fast5')]
    if verbose:
        print('{} {} found'.format(len(fast5s), 'fast5' if len(fast5s) == 1 else 'fast5s'),
              file=sys.stderr)
    return fast5s


def determine_single_or_multi_fast5s(fast5s):
    """Inspect up to five randomly chosen files (reference load_fast5s.py:67-90)."""
    sample = list(fast5s)
    random.shuffle(sample)
    kinds = set()
    for fast5_file in sample[:5]:
        keys = get_root_level_keys(fast5_file)
        if 'Raw' in keys:
            kinds.add('single-old')
            continue
        read_count = sum(1 for k in keys if k.startswith('read_'))
        if read_count == 1:
            kinds.add('single-new')
        elif read_count > 1:
            kinds.add('multi')
    if 'multi' in kinds and 'single-old' in kinds:
        sys.exit('Error: your reads appear to be a mixture of old and new formats. Deepbinner '
                 'can handle one or the other, but not both at once.')
    return 'multi' if 'multi' in kinds else 'single'


def get_root_level_keys(fast5_file):
    try:
        with hdf5_lite.File(str(fast5_file), 'r') as hdf5_file:
            return list(hdf5_file.keys())
    except OSError:
        return []

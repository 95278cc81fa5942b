"""
``deepbinner realtime``: follow a directory while a sequencing run writes fast5 files into it and
sort them into ``barcodeNN/`` / ``unclassified/`` under the output directory.

What the user sees is the reference's ``deepbinner/realtime.py`` (:28-196): a 5 s polling loop,
at most 20,000 one-read files or 5 multi-read files per pass so that files start moving early,
the same directory names, progress lines and error texts, ``--stop`` to exit once the input
directory is drained.  How it is put together is this package's own: one ``Session`` object owns
the models and the bookkeeping, a ``MoveTally`` does the filing, and multi-read files are
classified straight from the container with this package's fast5 readers
(``load_fast5s.iter_reads``) - the reference shells out to ``multi_to_single_fast5`` and bins the
unpacked copies (:183-190), which is still done when that tool is installed.  A multi-read file
holds reads of many barcodes and cannot be moved into one bin; without the tool every read is
written out as a one-read fast5 of its own into its bin by this package's HDF5 writer
(``hdf5_write``: ``<out_dir>/barcodeNN/<read_id>.fast5``, whole signal, deflate level 1) and every
pass appends ``read_id<TAB>barcode<TAB>source_file`` rows to
``<out_dir>/multi_read_classifications.tsv`` (``DEEPBINNER_REALTIME_TABLE_ONLY=1``: only those).
"""

import os
import weakref
import pathlib
import shutil
import subprocess
import sys
import tempfile
import threading
import time

from . import classify
from .load_fast5s import determine_single_or_multi_fast5s, iter_reads, reader_kind
from .misc import print_summary_table, usable_cpus

POLL_SECONDS = 5
PER_PASS = {'single': 20000, 'multi': 5}       # reference realtime.py:86-94


# The streaming path with the GPU inflating: what it was measured with on an MI355X box (16
# usable cores, one GPU; profiles/r03_gpu_inflate_split.txt, profiles/r05_multi_read_rate.json) -
# three containers in flight per GPU, each on its own queue (model replica), whatever share of
# the inflating the GPU takes (two leave the GPU idle while a container is uploaded, four and more
# only add inflate kernels that compete with the one forward kernel the device runs at a time:
# dbh_api.hip, "THE FORWARD STREAM"; profiles/r05_k2/forward_stream_sweep_16_containers.txt).
# No CUs are left out of the forward kernel's launches for the inflate kernels: its workgroups take their windows off a counter, and
# one that finds its CU taken simply takes fewer (while they walked fixed shares, 32 were).
INFLATE_QUEUES = 3
INFLATE_CUS = 0
# ... unless the streams are LONG (a stream is one wave's sequential work in the inflate kernels: a
# container of 1,000 reads of ~100 k samples keeps a quarter of the wave slots 4,000 ordinary reads do,
# each for four times as long, and the forward kernel on every CU starves them): a container whose
# zlib streams average more than LONG_STREAM_BYTES leaves LONG_STREAM_CUS CUs out of its forward
# launches for the inflate kernels of the containers behind it - 139 k reads/s instead of 115 k on such
# containers, the host inflating nothing; ordinary containers (34 KB streams) are best at 0 (209 k
# against 199 k at 32: profiles/r06_loader/long_reads_cu_sweep.txt).  DEEPBINNER_INFLATE_CUS overrides.
HOST_ONLY_CORES_PER_GPU = 30
RAW_LOADER_THREADS_PER_GPU = 4
LONG_STREAM_BYTES = 64 * 1024
LONG_STREAM_CUS = 64


def inflate_cus_for(comp_bytes, modes):
    """CUs a container with these raw records (their ``comp_bytes`` and ``mode``) leaves to the
    inflate kernels, or None where DEEPBINNER_INFLATE_CUS has said it already."""
    if os.environ.get('DEEPBINNER_INFLATE_CUS') is not None:
        return None
    total = count = 0
    for size, mode in zip(comp_bytes, modes):
        if mode == 0:                   # fast5_native.RAW_ZLIB
            total += int(size)
            count += 1
    return LONG_STREAM_CUS if count and total / count > LONG_STREAM_BYTES else INFLATE_CUS


def host_inflate_share(n_gpus):
    """Per cent of a container's compressed bytes the host's threads inflate themselves (its
    longest streams); the GPUs inflate the rest.  100 = everything on the host (the CPU-only
    loader path), 0 = everything on the GPUs.  DEEPBINNER_GPU_INFLATE=0 / =1 force either end,
    DEEPBINNER_HOST_INFLATE_SHARE=<per cent> any split.

    Left alone: all or nothing.  Measured with round 6's kernels (profiles/r06_loader/
    long_reads_cu_sweep.txt and host_share_ordinary.txt; 27 k-sample reads, gzip 1, 16 loader
    threads, three containers in flight): a GPU alone settles at 209-210 k reads/s for 17 us of host
    CPU per read; with the host's threads taking 20 % of the bytes 205 k for 39 us, with 46 % (what
    round 5's rule gave this box, when the help was worth +5 %) 199-202 k for 66 us.  So the GPUs
    inflate every stream - unless the host has cores to spare (30 per GPU and more: the CPU-only
    loader path is as fast and the inflate kernels are not used at all).  A 16-core host in front
    of eight GPUs - BASELINE.json configs[4] - nothing either way."""
    flag = os.environ.get('DEEPBINNER_GPU_INFLATE')
    if flag == '0':
        return 100
    if flag == '1':
        return 0
    explicit = os.environ.get('DEEPBINNER_HOST_INFLATE_SHARE')
    if explicit:
        return max(0, min(100, int(explicit)))
    cores, gpus = float(usable_cpus()), float(max(n_gpus, 1))
    return 100 if cores / gpus >= HOST_ONLY_CORES_PER_GPU else 0


def queue_clones(pair, n_more):
    """``n_more`` further (start, end) pairs of the same weights on the same GPU.  They are kept
    ON the pair's first model (``_queue_clones``: partner pair -> clones, matched by identity), so
    that they live exactly as long as it does and ``HipModel.close()`` releases them with it - a
    cache keyed by ``id()`` outside the models would hand a new model that happens to get a freed
    model's id the old model's weights.  The pair is remembered through weak references (the
    holder is part of it: a strong one would be a cycle that only the collector breaks), and an
    entry whose partner is gone or closed gives its clones back at once."""
    holder = next(m for m in pair if m is not None)
    cache = holder.__dict__.setdefault('_queue_clones', [])
    more = None
    for entry in list(cache):
        known, clones = entry
        alive = [None if r is None else r() for r in known]
        if any(r is not None and (m is None or not getattr(m, 'handle', None))
               for r, m in zip(known, alive)):
            cache.remove(entry)             # the partner was closed or collected on its own
            for group in clones:
                for clone in group:
                    if clone is not None:
                        clone.close()
            continue
        if len(alive) == len(pair) and all(a is b for a, b in zip(alive, pair)):
            more = clones
    if more is None:
        more = []
        cache.append((tuple(None if m is None else weakref.ref(m) for m in pair), more))
    while len(more) < n_more:
        more.append(tuple(m.clone() if m is not None else None for m in pair))
    return more[:n_more]


def inflate_queues(replicas, host_share=50):
    """-> (replicas, models): INFLATE_QUEUES (start, end) pairs per GPU instead of one - more
    model replicas on the same GPU, each with its own streams and buffers (kept on the models
    they copy for the passes to come: queue_clones) - so that one container's chunks are
    inflated while the one before it is classified; every model's forward launches leave INFLATE_CUS CUs to the inflate kernels
    (``reserve_cus(0)`` on the models returned gives them back).  DEEPBINNER_INFLATE_QUEUES /
    DEEPBINNER_INFLATE_CUS."""
    n_queues = int(os.environ.get('DEEPBINNER_INFLATE_QUEUES', 0) or 0)
    if n_queues < 1:
        # (with kernel 1 one lane per stream a container the GPU inflated alone lasted as long as
        # its longest read, and twice the queues hid that; with one wavefront per stream three
        # queues are best at any share: 151 k reads/s against 142 k with six, the host inflating
        # nothing - profiles/r05_wave/README.md)
        n_queues = INFLATE_QUEUES
    n_cus = max(0, int(os.environ.get('DEEPBINNER_INFLATE_CUS', INFLATE_CUS)))
    out = []
    for pair in replicas:
        out.extend([pair] + queue_clones(pair, n_queues - 1))
    # queue k of every GPU before queue k + 1 of any: consecutive containers go to different GPUs
    out = [out[d * n_queues + k] for k in range(n_queues) for d in range(len(replicas))]
    models = [m for pair in out for m in pair if m is not None]
    for model in models:
        model.reserve_cus(n_cus if n_queues > 1 else 0)
    return out, models


def fast5s_under(directory):
    """``sorted(directory.glob('**/*.fast5'))`` as strings - the reference's listing
    (realtime.py:73-83), in the same order, by a walk that is four times as fast on a directory of
    100,000 files (it is done again for every pass)."""
    found = []
    for where, _, names in os.walk(str(directory)):
        found.extend(os.path.join(where, name) for name in names if name.endswith('.fast5'))
    found.sort(key=lambda path: path.split(os.sep))
    return found


def bin_name(barcode_call):
    """Directory a call is filed under (reference realtime.py:146-150)."""
    return 'unclassified' if barcode_call == 'none' else 'barcode%02d' % int(barcode_call)


class MoveTally:
    """Files one pass's fast5s into their bins and keeps count of what could not be moved."""

    def __init__(self, out_dir, total):
        self.out_dir, self.total = pathlib.Path(out_dir), total
        self.moved = self.clashes = self.failures = 0
        self._bins = {}             # barcode call -> its directory (made once)
        self._shown = -1

    def _bin(self, barcode_call):
        target = self._bins.get(barcode_call)
        if target is None:
            target = self.out_dir / bin_name(barcode_call)
            if not target.is_dir():
                try:
                    target.mkdir(parents=True)
                except OSError:
                    sys.exit('Error: unable to create output directory {}'.format(target))
            target = self._bins[barcode_call] = str(target)
        return target

    def file(self, fast5_file, barcode_call, unmovable):
        """One file into its bin (reference realtime.py:111-144): never over a file that is
        there already; a rename where that works (the same file system), a copy where not."""
        target = os.path.join(self._bin(barcode_call), os.path.basename(fast5_file))
        if os.path.isfile(target):
            self.clashes += 1
            unmovable.add(fast5_file)          # never look at it again
        else:
            try:
                try:
                    os.rename(fast5_file, target)
                except OSError:
                    shutil.move(fast5_file, target)
                self.moved += 1
            except OSError:
                self.failures += 1
        # (the reference redraws the line per file; so does this up to a thousand files a pass,
        # beyond that a thousand times a pass)
        if (self.moved - self._shown >= self.total // 1000 or
                self.moved + self.clashes + self.failures >= self.total):
            self._shown = self.moved
            print('\rMoving fast5s:      {} / {} ({:.1f}%)'.format(
                self.moved, self.total, 100.0 * self.moved / self.total), end='', flush=True)

    def report(self):
        print()
        where = str(self.out_dir)
        for count, one, many in (
                (self.clashes,
                 'Error: could not move 1 fast5 file because it already exists in {}',
                 'Error: could not move {} fast5 files because they already exist in {}'),
                (self.failures, 'Error: failed to move 1 fast5 file to {}',
                 'Error: failed to move {} fast5 files to {}')):
            if count == 1:
                print(one.format(where))
            elif count > 1:
                print(many.format(count, where))
        if self.failures == self.total:
            sys.exit('Error: no files were successfully moved to {}'.format(where))


class Session:
    """One ``deepbinner realtime`` run: the loaded models plus what has been seen so far."""

    def __init__(self, args):
        self.args = args
        args.verbose = False
        self.in_dir, self.out_dir = pathlib.Path(args.in_dir), pathlib.Path(args.out_dir)
        self.out_inside_in = self.in_dir in self.out_dir.parents
        classify.set_tensorflow_threads(args)
        (self.start_model, self.start_size, self.end_model, self.end_size, self.n_classes,
         _) = classify.load_and_check_models(args.start_model, args.end_model, args.scan_size,
                                             out_dest=sys.stdout)
        self.unmovable = set()
        self._names_in_bin = {}
        self.table_only = os.environ.get('DEEPBINNER_REALTIME_TABLE_ONLY') == '1'
        self._prepare_out_dir()

    def _prepare_out_dir(self):
        if self.out_dir.is_file():
            sys.exit('Error: {} is an existing file'.format(self.out_dir))
        if not self.out_dir.is_dir():
            try:
                os.makedirs(str(self.out_dir), exist_ok=True)
            except OSError:
                sys.exit('Error: unable to create output directory {}'.format(self.out_dir))
            print()
            print('Making output directory: {}/'.format(self.out_dir))

    def _models(self):
        return (self.start_model, self.start_size, self.end_model, self.end_size, self.n_classes)

    def waiting_files(self):
        """fast5 files under in_dir that are neither already binned nor given up on."""
        found = fast5s_under(self.in_dir)
        if self.out_inside_in:
            binned = set(fast5s_under(self.out_dir))
            found = [f for f in found if f not in binned]
        return [f for f in found if f not in self.unmovable]

    def handle(self, fast5s):
        kind = determine_single_or_multi_fast5s(fast5s)
        print()
        print('Found {:,} fast5 files in {}'.format(len(fast5s), self.args.in_dir))
        assert kind in PER_PASS
        todo = fast5s[:PER_PASS[kind]]
        if kind == 'single':
            calls = self._bin_one_read_files(todo)
        elif shutil.which('multi_to_single_fast5') is None:
            # every container that waits, in one stream; reported pass by pass as the reference
            # would work through them
            self.unmovable.update(fast5s)
            for calls, left in self._tabulate_multi_read_files(fast5s, PER_PASS[kind]):
                print()
                print_summary_table(calls, output=sys.stdout)
                if left:
                    print()
                    print('Found {:,} fast5 files in {}'.format(left, self.args.in_dir))
            return
        else:
            self.unmovable.update(todo)
            with tempfile.TemporaryDirectory() as scratch:
                print('Unpacking fast5s with multi_to_single_fast5:')
                for path in todo:
                    subprocess.check_output(['multi_to_single_fast5', '-i', path, '-s', scratch])
                print()
                unpacked = [str(p) for p in sorted(pathlib.Path(scratch).glob('**/*.fast5'))]
                calls = self._bin_one_read_files(unpacked)
        print_summary_table(calls, output=sys.stdout)

    def _bin_one_read_files(self, fast5s):
        calls, source = classify.classify_fast5_files(
            fast5s, *self._models(), self.args, full_output=False, verified_single_read=True)
        print()
        tally = MoveTally(self.out_dir, len(fast5s))
        for read_id, call in calls.items():
            tally.file(source[read_id], call, self.unmovable)
        tally.report()
        return calls

    # ---- multi-read containers ---------------------------------------------------------------
    # The reference unpacks them with an external tool and bins the copies (realtime.py:183-190).
    # Here they are classified where they are: a container is ONE unit of work - loaded by the
    # native loader's thread team several containers ahead (fast5_native.stream_reads), its packed
    # buffer (pinned host memory when a GPU model is loaded) handed to the C ABI as it is, both
    # models and combine_calls in one call (classify.classify_packed_numbers) on whichever device
    # replica is next (classify.dispatch_batches) - BASELINE.json configs[4]: one host streaming
    # multi-read files into several GPUs.
    def _native_writer(self):
        """The one-read files of the bins are written by libdeepbinner_fast5.so itself
        (f5_write_single_reads: the chunk as stored + the read's metadata, no inflate, no deflate,
        no Python per read) - unless the Python reader was asked for, or
        DEEPBINNER_PYTHON_WRITER=1 asks for hdf5_write.py (the writer this one is tested
        against)."""
        return (not self.table_only and reader_kind() == 'native' and
                os.environ.get('DEEPBINNER_PYTHON_WRITER') != '1')

    def _signals_wanted(self):
        """Whole signals are loaded only for the Python writer to write them out again."""
        return not self.table_only and not self._native_writer()

    def _keep(self):
        """Samples per read end the loaders keep: only the scanned ends (all that call_batch
        looks at) unless the whole signals are needed."""
        return None if self._signals_wanted() else classify.scanned_end_samples(self.args.scan_size)

    def _packed_containers(self, fast5s):
        """(container number, path, read ids, samples, offsets) per readable container, in order;
        unreadable reads are dropped (the reference skips what it cannot read,
        load_fast5s.py:47-49)."""
        import numpy as np
        from . import fast5_native
        threads = int(getattr(self.args, 'loader_procs', 0) or 0)
        stream = fast5_native.stream_reads(fast5s, keep=self._keep(), threads=threads,
                                           depth=int(os.environ.get('DEEPBINNER_LOADER_DEPTH', 0)))
        for index, ids, samples, offsets, status in stream:
            if ids is None:
                continue
            classify.warn_about_filters(status)
            where = list(range(len(ids)))          # which read of the container each one is
            if any(rid is None for rid in ids):
                where = [i for i, rid in enumerate(ids) if rid is not None]
                parts = [samples[offsets[i]:offsets[i + 1]] for i in where]
                lengths = [len(part) for part in parts]
                samples = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int16)
                offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
                ids = [ids[i] for i in where]
            yield index + 1, fast5s[index], ids, samples, offsets, where

    def _classify_container(self, item, start_replica, end_replica):
        number, path, ids, samples, offsets, where = item
        numbers = classify.classify_packed_numbers(samples, offsets, start_replica, end_replica,
                                                   self.args)
        names = [classify.call_name(c) for c in numbers.tolist()]
        signal = (lambda i: samples[offsets[i]:offsets[i + 1]]) if self._signals_wanted() else None
        return number, path, ids, names, signal, where

    # The same with (part of) the inflating on the GPU: the loader hands over Signal chunks as
    # stored - zlib streams; 85 % of what loading a read costs a CPU core is inflating them, and a
    # host has few cores per GPU - and dbh_classify_pair_deflated does the
    # rest.  The host's threads keep the longest streams of every container (a lane of the GPU
    # decoder walks ONE stream, however long): `host_inflate_share` of the bytes.
    def _raw_containers(self, fast5s, host_share, n_gpus=1):
        from . import fast5_native
        threads = int(getattr(self.args, 'loader_procs', 0) or 0)
        if threads <= 0 and host_share == 0:
            # nothing to inflate: a read costs a loader thread ~5 us, and a GPU takes ~210 k a second -
            # two threads feed it, sixteen cost the process 19 us of CPU per read instead of 13
            # (woken sixteen times per container for a fifth of what they can deliver:
            # profiles/r06_loader/loader_team_size.txt)
            threads = min(usable_cpus(), RAW_LOADER_THREADS_PER_GPU * max(1, n_gpus))
        stream = fast5_native.stream_raw(fast5s, threads=threads, host_inflate_above=-host_share,
                                         depth=int(os.environ.get('DEEPBINNER_LOADER_DEPTH', 0)))
        for index, ids, offsets, status, comp, records in stream:
            if ids is None:
                continue
            classify.warn_about_filters(status)
            yield index + 1, fast5s[index], ids, offsets, comp, records

    def _classify_raw_container(self, item, start_replica, end_replica):
        import numpy as np
        from . import fast5_native, hip_backend
        number, path, ids, offsets, comp, records = item
        cus = inflate_cus_for(records['comp_bytes'].tolist(), records['mode'].tolist())
        if cus is not None:
            for model in (start_replica, end_replica):
                if model is not None:
                    model.reserve_cus(cus)
        result = hip_backend.classify_pair_deflated(
            start_replica, end_replica, comp, records, offsets, int(self.args.scan_size),
            self.args.score_diff, classify.combine_mode(self.args) if start_replica is not None and
            end_replica is not None else 'require_either', want_samples=self._signals_wanted())
        numbers, stream_status = result[0], result[1]
        samples = result[2] if self._signals_wanted() else None
        redone = {}
        for i in sorted(set(records['read'][stream_status != 0].tolist())):
            # a stream the GPU decoder refused (damaged, or beyond it): zlib on the host has the
            # last word, as it has in the reference (h5py -> libhdf5 -> zlib)
            try:
                _, one, one_offsets, one_status = fast5_native.load_reads(path, first=i, count=1,
                                                                          threads=1)
            except OSError:
                one_status = [1]
            if one_status[0] != 0:
                ids[i] = None
                continue
            numbers[i] = classify.classify_packed_numbers(one, one_offsets, start_replica,
                                                          end_replica, self.args)[0]
            redone[i] = np.array(one)
        keep = [i for i, rid in enumerate(ids) if rid is not None]
        names = [classify.call_name(int(numbers[i])) for i in keep]

        def signal(k):
            i = keep[k]
            return redone[i] if i in redone else samples[offsets[i]:offsets[i + 1]]

        return (number, path, [ids[i] for i in keep], names,
                signal if samples is not None else None, keep)

    def _read_chunks(self, fast5s):
        """The same units for the Python reader and for models without the packed entry point:
        (container number, path, read ids, signals) per --batch_size reads."""
        for number, path in enumerate(fast5s, start=1):
            try:
                reads = list(iter_reads(path))
            except OSError:
                continue
            for chunk in classify.chunker(reads, self.args.batch_size):
                yield number, path, [r[0] for r in chunk], [r[1] for r in chunk]

    def _classify_chunk(self, item, start_replica, end_replica):
        number, path, ids, signals = item
        found = {}
        classify.classify_read_batch(ids, signals, start_replica, self.start_size, end_replica,
                                     self.end_size, self.n_classes, self.args, found)
        return number, path, ids, [found[rid] for rid in ids], signals.__getitem__, None

    def _tabulate_multi_read_files(self, fast5s, per_pass):
        """Classifies (and bins) the reads of the multi-read containers ``fast5s`` where they are;
        a generator: after every ``per_pass`` containers -> ({read id: call} of those, how many
        containers are left)."""
        from concurrent.futures import ThreadPoolExecutor
        from .hdf5_write import write_single_read_fast5
        n_writers = min(16, usable_cpus())
        writers = ThreadPoolExecutor(max_workers=n_writers,
                                     thread_name_prefix='deepbinner-fast5-writer')
        # at most this many reads wait to be written (each holds its signal - and with it the
        # container's buffer - in memory)
        in_flight = threading.BoundedSemaphore(64 * n_writers)
        failures = []
        clashes = []        # files that were there already (an earlier run into the same out_dir):
                            # left as they are and counted, like the reference's moves (:111-144)

        def bin_read(name, read_id, signal, call, source, metadata):
            try:
                target = self.out_dir / bin_name(call)
                os.makedirs(str(target), exist_ok=True)
                write_single_read_fast5(str(target / (name + '.fast5')), read_id, signal,
                                        metadata=metadata(source, read_id))
            except FileExistsError as e:
                clashes.append(str(e))
            except Exception as e:                       # surfaces when the pass ends
                failures.append(e)
            finally:
                in_flight.release()

        models = [m for m in (self.start_model, self.end_model) if m is not None]
        packed = reader_kind() == 'native' and all(hasattr(m, 'classify_packed') for m in models)
        replicas = classify.device_replicas(self.start_model, self.end_model)
        host_share = host_inflate_share(len({getattr(r[0] or r[1], 'device', 0) for r in replicas}))
        queues = []
        if packed and host_share < 100 and all(hasattr(m, 'handle') for m in models):
            items = self._raw_containers(
                fast5s, host_share, len({getattr(r[0] or r[1], 'device', 0) for r in replicas}))
            work = self._classify_raw_container
            replicas, queues = inflate_queues(replicas, host_share)
        elif packed:
            items, work = self._packed_containers(fast5s), self._classify_container
        else:
            items, work = self._read_chunks(fast5s), self._classify_chunk
        metadata = MetadataSource() if not self.table_only else None     # (the Python writer's)
        # The native writer: one call per container, two containers at a time on background
        # threads (the library's own worker threads do the reads of a container in parallel - four
        # are as good as sixteen: the creates in a directory serialise -, the call releases the
        # interpreter lock), at most three containers waiting, so that writing container k
        # overlaps loading and classifying the ones behind it.
        native_jobs = ThreadPoolExecutor(max_workers=2, thread_name_prefix='deepbinner-fast5-bins')
        native_waiting = threading.BoundedSemaphore(3)
        known_bins = set()

        def bin_container(source, where, targets):
            from . import fast5_native
            try:
                t0 = time.perf_counter()
                status, _ = fast5_native.write_single_reads(source, where, targets,
                                                            threads=max(2, usable_cpus() // 4))
                if os.environ.get('DEEPBINNER_REALTIME_TIMING'):
                    print('wrote {} reads of {} in {:.1f} ms'.format(
                        len(targets), os.path.basename(source),
                        (time.perf_counter() - t0) * 1e3), file=sys.stderr)
                clashes.extend(t for t, st in zip(targets, status.tolist())
                               if st == fast5_native.F5_ERR_EXISTS)
                bad = [t for t, st in zip(targets, status.tolist())
                       if st not in (fast5_native.F5_OK, fast5_native.F5_ERR_EXISTS)]
                if bad:
                    failures.append(OSError('{} ({} of {} reads of {})'.format(
                        bad[0], len(bad), len(targets), source)))
            except Exception as e:                       # surfaces when the pass ends
                failures.append(e)
            finally:
                native_waiting.release()

        # The units go round the devices the models are replicated on (one, usually).  ALL the
        # containers given go through one loader stream and one dispatcher - nothing drains
        # between two passes - while the output keeps the reference's rhythm: a progress line and
        # a summary per `per_pass` containers (realtime.py:86-94 takes five multi-read files a pass).
        def group_of(number):
            return (number - 1) // per_pass

        def close_group(group, group_calls, group_written, jobs):
            for job in jobs:                    # this pass's files are on disk before it is
                job.result()                    # reported
            if failures:
                sys.exit('Error: failed to write {} one-read fast5 file{} into {} ({})'.format(
                    len(failures), '' if len(failures) == 1 else 's', self.out_dir, failures[0]))
            n_clashes = len(clashes)
            del clashes[:]
            if group_written - n_clashes > 0:
                print()
                print('Wrote {:,} one-read fast5 files into {}'.format(group_written - n_clashes,
                                                                      self.out_dir), end='')
            if n_clashes:
                print()
                print('Error: could not write {:,} one-read fast5 file{} because {} already exist{} '
                      'in {}'.format(n_clashes, '' if n_clashes == 1 else 's',
                                     'it' if n_clashes == 1 else 'they',
                                     's' if n_clashes == 1 else '', self.out_dir), end='')
            return group_calls, max(len(fast5s) - (group + 1) * per_pass, 0)

        group, calls, done, written, jobs = 0, {}, 0, 0, []
        try:
            with open(str(self.out_dir / 'multi_read_classifications.tsv'), 'at') as table:
                classify.print_classification_progress(0, 1, 'reads', out_dest=sys.stdout)
                for number, path, ids, names, signal, where in classify.dispatch_batches(
                        items, replicas, work):
                    while group_of(number) > group:
                        table.flush()
                        yield close_group(group, calls, written, jobs)
                        group, calls, done, written, jobs = group + 1, {}, 0, 0, []
                        classify.print_classification_progress(0, 1, 'reads', out_dest=sys.stdout)
                    calls.update(zip(ids, names))
                    table.writelines('{}\t{}\t{}\n'.format(rid, name, path)
                                     for rid, name in zip(ids, names))
                    if not self.table_only and signal is None and where is not None:
                        targets = []
                        for rid, name in zip(ids, names):
                            bin_dir = self.out_dir / bin_name(name)
                            if name not in known_bins:
                                os.makedirs(str(bin_dir), exist_ok=True)
                                known_bins.add(name)
                            targets.append(str(bin_dir / (self._file_name(rid, name) + '.fast5')))
                        native_waiting.acquire()
                        jobs.append(native_jobs.submit(bin_container, path, list(where), targets))
                        written += len(targets)
                    elif not self.table_only:      # zlib and file writes release the GIL
                        for i, (rid, name) in enumerate(zip(ids, names)):
                            in_flight.acquire()
                            jobs.append(writers.submit(bin_read, self._file_name(rid, name), rid,
                                                       signal(i), name, path, metadata))
                            written += 1
                    done += len(ids)
                    # the total is known once the pass's last container is open; until then,
                    # extrapolate
                    in_pass = min(per_pass, len(fast5s) - group * per_pass)
                    total = max(done * in_pass // (number - group * per_pass), 1)
                    classify.print_classification_progress(min(done, total), total, 'reads',
                                                           out_dest=sys.stdout)
            yield close_group(group, calls, written, jobs)
        finally:
            writers.shutdown(wait=True)
            native_jobs.shutdown(wait=True)
            if metadata is not None:
                metadata.close()
            for model in queues:                # the forward kernel gets every CU back
                model.reserve_cus(0)

    def _file_name(self, read_id, call):
        """File name (without .fast5) of a binned read.  The id is an attribute of an untrusted
        file: one that is not a plain name (path separators, dots only, control characters,
        over-long) is replaced by a digest of itself, and a name already used in this bin during
        this run gets a numbered suffix instead of overwriting the earlier read."""
        import hashlib
        plain = (0 < len(read_id) <= 128 and read_id not in ('.', '..') and
                 all(c.isalnum() or c in '-_.' for c in read_id))
        name = read_id if plain else 'read_' + hashlib.sha256(read_id.encode()).hexdigest()[:32]
        used = self._names_in_bin.setdefault(bin_name(call), set())
        candidate, k = name, 1
        while candidate in used:
            candidate = '{}_{}'.format(name, k)
            k += 1
        used.add(candidate)
        return candidate


class MetadataSource:
    """What a basecaller needs beside the signal: per read the attributes of ``Raw`` and of the
    ``channel_id`` / ``tracking_id`` / ``context_tags`` groups of its container (ont_fast5_api's
    multi_to_single_fast5 copies them; the reference bins its output, realtime.py:183-190).  Read
    with this package's Python HDF5 reader, one open container at a time per writer thread."""

    GROUPS = ('channel_id', 'tracking_id', 'context_tags')

    class _Slot:
        """One writer thread's open container (a plain object: ``close()`` runs on another
        thread, where the thread-local itself would show nothing)."""
        file = path = None

        def __init__(self):
            self.shared = {}

    def __init__(self):
        self._local = threading.local()
        self._slots = []            # every thread's slot, for close()
        self._lock = threading.Lock()

    def _slot(self):
        slot = getattr(self._local, 'slot', None)
        if slot is None:
            slot = self._local.slot = self._Slot()
            with self._lock:
                self._slots.append(slot)
        return slot

    def _container(self, path):
        from . import hdf5_lite
        slot = self._slot()
        if slot.path != path:
            if slot.file is not None:
                slot.file.close()
            slot.file = slot.path = None
            slot.file = hdf5_lite.File(path, 'r')
            slot.path = path
            slot.shared = {}
        return slot.file

    def __call__(self, path, read_id):
        try:
            container = self._container(path)
            group = container['read_' + read_id]
            found = {'read': dict(group.attrs.items()), 'Raw': dict(group['Raw'].attrs.items())}
            for name in self.GROUPS:
                if name in group:
                    child = group[name]
                    # the same channel / run for every read of a container, as a rule
                    key = (name, getattr(child, 'addr', id(child)))
                    shared = self._slot().shared
                    if key not in shared:
                        shared[key] = dict(child.attrs.items())
                    found[name] = shared[key]
            return found
        except (OSError, KeyError, ValueError):
            return None             # the signal and the read id are what binning cannot do without

    def close(self):
        """Closes every writer thread's container (call when the writers are idle)."""
        with self._lock:
            for slot in self._slots:
                try:
                    if slot.file is not None:
                        slot.file.close()
                except Exception:
                    pass
                slot.file = slot.path = None
                slot.shared = {}


def realtime(args):
    """Entry point of the ``realtime`` sub-command (reference realtime.py:28-70)."""
    print()
    session = Session(args)
    idle_announced = False
    try:
        while True:
            fast5s = session.waiting_files()
            if fast5s:
                time.sleep(POLL_SECONDS)        # let files that are still being written settle
                session.handle(fast5s)
                idle_announced = False
                continue
            if args.stop:
                return
            if idle_announced:
                print('.', end='', flush=True)
            else:
                print('\nWaiting for new fast5 files (Ctrl-C to stop)', end='', flush=True)
                idle_announced = True
            time.sleep(POLL_SECONDS)
    except KeyboardInterrupt:
        print('\n\nStopping Deepbinner real-time binning\n')

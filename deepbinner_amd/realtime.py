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
import pathlib
import shutil
import subprocess
import sys
import tempfile
import threading
import time

from . import classify
from .load_fast5s import determine_single_or_multi_fast5s, iter_reads, reader_kind
from .misc import print_summary_table

POLL_SECONDS = 5
PER_PASS = {'single': 20000, 'multi': 5}       # reference realtime.py:86-94


def bin_name(barcode_call):
    """Directory a call is filed under (reference realtime.py:146-150)."""
    return 'unclassified' if barcode_call == 'none' else 'barcode%02d' % int(barcode_call)


class MoveTally:
    """Files one pass's fast5s into their bins and keeps count of what could not be moved."""

    def __init__(self, out_dir, total):
        self.out_dir, self.total = pathlib.Path(out_dir), total
        self.moved = self.clashes = self.failures = 0

    def _bin(self, barcode_call):
        target = self.out_dir / bin_name(barcode_call)
        if not target.is_dir():
            try:
                target.mkdir(parents=True)
            except OSError:
                sys.exit('Error: unable to create output directory {}'.format(target))
        return target

    def file(self, fast5_file, barcode_call, unmovable):
        target = self._bin(barcode_call)
        if (target / pathlib.Path(fast5_file).name).is_file():
            self.clashes += 1
            unmovable.add(fast5_file)          # never look at it again
        else:
            try:
                shutil.move(fast5_file, str(target))
                self.moved += 1
            except OSError:
                self.failures += 1
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
        found = [str(p) for p in sorted(self.in_dir.glob('**/*.fast5'))]
        if self.out_inside_in:
            binned = {str(p) for p in self.out_dir.glob('**/*.fast5')}
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
            self.unmovable.update(todo)
            calls = self._tabulate_multi_read_files(todo)
            print()
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

    def _reads_of(self, path):
        """(read_id, signal) of every read in a multi-read file.  With the native reader the
        whole file is inflated by its worker threads in one call.  When the reads are only
        tabulated, only the scanned ends of long reads come back (all that call_batch looks at);
        when they are binned too, the whole signals do."""
        if reader_kind() != 'native':
            return iter_reads(path)
        from . import fast5_native
        keep = classify.scanned_end_samples(self.args.scan_size) if self.table_only else None
        try:
            ids, samples, offsets, status = fast5_native.load_reads(
                path, keep=keep, threads=int(getattr(self.args, 'loader_procs', 0) or 0))
        except OSError:
            return []
        classify.warn_about_filters(status)
        reads = classify.PackedBatch((rid, samples[offsets[i]:offsets[i + 1]])
                                     for i, rid in enumerate(ids) if rid is not None)
        if len(reads) == len(ids) and keep is not None:
            # nothing dropped, long reads already cut to their scanned ends: the packed buffer is
            # what the C ABI takes
            reads.samples, reads.offsets, reads.complete = samples, offsets, True
        return reads

    def _containers(self, fast5s):
        """(path, reads) per file, file k + 1 being loaded on a background thread (the native
        loader releases the GIL) while the caller classifies file k."""
        box = {}

        def load(path):
            try:
                reads = self._reads_of(path)
                box[path] = reads if isinstance(reads, list) else list(reads)
            except Exception as e:          # surfaces on the consuming side
                box[path] = e

        worker = None
        for k, path in enumerate(fast5s):
            if worker is None:
                load(path)
            else:
                worker.join()
            if k + 1 < len(fast5s):
                worker = threading.Thread(target=load, args=(fast5s[k + 1],), daemon=True)
                worker.start()
            reads = box.pop(path)
            if isinstance(reads, Exception):
                raise reads
            yield path, reads

    def _tabulate_multi_read_files(self, fast5s):
        from concurrent.futures import ThreadPoolExecutor
        from .hdf5_write import write_single_read_fast5
        calls, done, written = {}, 0, []
        writers = ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4),
                                     thread_name_prefix='deepbinner-fast5-writer')

        def bin_read(read_id, signal, call):
            target = self.out_dir / bin_name(call)
            os.makedirs(str(target), exist_ok=True)
            write_single_read_fast5(str(target / (read_id + '.fast5')), read_id, signal)

        with open(str(self.out_dir / 'multi_read_classifications.tsv'), 'at') as table:
            for n_files, (path, reads) in enumerate(self._containers(fast5s), start=1):
                signals_of = dict(reads) if not self.table_only else {}
                # the total is known once the last container is open; until then, extrapolate
                total = max((done + len(reads)) * len(fast5s) // n_files, 1)
                classify.print_classification_progress(done, total, 'reads', out_dest=sys.stdout)
                def chunks():
                    for at, chunk in enumerate(classify.chunker(reads, self.args.batch_size)):
                        ids = [r[0] for r in chunk]
                        signals = [r[1] for r in chunk]
                        if getattr(reads, 'complete', False):
                            # this chunk's part of the container's packed buffer, as the C ABI
                            # takes it
                            lo = at * self.args.batch_size
                            offsets = reads.offsets[lo:lo + len(chunk) + 1]
                            signals = classify.PackedSignals(
                                signals, reads.samples[offsets[0]:offsets[-1]],
                                offsets - offsets[0])
                        yield ids, signals

                def classify_chunk(chunk, start_replica, end_replica):
                    ids, signals = chunk
                    found = {}
                    classify.classify_read_batch(ids, signals, start_replica, self.start_size,
                                                 end_replica, self.end_size, self.n_classes,
                                                 self.args, found)
                    return ids, found

                # the chunks go round the devices the models are replicated on (one, usually)
                replicas = classify.device_replicas(self.start_model, self.end_model)
                for ids, found in classify.dispatch_batches(chunks(), replicas, classify_chunk):
                    calls.update(found)
                    table.writelines('{}\t{}\t{}\n'.format(rid, calls[rid], path) for rid in ids)
                    if not self.table_only:      # zlib and file writes release the GIL
                        written += [writers.submit(bin_read, rid, signals_of[rid], calls[rid])
                                    for rid in ids]
                    done += len(ids)
                    classify.print_classification_progress(min(done, total), total, 'reads',
                                                           out_dest=sys.stdout)
        for job in written:
            job.result()
        writers.shutdown()
        if written:
            print()
            print('Wrote {:,} one-read fast5 files into {}'.format(len(written), self.out_dir),
                  end='')
        return calls


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

"""
``deepbinner classify`` — host side of the hot path, a drop-in for the reference's
``deepbinner/classify.py`` (same function names, arguments, return values, stdout/stderr text and
``sys.exit('Error: ...')`` behaviour; reference line numbers cited per function).

What differs is where the arithmetic happens: the model object is a ``HipModel`` (weights resident
in HBM behind the C ABI of ``include/deepbinner_hip.h``) and ``call_batch`` hands the whole
per-batch job — window slicing, normalisation, the CNN, the min/max merge, renormalisation and the
barcode call — to the GPU in one ``dbh_classify_i16`` call.  A model object that only offers
``predict`` (seam b1) is still accepted and driven through the same windowing/merge logic on the
host, which is how the host logic is tested without a GPU.
"""

import os
import pathlib
import sys

import numpy as np

from . import hdf5_lite
from .load_fast5s import (find_all_fast5s, get_read_id_and_signal,
                          determine_single_or_multi_fast5s, LoaderPool, choose_loader_procs,
                          reader_kind)
from .misc import print_summary_table, usable_cpus
from .model_format import ModelWeights
from .trim_signal import normalise


_DEVICES = None        # HIP ordinals the models are replicated on (set_tensorflow_threads)


def build_model(weights):
    """ModelWeights -> device-resident model.  The single place a backend is chosen; there is no
    CPU implementation to fall back to.  With several devices (``--devices N``) the model is
    replicated, one copy per GPU, for the batch dispatcher (``dispatch_batches``)."""
    from .hip_backend import HipModel, use_pinned_loader_buffers
    if _DEVICES and len(_DEVICES) > 1:
        model = ReplicatedModel([HipModel(weights, device=d) for d in _DEVICES])
    else:
        model = HipModel(weights)
    # the native loader's batches go to the GPU as they are: keep them in pinned host memory, so
    # that the upload needs no staging copy (DEEPBINNER_PINNED_LOADER=0: pageable, for comparison)
    if reader_kind() == 'native' and os.environ.get('DEEPBINNER_PINNED_LOADER', '1') != '0':
        use_pinned_loader_buffers()
    return model


class ReplicatedModel:
    """One trained model resident on several GPUs.  It looks like the model on the first of
    them (``inputs`` / ``outputs`` / ``predict`` / ``classify_signals`` / ``classify_packed``:
    what ``load_trained_model`` and ``call_batch`` use); ``replicas`` is what the dispatcher
    deals batches to."""

    def __init__(self, replicas):
        self.replicas = list(replicas)
        first = self.replicas[0]
        self.inputs, self.outputs = first.inputs, first.outputs
        self.n_classes, self.input_size = first.n_classes, first.input_size
        self.predict = first.predict
        self.classify_signals = first.classify_signals
        self.classify_packed = first.classify_packed

    def close(self):
        for r in self.replicas:
            r.close()


def device_replicas(start_model, end_model):
    """[(start replica, end replica)] per device the models live on; one pair for plain models."""
    counts = {len(m.replicas) for m in (start_model, end_model) if isinstance(m, ReplicatedModel)}
    if not counts:
        return [(start_model, end_model)]
    n = counts.pop()
    assert not counts, 'start and end models are replicated on different device sets'
    pick = lambda m, d: m.replicas[d] if isinstance(m, ReplicatedModel) else m   # noqa: E731
    return [(pick(start_model, d), pick(end_model, d)) for d in range(n)]


def dispatch_batches(batches, replicas, work, depth=2):
    """The single-process multi-device dispatcher (BASELINE.json configs[4]: one host feeding
    several GPUs): ``batches`` (any iterator - the loaders produce them ahead of time on their
    own threads) are dealt round-robin to the devices; device d's worker thread runs
    ``work(batch, *replicas[d])`` on that device's model replicas - each with its own streams and
    pinned double buffers behind the C ABI, so the devices' H2D copies, kernels and D2H copies all
    overlap; at most ``depth`` batches per device are in flight; results come back in the order
    the batches went in.  The reference is one loop on one device (classify.py:141-171)."""
    import collections
    from concurrent.futures import ThreadPoolExecutor
    if len(replicas) == 1:
        for batch in batches:
            yield work(batch, *replicas[0])
        return
    pools = [ThreadPoolExecutor(max_workers=1, thread_name_prefix='deepbinner-device-%d' % d)
             for d in range(len(replicas))]
    pending = collections.deque()
    try:
        for k, batch in enumerate(batches):
            d = k % len(replicas)
            pending.append(pools[d].submit(work, batch, *replicas[d]))
            if len(pending) >= depth * len(replicas):
                yield pending.popleft().result()
        while pending:
            yield pending.popleft().result()
    finally:
        for pool in pools:
            pool.shutdown(wait=True)


def classify(args):
    """``deepbinner classify`` (reference classify.py:32-55): load the model(s), work out what
    ``args.input`` is and hand it to the matching driver."""
    set_tensorflow_threads(args)
    *models, model_count = load_and_check_models(args.start_model, args.end_model, args.scan_size)

    kind = determine_input_type(args.input)
    if kind == 'training_data':
        if model_count == 2:
            sys.exit('Error: training data can only be classified using a single model')
        print('', file=sys.stderr)
        classify_training_data(args.input, *models, args)
        return
    print('', file=sys.stderr)
    assert kind in ('directory', 'single_fast5')
    from . import sharding
    rank, _, world = sharding.env_world()
    if kind == 'single_fast5' and world > 1 and rank != 0:
        return          # one file is one rank's work: rank 0 classifies and prints it
    fast5s = find_all_fast5s(args.input, verbose=True) if kind == 'directory' else [args.input]
    if kind == 'directory' and world > 1:
        # one process per GPU (torch.distributed.run): shard the reads, gather the calls
        sharding.classify_fast5_files_sharded(fast5s, *models, args)
    else:
        classify_fast5_files(fast5s, *models, args)


def load_and_check_models(start_model_filename, end_model_filename, scan_size,
                          out_dest=None):
    """-> (start_model, start_input_size, end_model, end_input_size, output_size, model_count)
    (reference classify.py:58-83): either file name may be None; input sizes must fit the scan
    size and two models must agree on the number of classes."""
    loaded = []
    for filename in (start_model_filename, end_model_filename):
        if filename is None:
            loaded.append((None, None, None))
            continue
        model, input_size, n_classes = load_trained_model(filename, out_dest=out_dest)
        check_input_size(input_size, scan_size)
        loaded.append((model, input_size, n_classes))
    class_counts = [n for _, _, n in loaded if n is not None]
    if len(set(class_counts)) > 1:
        sys.exit('Error: two models have different number of barcode classes')
    (start_model, start_size, _), (end_model, end_size, _) = loaded
    return (start_model, start_size, end_model, end_size,
            class_counts[0] if class_counts else None, len(class_counts))


def load_trained_model(model_file, out_dest=None):
    """-> (model, input_size, output_size) from one of the reference's Keras-2.1.4 HDF5 model
    files or one of this package's ``.dbw`` weight files (reference classify.py:86-103, same
    messages; ``out_dest`` defaults to the stderr of the moment of the call, where the reference
    binds the one of import time)."""
    out_dest = sys.stderr if out_dest is None else out_dest
    if not pathlib.Path(model_file).is_file():
        sys.exit('Error: {} does not exist'.format(model_file))
    print('Loading {}... '.format(model_file), file=out_dest, end='', flush=True)
    invalid = ('Error: model input has incorrect shape - are you sure that {} is a valid '
               'model file?'.format(model_file))
    try:
        weights, _ = ModelWeights.load(str(model_file))
    except Exception:       # whatever a damaged or foreign file trips inside the readers
        sys.exit(invalid)
    model = build_model(weights)
    print('done', file=out_dest)
    # the shape contract of seam b1: one input (None, L, 1) with L > 10, one output (None, C)
    shapes = [tuple(t.shape) for t in model.inputs]
    if len(shapes) != 1 or len(shapes[0]) != 3 or shapes[0][2] != 1 or not shapes[0][1] > 10:
        sys.exit(invalid)
    out_shape = tuple(model.outputs[0].shape)
    if len(out_shape) < 2:
        sys.exit(invalid)
    if hasattr(model, 'classify_packed') and int(shapes[0][1]) != MODEL_INPUT_SIZE:
        # the loaders keep scanned_end_samples() of each read end, sized for this input length
        sys.exit('Error: model input size {} is not supported (the loaders and the device library '
                 'are built for {})'.format(int(shapes[0][1]), MODEL_INPUT_SIZE))
    return model, int(shapes[0][1]), int(out_shape[1])


def classify_fast5_files(fast5_files, start_model, start_input_size, end_model, end_input_size,
                         output_size, args, full_output=True, summary_table=True,
                         verified_single_read=False):
    """Reference classify.py:106-180. -> (classifications, read_id_to_fast5_file)."""
    if not fast5_files:
        sys.exit('Error: no fast5 files found')
    out_dest = sys.stderr if full_output else sys.stdout

    if not verified_single_read:
        if determine_single_or_multi_fast5s(fast5_files) == 'multi':
            sys.exit('Error: deepbinner classify requires one-read-per-file fast5s - convert with '
                     'multi_to_single_fast5 before running')

    using_read_starts = start_model is not None
    using_read_ends = end_model is not None

    print_classification_progress(0, len(fast5_files), 'fast5s', out_dest=out_dest)
    if full_output:
        print_output_header(args.verbose, using_read_starts, using_read_ends, output_size)

    classifications, read_id_to_fast5_file = {}, {}

    def classify_loaded(loaded, start_replica, end_replica):
        """One loaded batch on one device -> (its reads' files, calls, table rows)."""
        if isinstance(loaded, RawBatch):
            return _classify_raw_batch(loaded, start_replica, end_replica, args)
        files, read_ids, signals = {}, [], []
        for fast5_file, read_id, signal in loaded:
            if signal is None:
                continue
            files[read_id] = fast5_file
            read_ids.append(read_id)
            signals.append(signal)
        if getattr(loaded, 'complete', False):      # nothing was skipped: the packed buffer
            signals = PackedSignals(signals, loaded.samples, loaded.offsets)   # is these reads
        calls = {}
        lines = classify_read_batch(read_ids, signals, start_replica, start_input_size,
                                    end_replica, end_input_size, output_size, args, calls)
        return files, calls, lines

    replicas = device_replicas(start_model, end_model)
    host_share = raw_inflate_share(start_model, end_model, args, len(fast5_files), replicas)
    if host_share is None:
        batches = load_in_batches(fast5_files, args)
    else:
        # Signals as stored, inflated on the GPU beside the classification of the batch before:
        # several batches in flight per GPU, each on a replica of the models
        from . import realtime
        replicas, _ = realtime.inflate_queues(replicas, host_share)
        batches = _raw_batches(fast5_files, args, host_share, len(replicas))
    for files, calls, lines in dispatch_batches(batches, replicas, classify_loaded):
        read_id_to_fast5_file.update(files)
        classifications.update(calls)
        if full_output:
            for line in lines:
                print(line)
        print_classification_progress(len(classifications), len(fast5_files), 'fast5s',
                                      out_dest=out_dest)

    if full_output:
        print('', file=sys.stderr)
        if summary_table:
            print_summary_table(classifications)
    return classifications, read_id_to_fast5_file


MODEL_INPUT_SIZE = 1024        # every shipped model; dbh_model_create refuses anything else


def scanned_end_samples(scan_size, input_size=MODEL_INPUT_SIZE):
    """Samples at either end of a read that some window can touch: the loaders may drop the
    middle of longer reads.  Windows start every input_size // 2 samples up to scan_size and are
    input_size long (reference classify.py:330-349).  The loaders run before any model object is
    in reach, hence the constant - which is also what the device library insists on."""
    return int(scan_size) + input_size // 2


RAW_CLASSIFY_MIN_FILES = 8192       # below that the CPU loader is done before a pipeline fills
RAW_BATCH_FILES = 4096              # one-read files per GPU-inflated batch (a container's worth)


def raw_inflate_share(start_model, end_model, args, n_files, replicas):
    """Per cent of the inflating the host keeps when ``classify`` hands the Signals of one-read
    files to the GPU as stored (realtime.host_inflate_share), or None if they go through the
    CPU loader: another reader or backend, few files, or a host with cores to spare.  (The
    verbose table takes this route too: dbh_classify_pair_deflated_verbose hands the sides' calls
    and probabilities back.)"""
    models = [m for m in (start_model, end_model) if m is not None]
    if (reader_kind() != 'native' or
            n_files < int(os.environ.get('DEEPBINNER_RAW_CLASSIFY_MIN_FILES',
                                         RAW_CLASSIFY_MIN_FILES)) or
            not all(hasattr(pick, 'handle') for pair in replicas for pick in pair
                    if pick is not None) or not models):
        return None
    from . import realtime
    share = realtime.host_inflate_share(len({getattr(r[0] or r[1], 'device', 0)
                                             for r in replicas}))
    return share if share < 100 else None


class RawBatch:
    """A batch of one-read files with their Signals as stored (fast5_native.load_batch_raw)."""

    def __init__(self, files, loaded):
        self.files = list(files)
        self.read_ids, self.offsets, self.status, self.comp, self.records = loaded


def _raw_batches(fast5_files, args, host_share, n_queues):
    """RawBatch after RawBatch, loaded ahead of the GPU: two loads at a time on background
    threads (each on half of the loader's threads), as many waiting as there are queues."""
    import collections
    from concurrent.futures import ThreadPoolExecutor
    from . import fast5_native
    threads = int(getattr(args, 'loader_procs', 0) or 0) or max(1, min(32, usable_cpus()))
    size = max(int(args.batch_size), RAW_BATCH_FILES)
    chunks = list(chunker(fast5_files, size))

    def load(chunk):
        return RawBatch(chunk, fast5_native.load_batch_raw(chunk, max(1, threads // 2),
                                                            -host_share))

    with ThreadPoolExecutor(max_workers=2, thread_name_prefix='deepbinner-raw-loader') as pool:
        waiting = collections.deque()
        upcoming = iter(chunks)
        for chunk in upcoming:
            waiting.append(pool.submit(load, chunk))
            if len(waiting) > n_queues:
                break
        while waiting:
            batch = waiting.popleft().result()
            chunk = next(upcoming, None)
            if chunk is not None:
                waiting.append(pool.submit(load, chunk))
            if (batch.status == fast5_native.F5_ERR_MULTI).any():
                sys.exit('Error: Deepbinner does not (yet) support multi-read fast5 files')
            warn_about_filters(batch.status)
            yield batch


def _classify_raw_batch(batch, start_replica, end_replica, args):
    """One RawBatch on one queue -> (its reads' files, calls, table rows): upload, inflate, both
    models and combine_calls in one call of the C ABI; a stream the GPU's decoder refuses is
    inflated again by the host's loader, which has the last word."""
    from . import fast5_native, hip_backend
    both = start_replica is not None and end_replica is not None
    verbose = bool(getattr(args, 'verbose', False))
    result = hip_backend.classify_pair_deflated(
        start_replica, end_replica, batch.comp, batch.records, batch.offsets, int(args.scan_size),
        args.score_diff, combine_mode(args) if both else 'require_either', want_sides=verbose)
    numbers, stream_status = result[0], result[1]
    sides = result[2] if verbose else None
    read_ids = list(batch.read_ids)
    redone = {}           # read index -> its verbose row, for reads the host had to decode
    for i in sorted(set(batch.records['read'][stream_status != 0].tolist())):
        ids, samples, offsets, status = fast5_native.load_batch(
            [batch.files[i]], scanned_end_samples(args.scan_size), 1)
        if status[0] != 0:
            read_ids[i] = None
            continue
        if verbose:
            model = start_replica if start_replica is not None else end_replica
            row = classify_read_batch(
                [read_ids[i]], [samples[offsets[0]:offsets[1]]], start_replica,
                getattr(start_replica, 'input_size', None), end_replica,
                getattr(end_replica, 'input_size', None), model.n_classes, args, {})[0]
            redone[i] = row
            numbers[i] = _CALL_NAMES.index(row.split('\t')[1])
        else:
            numbers[i] = classify_packed_numbers(samples, offsets, start_replica, end_replica,
                                                 args)[0]
    files, calls, lines = {}, {}, []
    for i, (read_id, fast5_file, number) in enumerate(zip(read_ids, batch.files, numbers.tolist())):
        if read_id is None:
            continue
        files[read_id] = fast5_file
        calls[read_id] = _CALL_NAMES[number]
        if not verbose:
            lines.append(read_id + '\t' + _CALL_NAMES[number])
        elif i in redone:
            lines.append(redone[i])
        else:
            # the reference's verbose row (classify.py:157-171): per side the 2-decimal
            # probabilities, followed - with two models - by that side's own call
            output = [read_id, _CALL_NAMES[number]]
            for side, replica in (('start', start_replica), ('end', end_replica)):
                if replica is None:
                    continue
                output += ['%.2f' % x for x in sides[side + '_probs'][i]]
                if both:
                    output.append(_CALL_NAMES[int(sides[side + '_calls'][i])])
            lines.append('\t'.join(output))
    return files, calls, lines


def load_in_batches(fast5_files, args):
    """The reference's ``for fast5_batch in chunker(...)`` + per-file load (classify.py:141-150):
    yields, per batch of ``args.batch_size`` files, the list of (fast5_file, read_id, signal).
    With more than one loader process (``--loader_procs``, or automatically for big jobs) the
    files of later batches are loaded while the caller classifies the current one."""
    if reader_kind() == 'native':
        yield from _native_batches(fast5_files, args)
        return
    procs = choose_loader_procs(getattr(args, 'loader_procs', None), len(fast5_files))
    if procs <= 1:
        for fast5_batch in chunker(fast5_files, args.batch_size):
            yield [(f,) + tuple(get_read_id_and_signal(f)) for f in fast5_batch]
        return
    # only the scanned ends of a read matter to call_batch: spare the result pipe the middle
    keep = scanned_end_samples(args.scan_size)
    with LoaderPool(procs) as pool:
        batch = []
        for item in pool.load(list(fast5_files), keep):
            batch.append(item)
            if len(batch) == args.batch_size:
                yield batch
                batch = []
        if batch:
            yield batch


def _native_batches(fast5_files, args):
    """load_in_batches on the native loader (libdeepbinner_fast5.so): every batch is parsed and
    inflated by the library's own worker threads (``--loader_procs`` of them; 0 = one per hardware
    thread this process may keep busy - misc.usable_cpus - at most 32), and the next batch is loaded on a background thread - the
    call releases the GIL - while the caller classifies the current one."""
    from concurrent.futures import ThreadPoolExecutor
    from . import fast5_native
    keep = scanned_end_samples(args.scan_size)
    threads = int(getattr(args, 'loader_procs', 0) or 0) or max(1, min(32, usable_cpus()))
    batches = list(chunker(fast5_files, args.batch_size))

    def load(batch):
        return fast5_native.load_batch(batch, keep, threads)

    with ThreadPoolExecutor(max_workers=1) as executor:
        pending = executor.submit(load, batches[0]) if batches else None
        for i, batch in enumerate(batches):
            read_ids, samples, offsets, status = pending.result()
            pending = executor.submit(load, batches[i + 1]) if i + 1 < len(batches) else None
            if (status == fast5_native.F5_ERR_MULTI).any():
                sys.exit('Error: Deepbinner does not (yet) support multi-read fast5 files')
            warn_about_filters(status)
            loaded = PackedBatch((f, read_ids[k], samples[offsets[k]:offsets[k + 1]]
                                  if read_ids[k] is not None else None)
                                 for k, f in enumerate(batch))
            loaded.samples, loaded.offsets = samples, offsets
            loaded.complete = all(r is not None for r in read_ids)
            yield loaded


_FILTER_WARNING_GIVEN = False


def warn_about_filters(status):
    """The reference (h5py without the plugin) skips VBZ-compressed files without a word, and so
    does this package - but it says so once, on stderr, since a whole run of them classifies
    nothing."""
    global _FILTER_WARNING_GIVEN
    from . import fast5_native
    skipped = int((status == fast5_native.F5_ERR_FILTER).sum())
    if skipped and not _FILTER_WARNING_GIVEN:
        _FILTER_WARNING_GIVEN = True
        print('\nWarning: skipping reads whose signal is compressed with a filter this build '
              'cannot decode (VBZ?); convert them with compress_fast5 -c gzip', file=sys.stderr)


class PackedBatch(list):
    """What the native loader returns for a batch - the (fast5_file, read_id, signal) triples the
    reference's loop builds - together with the packed buffer the signals are slices of:
    ``samples`` / ``offsets`` in exactly the layout the C ABI takes (dbh_classify_i16), so that
    a batch without unreadable files can go to the GPU as it is (``complete``)."""
    samples = offsets = None
    complete = False


class PackedSignals(list):
    """The list of signals ``call_batch`` is handed, plus the packed form of the same reads."""

    def __init__(self, signals, samples, offsets):
        super().__init__(signals)
        self.packed = (samples, offsets)


_HELPER = __import__('threading').local()


def _helper_thread():
    """The thread that drives the end model beside the calling thread's start model - one per
    calling thread, i.e. one per device when the dispatcher runs several."""
    if getattr(_HELPER, 'pool', None) is None:
        from concurrent.futures import ThreadPoolExecutor
        _HELPER.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix='deepbinner-end-model')
    return _HELPER.pool


def _independent_models(start_model, end_model):
    """Two distinct GPU models can be driven from two threads at once; anything else (one model
    object used for both sides, a stand-in without the packed entry point) is called in turn."""
    return (start_model is not end_model and hasattr(start_model, 'classify_packed')
            and hasattr(end_model, 'classify_packed'))


_CALL_NAMES = ['none'] + [str(i) for i in range(1, 256)]


def call_name(number):
    """'none' or the barcode number as the reference prints it (classify.py:289-295)."""
    return _CALL_NAMES[number]


def _array_path(signals, start_model, end_model):
    """A packed batch and GPU models only: the calls can stay arrays until the table is printed."""
    models = [m for m in (start_model, end_model) if m is not None]
    return (getattr(signals, 'packed', None) is not None and bool(models)
            and all(hasattr(m, 'classify_packed') for m in models)
            and (len(models) == 1 or models[0] is not models[1]))


def combine_mode(args):
    """The DBH_REQUIRE_* name of the run's two-model rule (reference deepbinner.py:308-316)."""
    if getattr(args, 'require_both', False):
        return 'require_both'
    if getattr(args, 'require_start', False):
        return 'require_start'
    assert args.require_either
    return 'require_either'


def classify_packed_numbers(samples, offsets, start_model, end_model, args):
    """Final call numbers (0 = 'none') of a packed batch of any size, without the per-read Python
    of call_batch / combine_calls.  Two GPU models on one device: ONE call of the C ABI
    (``dbh_classify_pair_i16``: one upload - none if the loader left the batch in pinned memory -
    both models' kernels, the merges and ``combine_calls`` on the device).  Otherwise each model's
    own call, the end model's on a helper thread, combined as arrays."""
    scan_size = int(args.scan_size)
    if start_model is not None and end_model is not None:
        if (hasattr(start_model, 'handle') and hasattr(end_model, 'handle') and
                getattr(start_model, 'device', 0) == getattr(end_model, 'device', 1)):
            from . import hip_backend
            return hip_backend.classify_pair(start_model, end_model, samples, offsets, scan_size,
                                             args.score_diff, combine_mode(args))

        def numbers(model, side):
            return model.classify_packed(samples, offsets, side, scan_size, args.score_diff)[1]

        pending = _helper_thread().submit(numbers, end_model, 'end')
        start_numbers = numbers(start_model, 'start')
        return combine_call_numbers(start_numbers, pending.result(), args)
    model, side = (start_model, 'start') if start_model is not None else (end_model, 'end')
    return model.classify_packed(samples, offsets, side, scan_size, args.score_diff)[1]


def _classify_packed_batch(read_ids, signals, start_model, end_model, args, classifications):
    """classify_read_batch for a packed batch and non-verbose output (no probabilities to print):
    call numbers as arrays, turned into the same strings and table rows at the end."""
    samples, offsets = signals.packed
    final = classify_packed_numbers(samples, offsets, start_model, end_model, args)
    names = [_CALL_NAMES[c] for c in final.tolist()]
    classifications.update(zip(read_ids, names))
    return [read_id + '\t' + name for read_id, name in zip(read_ids, names)]


def classify_read_batch(read_ids, signals, start_model, start_input_size, end_model,
                        end_input_size, output_size, args, classifications):
    """The body of the reference's per-batch loop (classify.py:141-171) for reads already in
    memory: run the model(s), combine, record calls, and return the TSV lines."""
    using_read_starts = start_model is not None
    using_read_ends = end_model is not None
    if not args.verbose and _array_path(signals, start_model, end_model):
        return _classify_packed_batch(read_ids, signals, start_model, end_model, args,
                                      classifications)
    start_calls = start_probs = end_calls = end_probs = None
    if using_read_starts and using_read_ends and _independent_models(start_model, end_model):
        # the two models' host <-> device round trips side by side: the end model's call runs on
        # a helper thread (the C ABI releases the GIL and each model has its own streams)
        pending = _helper_thread().submit(call_batch, end_input_size, output_size, read_ids,
                                          signals, end_model, args, 'end')
        start_calls, start_probs = call_batch(start_input_size, output_size, read_ids, signals,
                                              start_model, args, 'start')
        end_calls, end_probs = pending.result()
    else:
        if using_read_starts:
            start_calls, start_probs = call_batch(start_input_size, output_size, read_ids,
                                                  signals, start_model, args, 'start')
        if using_read_ends:
            end_calls, end_probs = call_batch(end_input_size, output_size, read_ids, signals,
                                              end_model, args, 'end')
    lines = []
    for i, read_id in enumerate(read_ids):
        if using_read_starts and using_read_ends:
            final_barcode_call = combine_calls(start_calls[i], end_calls[i], args)
        elif using_read_starts:
            final_barcode_call = start_calls[i]
        else:
            final_barcode_call = end_calls[i]
        classifications[read_id] = final_barcode_call
        output = [read_id, final_barcode_call]
        if args.verbose:
            if using_read_starts:
                output += ['%.2f' % x for x in start_probs[i]]
                if using_read_ends:
                    output.append(start_calls[i])
            if using_read_ends:
                output += ['%.2f' % x for x in end_probs[i]]
                if using_read_starts:
                    output.append(end_calls[i])
        lines.append('\t'.join(output))
    return lines


def classify_training_data(input_file, start_model, start_input_size, end_model, end_input_size,
                           output_size, args):
    """Classify a text file of ``label<TAB>v1,v2,...`` lines with ONE model (reference
    classify.py:183-239).  Reads are named ``line_<n>_barcode_<label>``; like the reference, the
    windows are always cut from the start of each signal (classify.py:223-224), whichever model
    was given."""
    assert (start_model is None) != (end_model is None)
    model, input_size = ((start_model, start_input_size) if end_model is None
                         else (end_model, end_input_size))
    with open(input_file, 'rt') as text:
        records = [line.rstrip().split('\t') for line in text]
    total = len(records)
    print_classification_progress(0, total, 'training data')
    print_output_header(args.verbose, start_model is not None, end_model is not None, output_size)

    classifications = {}
    for first in range(0, max(total, 1), args.batch_size):
        batch = records[first:first + args.batch_size]
        read_ids = ['line_{}_barcode_{}'.format(first + k + 1, label)
                    for k, (label, _) in enumerate(batch)]
        signals = [np.array([int(v) for v in values.split(',')]) for _, values in batch]
        for k, signal in enumerate(signals):
            # the device path carries raw signals as int16, which is what a sequencer produces;
            # the reference would normalise any integers (float64) - say which line it is
            if len(signal) and (signal.min() < -32768 or signal.max() > 32767):
                sys.exit('Error: line {} of {} holds signal values outside the int16 range'
                         .format(first + k + 1, input_file))
        calls, probs = call_batch(input_size, output_size, read_ids, signals, model, args, 'start')
        for read_id, call, row in zip(read_ids, calls, probs):
            classifications[read_id] = call
            fields = [read_id, call] + (['%.2f' % p for p in row] if args.verbose else [])
            print('\t'.join(fields))
        print_classification_progress(len(classifications), total, 'training data')

    print('', file=sys.stderr)
    print_summary_table(classifications)


def determine_input_type(input_file_or_dir):
    """'directory' | 'single_fast5' | 'training_data' (reference classify.py:242-263): a file that
    is not HDF5 counts as training data when its first line is ``<int>TAB<int>,<int>,...`` with
    more than ten values."""
    path = pathlib.Path(input_file_or_dir)
    if path.is_dir():
        return 'directory'
    if not path.is_file():
        sys.exit('Error: {} is neither a file nor a directory'.format(input_file_or_dir))
    try:
        hdf5_lite.File(str(path), 'r').close()
    except OSError:
        pass
    else:
        return 'single_fast5'
    try:
        with open(str(path)) as lines:
            label, _, values = lines.readline().partition('\t')
        int(label)
        if len([int(v) for v in values.split('\t')[0].split(',')]) > 10:
            return 'training_data'
    except (ValueError, UnicodeDecodeError):
        pass
    sys.exit('Error: could not determine input type')


def chunker(seq, size):
    """Consecutive slices of ``seq`` of at most ``size`` items."""
    for first in range(0, len(seq), size):
        yield seq[first:first + size]


def print_output_header(verbose, using_read_starts, using_read_ends, output_size):
    """The TSV header row (reference classify.py:270-282)."""
    columns = ['read_ID', 'barcode_call']
    barcodes = ['none'] + [str(i) for i in range(1, output_size)]
    if verbose and using_read_starts and using_read_ends:
        for side in ('start', 'end'):
            columns += [side + '_' + b for b in barcodes] + [side + '_barcode_call']
    elif verbose:
        columns += barcodes
    print('\t'.join(columns))


def get_barcode_call_from_probabilities(probabilities, score_diff_threshold):
    """Reference classify.py:285-295: best class 0 -> 'none'; otherwise the best barcode must
    beat the runner-up (which may be class 0) by score_diff.  Ties go to the lower index."""
    ranked = sorted(enumerate(probabilities), key=lambda item: item[1], reverse=True)
    (best, best_p), (_, second_p) = ranked[0], ranked[1]
    if best == 0:
        return 'none'
    return str(best) if best_p - second_p >= score_diff_threshold else 'none'


def combine_calls(start_call, end_call, args):
    """Final call of a read from its start-model and end-model calls (reference
    classify.py:298-322): agreement always stands; otherwise require_both refuses, require_start
    keeps a start call the end model is silent on, require_either keeps whichever side called
    when the other is silent."""
    if start_call == end_call:
        return start_call
    if args.require_both:
        return 'none'
    if end_call == 'none':
        return start_call
    if args.require_start:
        return 'none'
    assert args.require_either
    return end_call if start_call == 'none' else 'none'


def combine_call_numbers(start_calls, end_calls, args):
    """``combine_calls`` over whole arrays of call numbers (0 = 'none'), as the device-side
    ``dbh_combine_calls_dev`` does it."""
    import numpy as np
    start_calls, end_calls = np.asarray(start_calls), np.asarray(end_calls)
    if args.require_both:
        keep = np.zeros(len(start_calls), dtype=bool)
        other = np.zeros_like(start_calls)
    elif args.require_start:
        keep, other = end_calls == 0, np.zeros_like(start_calls)
    else:
        assert args.require_either
        keep, other = end_calls == 0, np.where(start_calls == 0, end_calls, 0)
    return np.where(start_calls == end_calls, start_calls, np.where(keep, start_calls, other))


def call_batch(input_size, output_size, read_ids, signals, model, args, side):
    """Reference classify.py:325-384 -> (barcode_calls, probabilities)."""
    assert side in ('start', 'end')
    step_size = input_size // 2
    steps = int(args.scan_size / step_size)
    assert steps * step_size == args.scan_size

    if not read_ids:
        return [], []

    if hasattr(model, 'classify_signals'):
        # Seam b2: the whole of this function runs on the GPU.
        packed = getattr(signals, 'packed', None)
        if packed is not None and hasattr(model, 'classify_packed'):
            probs, calls = model.classify_packed(packed[0], packed[1], side, int(args.scan_size),
                                                 args.score_diff)
        else:
            probs, calls = model.classify_signals(signals, side, int(args.scan_size),
                                                  args.score_diff)
        barcode_calls = ['none' if c == 0 else str(int(c)) for c in calls]
        return barcode_calls, [row for row in probs]

    # Seam b1: host windowing around model.predict.
    merged = None
    for s in range(steps):
        sig_start = s * step_size
        sig_end = sig_start + input_size
        input_signals = np.zeros([len(read_ids), input_size], dtype=float)
        for i, signal in enumerate(signals):
            if side == 'start':
                window = normalise(signal[sig_start:sig_end])
                input_signals[i, :len(window)] = window          # zero-padded on the right
            else:
                a = max(len(signal) - sig_end, 0)
                b = max(len(signal) - sig_start, 0)
                window = normalise(signal[a:b])
                input_signals[i, input_size - len(window):] = window   # ... on the left
        labels = model.predict(np.expand_dims(input_signals, axis=2),
                               batch_size=args.batch_size)
        labels = np.asarray(labels)
        if merged is None:
            merged = np.array(labels, copy=True)
        else:
            # no-barcode: minimum over the ranges; each barcode: maximum over the ranges
            merged[:, 0] = np.minimum(merged[:, 0], labels[:, 0])
            merged[:, 1:] = np.maximum(merged[:, 1:], labels[:, 1:])

    probabilities = [make_sum_to_one(row) for row in merged]
    barcode_calls = [get_barcode_call_from_probabilities(p, args.score_diff)
                     for p in probabilities]
    return barcode_calls, probabilities


def make_sum_to_one(probabilities):
    """Reference classify.py:387-393, in float64 (what NumPy-1.x scalar promotion gave the
    reference): keep the no-barcode probability, rescale the rest to fill 1 - p_none."""
    values = [float(p) for p in probabilities]
    no_barcode_prob = values[0]
    factor = np.float64(1.0 - no_barcode_prob) / np.float64(sum(values[1:]))
    scaled = [p * factor for p in values]
    scaled[0] = no_barcode_prob
    return scaled


def check_input_size(input_size, scan_size):
    """Windows advance by half the model's input size, so that size must be even and the scan
    size a whole number of half-windows (reference classify.py:396-407, same messages)."""
    half, odd = divmod(input_size, 2)
    if odd:
        sys.exit('Error: the model input size must be even (currently {})'.format(input_size))
    if int(scan_size / half) * half != scan_size:
        examples = ', '.join([str(half * k) for k in range(2, 8)] + ['etc'])
        sys.exit('Error: --scan_size must be a multiple of half the model input size\n'
                 'acceptable values for --scan_size are ' + examples)


def print_classification_progress(completed, total, label, out_dest=None):
    out_dest = sys.stderr if out_dest is None else out_dest
    out_dest.write('\rClassifying %s: %s / %s (%.1f%%)' % (label, completed, total,
                                                          100.0 * completed / total))
    out_dest.flush()


def set_tensorflow_threads(args):
    """Reference classify.py:416-423 configured a TensorFlow session.  Here the only runtime
    choice is which GPU(s) to use: ``--devices N`` / ``DEEPBINNER_DEVICES=N`` replicates the models
    on GPUs 0..N-1 (``DEEPBINNER_DEVICE_ORDINALS=0,0`` names them explicitly - several replicas on
    one GPU are a test mode for one-GPU boxes) and deals the batches out to them; otherwise
    ``DEEPBINNER_DEVICE`` or, under a one-process-per-GPU launcher, ``LOCAL_RANK`` picks the one
    GPU.  The TensorFlow thread flags are accepted and ignored."""
    global _DEVICES
    n = int(getattr(args, 'devices', 0) or os.environ.get('DEEPBINNER_DEVICES', 0) or 0)
    if n > 1 and int(os.environ.get('WORLD_SIZE', 1)) > 1:
        # one process per GPU AND several GPUs per process: every rank would replicate on GPUs
        # 0..n-1.  The launcher has already dealt the GPUs out; say so instead of guessing.
        sys.exit('Error: --devices {} under a one-process-per-GPU launcher (WORLD_SIZE={}): use '
                 'one or the other'.format(n, os.environ['WORLD_SIZE']))
    if n > 1:
        explicit = os.environ.get('DEEPBINNER_DEVICE_ORDINALS')
        _DEVICES = [int(v) for v in explicit.split(',')] if explicit else list(range(n))
        if len(_DEVICES) != n:
            sys.exit('Error: DEEPBINNER_DEVICE_ORDINALS names {} devices, {} asked for'
                     .format(len(_DEVICES), n))
        from . import hip_backend
        visible = hip_backend.device_count()
        if max(_DEVICES) >= visible:
            sys.exit('Error: {} GPUs asked for, {} visible'.format(max(_DEVICES) + 1, visible))
        return
    _DEVICES = None
    ordinal = os.environ.get('DEEPBINNER_DEVICE', os.environ.get('LOCAL_RANK'))
    if ordinal is not None:
        from . import hip_backend
        hip_backend.set_device(int(ordinal))

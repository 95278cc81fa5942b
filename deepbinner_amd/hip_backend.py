"""
ctypes binding of ``libdeepbinner_hip.so`` (C ABI: ``include/deepbinner_hip.h``) and the
``HipModel`` object that stands where the reference has a Keras ``Model``:

* ``model.inputs[0].shape`` / ``model.outputs[0].shape`` / ``len(model.inputs)`` — what
  ``load_trained_model`` inspects (reference ``deepbinner/classify.py:92-99``);
* ``model.predict(x, batch_size=...)`` — seam b1 (``classify.py:361``);
* ``model.classify_signals(...)`` — seam b2, all of ``call_batch`` (``classify.py:325-384``) on
  the device.

The binding idiom follows the reference's own native binding
(``deepbinner/dtw_semi_global.py:30-41``): ``LoadLibrary`` + ``ndpointer(..., C_CONTIGUOUS)`` +
explicit ``restype``/``argtypes`` + caller-allocated outputs.

There is deliberately NO CPU fallback: if the library or a GPU is missing the error is loud.
"""

import ctypes
import os

import numpy as np
from numpy.ctypeslib import ndpointer

from .model_format import ModelWeights

# Kernel arguments in device memory: 54.4 us per forward launch against 56.5 us with the
# arguments fetched from host memory (HIP_FORCE_DEV_KERNARG=0), same box.  It is this image's
# default; say so before the HIP runtime starts, unless the user said otherwise.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')

_LIB_NAME = 'libdeepbinner_hip.so'
_lib = None

SIDE_START, SIDE_END = 0, 1


class HipBackendError(RuntimeError):
    pass


def library_path():
    # DEEPBINNER_HIP_LIB: developer knob for A/B runs of two builds on the same GPU box
    return os.environ.get('DEEPBINNER_HIP_LIB') or os.path.join(
        os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def _f32(flags='C_CONTIGUOUS'):
    return ndpointer(dtype=np.float32, flags=flags)


def load_library():
    """Load (once) and type the shared library.  Raises HipBackendError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.isfile(path):
        raise HipBackendError(
            '{} has not been built - run `python -c "import __graft_entry__ as g; g.build()"` '
            '(or `make -C deepbinner_amd/csrc`). There is no CPU fallback.'.format(path))
    # (the HIP runtime maps a process's streams onto four hardware queues unless told otherwise,
    # and streams that share one run their kernels one after the other: the streaming path keeps
    # several containers in flight, each on a stream of its own - realtime.inflate_queues)
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise HipBackendError('could not load {}: {}'.format(path, e)) from e

    c_int, c_i64, c_void_p, c_size_t = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_size_t
    P = ctypes.POINTER
    sigs = {
        'dbh_version': (ctypes.c_char_p, []),
        'dbh_status_string': (ctypes.c_char_p, [c_int]),
        'dbh_last_error': (ctypes.c_char_p, []),
        'dbh_device_count': (c_int, [P(c_int)]),
        'dbh_set_device': (c_int, [c_int]),
        'dbh_get_device': (c_int, [P(c_int)]),
        'dbh_device_name': (c_int, [c_int, ctypes.c_char_p, c_int]),
        'dbh_device_synchronize': (c_int, []),
        'dbh_malloc': (c_int, [P(c_void_p), c_size_t]),
        'dbh_free': (c_int, [c_void_p]),
        'dbh_malloc_host': (c_int, [P(c_void_p), c_size_t]),
        'dbh_free_host': (c_int, [c_void_p]),
        'dbh_host_device_pointer': (c_int, [c_void_p, P(c_void_p)]),
        'dbh_memcpy_h2d': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
        'dbh_memcpy_d2h': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
        'dbh_memcpy_d2d': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
        'dbh_stream_create': (c_int, [P(c_void_p)]),
        'dbh_stream_destroy': (c_int, [c_void_p]),
        'dbh_stream_synchronize': (c_int, [c_void_p]),
        'dbh_event_create': (c_int, [P(c_void_p)]),
        'dbh_event_destroy': (c_int, [c_void_p]),
        'dbh_event_record': (c_int, [c_void_p, c_void_p]),
        'dbh_event_synchronize': (c_int, [c_void_p]),
        'dbh_stream_wait_event': (c_int, [c_void_p, c_void_p]),
        'dbh_event_elapsed_ms': (c_int, [c_void_p, c_void_p, P(ctypes.c_float)]),
        'dbh_model_create': (c_int, [_f32(), c_i64, c_int, c_int, P(c_void_p)]),
        'dbh_model_destroy': (c_int, [c_void_p]),
        'dbh_model_input_size': (c_int, [c_void_p, P(c_int)]),
        'dbh_model_output_size': (c_int, [c_void_p, P(c_int)]),
        'dbh_predict': (c_int, [c_void_p, _f32(), c_i64, _f32()]),
        'dbh_predict_dev': (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
        'dbh_classify_i16': (c_int, [c_void_p, ndpointer(np.int16, flags='C_CONTIGUOUS'),
                                     ndpointer(np.int64, flags='C_CONTIGUOUS'), c_i64, c_int,
                                     c_int, ctypes.c_double, _f32(),
                                     ndpointer(np.int32, flags='C_CONTIGUOUS')]),
        'dbh_classify_pair_i16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int,
                                          ctypes.c_double, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p]),
        'dbh_model_set_host_group': (c_int, [c_void_p, c_i64]),
        'dbh_model_reserve_cus': (c_int, [c_void_p, c_int]),
        'dbh_host_alloc': (c_void_p, [c_size_t, c_void_p]),
        'dbh_host_release': (None, [c_void_p, c_void_p]),
        'dbh_host_is_pinned': (c_int, [c_void_p, c_size_t, P(c_int)]),
        'dbh_inflate_last_error': (ctypes.c_char_p, []),
        'dbh_inflate_workspace_bytes': (c_int, [c_i64, c_i64, P(c_size_t)]),
        'dbh_inflate_dev': (c_int, [c_void_p, c_i64, c_void_p, c_i64, c_i64, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p]),
        'dbh_inflate': (c_int, [c_void_p, c_size_t, c_void_p, c_i64, c_void_p, c_size_t, c_void_p,
                                c_int, P(ctypes.c_double)]),
        'dbh_classify_pair_deflated': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64,
                                               c_void_p, c_i64, c_int, ctypes.c_double, c_int,
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
        'dbh_classify_pair_deflated_verbose': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p,
                                                       c_i64, c_void_p, c_i64, c_int,
                                                       ctypes.c_double, c_int, c_void_p, c_void_p,
                                                       c_void_p, c_void_p, c_void_p, c_void_p,
                                                       c_void_p, c_void_p]),
        'dbh_classify_workspace_bytes': (c_int, [c_void_p, c_i64, c_int, P(c_size_t)]),
        'dbh_classify_i16_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int,
                                         ctypes.c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
        'dbh_classify_i16_batched_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_int,
                                                 c_int, ctypes.c_double, c_void_p, c_void_p,
                                                 c_void_p]),
        'dbh_normalise_windows_dev': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_void_p,
                                              c_void_p]),
        'dbh_merge_calls_dev': (c_int, [c_void_p, c_i64, c_int, c_int, ctypes.c_double, c_void_p,
                                        c_void_p, c_void_p]),
        'dbh_combine_calls_dev': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p]),
        'dbh_stage_floats': (c_int, [c_int, P(c_i64)]),
        'dbh_debug_forward': (c_int, [c_void_p, _f32(), c_i64, c_int, _f32()]),
        'dbh_forward_kernel_info': (c_int, [P(c_int), P(c_int), P(c_int)]),
        'dbh_forward_truncated_dev': (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p]),
        'dbh_forward_executed_mfmas': (c_int, [c_int, P(c_i64), P(c_i64)]),
        'dbh_forward_timeline': (c_int, [c_void_p, _f32(), c_i64,
                                         ndpointer(np.int64, flags='C_CONTIGUOUS')]),
        'dbh_model_set_read_length_hint': (c_int, [c_void_p, c_i64, c_i64]),
        'dbh_forward_timeline_i16': (c_int, [c_void_p, np.ctypeslib.ndpointer(np.int16, flags='C_CONTIGUOUS'),
                                     c_i64, np.ctypeslib.ndpointer(np.int64, flags='C_CONTIGUOUS')]),
        'dbh_forward_timing_enable': (c_int, [c_void_p, c_int]),
        'dbh_forward_timing_enable_span': (c_int, [c_void_p, c_int, c_int]),
        'dbh_forward_timing_read': (c_int, [c_void_p, P(ctypes.c_double), P(c_i64), P(c_i64)]),
        'dbh_forward_clock_enable': (c_int, [c_void_p, c_int]),
        'dbh_forward_clock_read': (c_int, [c_void_p, P(ctypes.c_double)]),
        'dbh_forward_phases_enable': (c_int, [c_void_p, c_int]),
        'dbh_forward_phases_read': (c_int, [c_void_p, P(ctypes.c_double), P(ctypes.c_int64)]),
        'dbh_comm_available': (c_int, []),
        'dbh_comm_last_error': (ctypes.c_char_p, []),
        'dbh_comm_init_all': (c_int, [c_int, P(c_int), c_int, P(c_void_p)]),
        'dbh_comm_unique_id': (c_int, [ctypes.c_char_p]),
        'dbh_comm_init_rank': (c_int, [ctypes.c_char_p, c_int, c_int, P(c_void_p)]),
        'dbh_comm_info': (c_int, [c_void_p, P(c_int), P(c_int), P(c_int)]),
        'dbh_comm_all_gather_i32': (c_int, [c_void_p, P(c_void_p), P(c_void_p), c_i64,
                                            P(c_void_p)]),
        'dbh_comm_destroy': (c_int, [c_void_p]),
    }
    for name, (restype, argtypes) in sigs.items():
        fn = getattr(lib, name)     # AttributeError here = header and library out of step
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    'dbh_version', 'dbh_status_string', 'dbh_last_error', 'dbh_device_count', 'dbh_set_device',
    'dbh_get_device', 'dbh_device_name', 'dbh_device_synchronize', 'dbh_malloc', 'dbh_free',
    'dbh_malloc_host', 'dbh_free_host', 'dbh_memcpy_h2d', 'dbh_memcpy_d2h', 'dbh_memcpy_d2d',
    'dbh_stream_create', 'dbh_stream_destroy', 'dbh_stream_synchronize', 'dbh_event_create',
    'dbh_event_destroy', 'dbh_event_record', 'dbh_event_synchronize', 'dbh_stream_wait_event',
    'dbh_event_elapsed_ms',
    'dbh_model_create', 'dbh_model_destroy', 'dbh_model_set_read_length_hint', 'dbh_model_input_size', 'dbh_model_output_size',
    'dbh_predict', 'dbh_predict_dev', 'dbh_classify_i16', 'dbh_classify_pair_i16',
    'dbh_model_set_host_group', 'dbh_model_reserve_cus', 'dbh_host_device_pointer',
    'dbh_host_alloc', 'dbh_host_release',
    'dbh_host_is_pinned',
    'dbh_classify_workspace_bytes', 'dbh_inflate_last_error', 'dbh_inflate_workspace_bytes',
    'dbh_inflate_dev', 'dbh_inflate', 'dbh_classify_pair_deflated',
    'dbh_classify_pair_deflated_verbose',
    'dbh_classify_i16_dev', 'dbh_classify_i16_batched_dev', 'dbh_normalise_windows_dev', 'dbh_merge_calls_dev', 'dbh_combine_calls_dev',
    'dbh_stage_floats', 'dbh_debug_forward', 'dbh_forward_kernel_info',
    'dbh_forward_truncated_dev', 'dbh_forward_executed_mfmas', 'dbh_forward_timeline', 'dbh_forward_timeline_i16', 'dbh_forward_timing_enable', 'dbh_forward_timing_enable_span',
    'dbh_forward_timing_read', 'dbh_forward_clock_enable', 'dbh_forward_clock_read', 'dbh_forward_phases_enable', 'dbh_forward_phases_read',
    'dbh_comm_available', 'dbh_comm_last_error', 'dbh_comm_init_all', 'dbh_comm_unique_id',
    'dbh_comm_init_rank', 'dbh_comm_info', 'dbh_comm_all_gather_i32', 'dbh_comm_destroy',
]


def check(status, what='libdeepbinner_hip call'):
    if status != 0:
        lib = load_library()
        msg = lib.dbh_status_string(status).decode()
        detail = (lib.dbh_comm_last_error() if status == 7 else lib.dbh_last_error()).decode()
        raise HipBackendError('{} failed: {}{}'.format(what, msg,
                                                       ' ({})'.format(detail) if detail else ''))


def device_count():
    lib = load_library()
    n = ctypes.c_int(0)
    status = lib.dbh_device_count(ctypes.byref(n))
    if status != 0:
        return 0
    return n.value


def device_name(ordinal=0):
    lib = load_library()
    buf = ctypes.create_string_buffer(256)
    check(lib.dbh_device_name(ordinal, buf, 256), 'dbh_device_name')
    return buf.value.decode()


def set_device(ordinal):
    check(load_library().dbh_set_device(int(ordinal)), 'dbh_set_device')


def get_device():
    ordinal = ctypes.c_int(0)
    check(load_library().dbh_get_device(ctypes.byref(ordinal)), 'dbh_get_device')
    return ordinal.value


COMBINE_MODES = {'require_either': 0, 'require_start': 1, 'require_both': 2}


def combine_calls_dev(start_calls_ptr, end_calls_ptr, n_reads, mode, out_ptr, stream=None):
    """combine_calls (classify.py:298-322) over device arrays of call numbers; ``mode`` is one of
    COMBINE_MODES.  Does not block."""
    check(load_library().dbh_combine_calls_dev(start_calls_ptr, end_calls_ptr, int(n_reads),
                                               COMBINE_MODES[mode], out_ptr, stream),
          'dbh_combine_calls_dev')


def synchronize():
    check(load_library().dbh_device_synchronize(), 'dbh_device_synchronize')


def classify_pair(start_model, end_model, samples, offsets, scan_size, score_diff,
                  mode='require_either', want_sides=False, want_probs=False):
    """One batch of packed reads (``samples`` int16, ``offsets`` int64: read i is
    ``samples[offsets[i]:offsets[i+1]]``, long reads possibly cut to their scanned ends) through
    the start and the end model in ONE call of the C ABI: one upload, both models' kernels, the
    min/max merges and ``combine_calls`` on the device -> final calls int32 [N] (0 = 'none').
    Either model may be None.  ``want_sides`` / ``want_probs``: also return the per-side calls /
    probabilities: (calls, (start_calls, end_calls), (start_probs, end_probs))."""
    lib = load_library()
    models = (start_model, end_model)
    if start_model is None and end_model is None:
        raise ValueError('no model')
    samples = np.ascontiguousarray(samples, dtype=np.int16)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    if n < 0 or (n and (offsets[0] != 0 or offsets[-1] != len(samples))):
        raise ValueError('offsets do not describe the sample buffer')
    calls = np.empty(max(n, 0), dtype=np.int32)
    n_classes = next(m.n_classes for m in models if m is not None)
    side_calls = [np.empty(n, dtype=np.int32) if (want_sides and m is not None) else None
                  for m in models]
    side_probs = [np.empty((n, n_classes), dtype=np.float32) if (want_probs and m is not None)
                  else None for m in models]
    ptr = lambda a: a.ctypes.data if a is not None else None      # noqa: E731
    if n > 0:
        check(lib.dbh_classify_pair_i16(
            start_model.handle if start_model is not None else None,
            end_model.handle if end_model is not None else None,
            samples.ctypes.data if samples.size else None, offsets.ctypes.data, n, int(scan_size),
            float(score_diff), COMBINE_MODES[mode], calls.ctypes.data, ptr(side_calls[0]),
            ptr(side_calls[1]), ptr(side_probs[0]), ptr(side_probs[1])), 'dbh_classify_pair_i16')
    if want_sides or want_probs:
        return calls, tuple(side_calls), tuple(side_probs)
    return calls


INFLATE_ZLIB, INFLATE_STORED = 0, 1
# dbh_inflate_stream (include/deepbinner_hip.h)
INFLATE_STREAM = np.dtype([('comp_offset', '<i8'), ('comp_bytes', '<i8'), ('out_offset', '<i8'),
                           ('out_bytes', '<i8'), ('mode', '<i4'), ('reserved', '<i4')])


def inflate(comp, streams, out_bytes, streams_per_lane=0):
    """zlib streams inflated on the GPU (``dbh_inflate``: host buffers in and out - tests and
    tools; the classify path keeps everything on the device).  ``comp``: uint8 array holding the
    streams, ``streams``: array of INFLATE_STREAM records, ``out_bytes``: size of the output
    buffer, ``streams_per_lane``: how many streams a lane of kernel 1 takes one after the other
    (0 = 1; the order is the records') -> (output uint8 array, status int32 per stream,
    milliseconds the two kernels took)."""
    comp = np.ascontiguousarray(comp, dtype=np.uint8)
    streams = np.ascontiguousarray(streams, dtype=INFLATE_STREAM)
    out = np.zeros(int(out_bytes), dtype=np.uint8)
    status = np.zeros(len(streams), dtype=np.int32)
    ms = ctypes.c_double(0)
    check(load_library().dbh_inflate(comp.ctypes.data, comp.nbytes, streams.ctypes.data,
                                     len(streams), out.ctypes.data, out.nbytes, status.ctypes.data,
                                     int(streams_per_lane), ctypes.byref(ms)), 'dbh_inflate')
    return out, status, ms.value


def classify_pair_deflated(start_model, end_model, comp, streams, offsets, scan_size, score_diff,
                           mode='require_either', want_samples=False, want_stages=False,
                           want_sides=False):
    """A raw batch of the native loader (``fast5_native.stream_raw``: the reads' Signal chunks
    as stored) -> final calls int32 [N] and the decoder's status per stream, in ONE call of the C
    ABI: upload, inflate on the GPU, both models, ``combine_calls``.  ``comp`` must include its 64
    bytes of padding (``stream_raw``'s arrays do).  ``want_samples``: also the decoded signals
    (int16, all reads back to back); ``want_stages``: milliseconds of upload / inflate / classify
    on the device; ``want_sides``: also what the verbose table prints - a dict with the sides'
    own calls (``start_calls`` / ``end_calls``, int32 [N]) and merged probabilities
    (``start_probs`` / ``end_probs``, float32 [N, classes]); None for a side without a model."""
    lib = load_library()
    comp = np.ascontiguousarray(comp, dtype=np.uint8)
    streams = np.ascontiguousarray(streams)
    if streams.dtype.itemsize != INFLATE_STREAM.itemsize:
        raise ValueError('stream records of {} bytes'.format(streams.dtype.itemsize))
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    calls = np.empty(max(n, 0), dtype=np.int32)
    status = np.zeros(len(streams), dtype=np.int32)
    samples = np.empty(int(offsets[-1]) if want_samples and n > 0 else 0, dtype=np.int16)
    stages = (ctypes.c_double * 3)()
    sides = {'start_calls': None, 'end_calls': None, 'start_probs': None, 'end_probs': None}
    if want_sides:
        for side, model in (('start', start_model), ('end', end_model)):
            if model is not None:
                sides[side + '_calls'] = np.zeros(max(n, 0), dtype=np.int32)
                sides[side + '_probs'] = np.zeros((max(n, 0), model.n_classes), dtype=np.float32)

    def ptr(a):
        return a.ctypes.data if a is not None else None
    if n > 0:
        check(lib.dbh_classify_pair_deflated_verbose(
            start_model.handle if start_model is not None else None,
            end_model.handle if end_model is not None else None,
            comp.ctypes.data, max(comp.nbytes - 64, 0), streams.ctypes.data, len(streams),
            offsets.ctypes.data, n, int(scan_size), float(score_diff), COMBINE_MODES[mode],
            calls.ctypes.data, status.ctypes.data, samples.ctypes.data if want_samples else None,
            stages if want_stages else None, ptr(sides['start_calls']), ptr(sides['end_calls']),
            ptr(sides['start_probs']), ptr(sides['end_probs'])), 'dbh_classify_pair_deflated_verbose')
    out = [calls, status]
    if want_samples:
        out.append(samples)
    if want_stages:
        out.append(list(stages))
    if want_sides:
        out.append(sides)
    return tuple(out)


_PINNED_LOADER = False


def use_pinned_loader_buffers(on=True):
    """Have the native fast5 loader keep its packed batches in pinned host memory (this library's
    ``dbh_host_alloc`` / ``dbh_host_release`` handed to ``f5_set_sample_allocator`` as plain C
    function pointers: the two libraries do not link each other), so that ``classify_packed`` /
    ``classify_pair`` upload them without a staging copy.  Needs a GPU; idempotent."""
    global _PINNED_LOADER
    from . import fast5_native
    if on == _PINNED_LOADER:
        return
    if on:
        lib = load_library()
        fast5_native.set_sample_allocator(ctypes.cast(lib.dbh_host_alloc, ctypes.c_void_p).value,
                                          ctypes.cast(lib.dbh_host_release, ctypes.c_void_p).value)
    else:
        fast5_native.set_sample_allocator(None, None)
    _PINNED_LOADER = bool(on)


def is_pinned(array):
    """Does this numpy array lie in pinned host memory?"""
    flag = ctypes.c_int(0)
    check(load_library().dbh_host_is_pinned(array.ctypes.data, array.nbytes, ctypes.byref(flag)),
          'dbh_host_is_pinned')
    return bool(flag.value)


class Stream:
    """A non-blocking HIP stream owned through the C ABI; ``ptr`` is the raw hipStream_t (what the
    ``*_dev`` entry points take, and what ``torch.cuda.ExternalStream`` wraps)."""

    def __init__(self):
        self._lib = load_library()
        s = ctypes.c_void_p()
        check(self._lib.dbh_stream_create(ctypes.byref(s)), 'dbh_stream_create')
        self.ptr = s.value

    def synchronize(self):
        check(self._lib.dbh_stream_synchronize(self.ptr), 'dbh_stream_synchronize')

    def wait_event(self, event):
        """Work queued from now on waits for ``event`` (an ``Event`` recorded on another stream)."""
        check(self._lib.dbh_stream_wait_event(self.ptr, event.handle), 'dbh_stream_wait_event')

    def close(self):
        if self.ptr:
            self._lib.dbh_stream_destroy(self.ptr)
            self.ptr = None


class DeviceBuffer:
    """A raw HBM allocation owned through the C ABI (no torch / no other GPU library)."""

    def __init__(self, nbytes):
        self._lib = load_library()
        self.nbytes = int(nbytes)
        ptr = ctypes.c_void_p()
        check(self._lib.dbh_malloc(ctypes.byref(ptr), self.nbytes), 'dbh_malloc')
        self.ptr = ptr.value

    @classmethod
    def from_array(cls, array, stream=None):
        array = np.ascontiguousarray(array)
        buf = cls(max(array.nbytes, 1))
        buf.upload(array, stream)
        return buf

    def upload(self, array, stream=None):
        array = np.ascontiguousarray(array)
        assert array.nbytes <= self.nbytes
        check(self._lib.dbh_memcpy_h2d(self.ptr, array.ctypes.data, array.nbytes, stream),
              'dbh_memcpy_h2d')
        check(self._lib.dbh_stream_synchronize(stream), 'dbh_stream_synchronize')

    def download(self, shape, dtype, stream=None):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(self._lib.dbh_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, stream),
              'dbh_memcpy_d2h')
        check(self._lib.dbh_stream_synchronize(stream), 'dbh_stream_synchronize')
        return out

    def free(self):
        if self.ptr:
            self._lib.dbh_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Event:
    def __init__(self):
        self._lib = load_library()
        e = ctypes.c_void_p()
        check(self._lib.dbh_event_create(ctypes.byref(e)), 'dbh_event_create')
        self.handle = e.value

    def record(self, stream=None):
        check(self._lib.dbh_event_record(self.handle, stream), 'dbh_event_record')

    def synchronize(self):
        check(self._lib.dbh_event_synchronize(self.handle), 'dbh_event_synchronize')

    def elapsed_ms(self, later):
        ms = ctypes.c_float(0)
        check(self._lib.dbh_event_elapsed_ms(self.handle, later.handle, ctypes.byref(ms)),
              'dbh_event_elapsed_ms')
        return ms.value

    def __del__(self):
        try:
            if self.handle:
                self._lib.dbh_event_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _TensorSpec:
    """Minimal stand-in for the Keras tensors load_trained_model looks at (classify.py:93-97)."""

    def __init__(self, shape):
        self.shape = tuple(shape)


class HipModel:
    """A trained Deepbinner model resident on one MI355X."""

    def __init__(self, weights, device=None):
        if not isinstance(weights, ModelWeights):
            raise TypeError('weights must be a ModelWeights')
        self._lib = load_library()
        if device_count() < 1:
            raise HipBackendError('no HIP device is visible - Deepbinner-AMD needs an MI355X '
                                  '(gfx950) GPU; there is no CPU fallback')
        if device is not None:
            set_device(device)
        self.device = get_device()
        self.weights = weights
        flat = weights.flat()
        handle = ctypes.c_void_p()
        check(self._lib.dbh_model_create(flat, flat.size, weights.n_classes, weights.input_size,
                                         ctypes.byref(handle)), 'dbh_model_create')
        self._handle = handle
        self.n_classes = weights.n_classes
        self.input_size = weights.input_size
        self.inputs = [_TensorSpec((None, self.input_size, 1))]
        self.outputs = [_TensorSpec((None, self.n_classes))]

    @property
    def handle(self):
        return self._handle

    def close(self):
        # further queues on the same GPU that realtime.queue_clones made of this model go with it
        for _pair, more in self.__dict__.pop('_queue_clones', []):
            for clones in more:
                for clone in clones:
                    if clone is not None and clone is not self:
                        clone.close()
        if self._handle:
            self._lib.dbh_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- seam b1 ------------------------------------------------------------------------------
    def predict(self, x, batch_size=None, verbose=0):
        """x: [N, 1024, 1] (or [N, 1024]) float -> float32 [N, n_classes], a fresh writable array.
        ``batch_size`` is accepted for signature compatibility; results do not depend on it."""
        x = np.asarray(x)
        if x.ndim == 3:
            if x.shape[2] != 1:
                raise ValueError('expected input of shape (N, {}, 1)'.format(self.input_size))
            x = x[:, :, 0]
        if x.ndim != 2 or x.shape[1] != self.input_size:
            raise ValueError('expected input of shape (N, {}, 1)'.format(self.input_size))
        x = np.ascontiguousarray(x, dtype=np.float32)      # Keras casts float64 -> float32 too
        probs = np.empty((x.shape[0], self.n_classes), dtype=np.float32)
        if x.shape[0]:
            check(self._lib.dbh_predict(self._handle, x, x.shape[0], probs), 'dbh_predict')
        return probs

    # -- seam b2 ------------------------------------------------------------------------------
    def classify_signals(self, signals, side, scan_size, score_diff):
        """signals: list of 1-D integer arrays -> (probs float32 [N, C], calls int32 [N])."""
        n = len(signals)
        scan_size = int(scan_size)
        keep = scan_size + self.input_size // 2      # samples any window can touch
        parts = []
        offsets = np.zeros(n + 1, dtype=np.int64)
        for i, s in enumerate(signals):
            s = np.asarray(s)
            if s.ndim != 1:
                raise ValueError('signals must be one-dimensional')
            if s.dtype != np.int16 and len(s) and (s.min() < -32768 or s.max() > 32767):
                raise ValueError('signal values do not fit int16')
            # only the scanned end of the read is shipped to the GPU; window contents and
            # positions are unchanged by dropping samples no window reaches
            part = s[:keep] if side == 'start' else s[max(len(s) - keep, 0):]
            parts.append(np.asarray(part, dtype=np.int16))
            offsets[i + 1] = offsets[i] + len(part)
        samples = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int16)
        samples = np.ascontiguousarray(samples, dtype=np.int16)
        if samples.size == 0:
            samples = np.zeros(1, dtype=np.int16)
        probs = np.empty((n, self.n_classes), dtype=np.float32)
        calls = np.empty(n, dtype=np.int32)
        if n:
            side_code = SIDE_START if side == 'start' else SIDE_END
            check(self._lib.dbh_classify_i16(self._handle, samples, offsets, n, side_code,
                                             scan_size, float(score_diff), probs, calls),
                  'dbh_classify_i16')
        return probs, calls

    def classify_packed(self, samples, offsets, side, scan_size, score_diff):
        """``classify_signals`` for reads that are already packed the way the C ABI takes them:
        read i is ``samples[offsets[i]:offsets[i+1]]`` (int16, int64).  A long read may have had
        its middle dropped as long as the first and the last ``scan_size + input_size // 2``
        samples are there (what the loaders keep): no window reaches further."""
        samples = np.ascontiguousarray(samples, dtype=np.int16)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        if n < 0 or (n and (offsets[0] != 0 or offsets[-1] != len(samples))):
            raise ValueError('offsets do not describe the sample buffer')
        if samples.size == 0:
            samples = np.zeros(1, dtype=np.int16)
        probs = np.empty((max(n, 0), self.n_classes), dtype=np.float32)
        calls = np.empty(max(n, 0), dtype=np.int32)
        if n > 0:
            check(self._lib.dbh_classify_i16(self._handle, samples, offsets, n,
                                             SIDE_START if side == 'start' else SIDE_END,
                                             int(scan_size), float(score_diff), probs, calls),
                  'dbh_classify_i16')
        return probs, calls

    # -- device-resident entry points (inputs/outputs are raw device pointers) -----------------
    def workspace_bytes(self, n_reads, scan_size):
        n = ctypes.c_size_t(0)
        check(self._lib.dbh_classify_workspace_bytes(self._handle, n_reads, int(scan_size),
                                                     ctypes.byref(n)),
              'dbh_classify_workspace_bytes')
        return n.value

    def classify_dev(self, samples_ptr, offsets_ptr, n_reads, side, scan_size, score_diff,
                     probs_ptr, calls_ptr, workspace_ptr, stream=None):
        check(self._lib.dbh_classify_i16_dev(self._handle, samples_ptr, offsets_ptr, n_reads,
                                             SIDE_START if side == 'start' else SIDE_END,
                                             int(scan_size), float(score_diff), probs_ptr,
                                             calls_ptr, workspace_ptr, stream),
              'dbh_classify_i16_dev')

    def classify_batched_dev(self, samples_ptr, offsets_ptr, n_reads, batch_size, side, scan_size,
                             score_diff, probs_ptr, calls_ptr, stream=None):
        check(self._lib.dbh_classify_i16_batched_dev(
            self._handle, samples_ptr, offsets_ptr, n_reads, int(batch_size),
            SIDE_START if side == 'start' else SIDE_END, int(scan_size), float(score_diff),
            probs_ptr, calls_ptr, stream), 'dbh_classify_i16_batched_dev')

    def predict_dev(self, x_ptr, n_windows, probs_ptr, stream=None):
        check(self._lib.dbh_predict_dev(self._handle, x_ptr, n_windows, probs_ptr, stream),
              'dbh_predict_dev')

    def forward_truncated_dev(self, x_ptr, n_windows, last_stage, stream=None):
        check(self._lib.dbh_forward_truncated_dev(self._handle, x_ptr, n_windows, last_stage,
                                                  stream), 'dbh_forward_truncated_dev')

    def timeline(self, x):
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1, self.input_size))
        out = np.zeros((x.shape[0], 8, 64), dtype=np.int64)
        check(self._lib.dbh_forward_timeline(self._handle, x, x.shape[0], out),
              'dbh_forward_timeline')
        return out

    def set_host_group(self, windows_per_group=0):
        """Windows per model and group of the host-buffer pipeline (0 = the default 32,768)."""
        check(self._lib.dbh_model_set_host_group(self._handle, int(windows_per_group)),
              'dbh_model_set_host_group')

    def reserve_cus(self, n_cus=0):
        """Leave ``n_cus`` CUs out of this model's forward launches (0 = take all again): room
        for the inflate kernels of the containers that follow on other queues."""
        check(self._lib.dbh_model_reserve_cus(self._handle, int(n_cus)), 'dbh_model_reserve_cus')

    def clone(self):
        """The same weights as another model on the same GPU: its own streams and buffers - one
        more queue."""
        return HipModel(self.weights, device=self.device)

    def set_read_length_hint(self, read_length, capacity_samples=0):
        """Tell the ``*_dev`` entry points that every read is ``read_length`` samples long (0
        clears it); the hint is checked on the device, a wrong one only costs time."""
        check(self._lib.dbh_model_set_read_length_hint(self._handle, int(read_length),
                                                       int(capacity_samples)),
              'dbh_model_set_read_length_hint')

    def timeline_i16(self, reads):
        reads = np.ascontiguousarray(np.asarray(reads, dtype=np.int16).reshape(-1, self.input_size))
        out = np.zeros((reads.shape[0], 8, 64), dtype=np.int64)
        check(self._lib.dbh_forward_timeline_i16(self._handle, reads, reads.shape[0], out),
              'dbh_forward_timeline_i16')
        return out

    def timing_enable(self, every_nth=1, span=1):
        """Bracket a run of ``span`` consecutive forward launches with one HIP event pair at
        every n-th launch (0/False = off, True = every launch on its own)."""
        check(self._lib.dbh_forward_timing_enable_span(self._handle, int(every_nth), int(span)),
              'dbh_forward_timing_enable_span')

    def clock_enable(self, on=True):
        """Have production launches of the forward kernel note the shader clock against the wall
        clock (see dbh_forward_clock_enable)."""
        check(self._lib.dbh_forward_clock_enable(self._handle, 1 if on else 0),
              'dbh_forward_clock_enable')

    def phases_enable(self, on=True):
        """Have production launches keep the phase stamps as well (on top of clock_enable): about 1 % of
        the kernel's time, so not for timed runs."""
        check(self._lib.dbh_forward_phases_enable(self._handle, 1 if on else 0), 'dbh_forward_phases_enable')

    def phases_read(self):
        """Mean shader cycles of the five phases of a group of windows in this model's latest forward
        launch (see dbh_forward_phases_read): stages A-C, the stage D-E chain, stage F, the batched
        tail, what lies between two groups; and the number of groups averaged."""
        out = (ctypes.c_double * 14)()
        n = ctypes.c_int64(0)
        check(self._lib.dbh_forward_phases_read(self._handle, out, ctypes.byref(n)), 'dbh_forward_phases_read')
        return [float(v) for v in out], int(n.value)

    def clock_read(self):
        """Shader clock (GHz) during this model's latest forward launch."""
        ghz = ctypes.c_double(0)
        check(self._lib.dbh_forward_clock_read(self._handle, ctypes.byref(ghz)),
              'dbh_forward_clock_read')
        return ghz.value

    def timing_read(self):
        ms, launches, windows = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_int64(0)
        check(self._lib.dbh_forward_timing_read(self._handle, ctypes.byref(ms),
                                                ctypes.byref(launches), ctypes.byref(windows)),
              'dbh_forward_timing_read')
        return ms.value, launches.value, windows.value

    # -- introspection ------------------------------------------------------------------------
    def debug_stage(self, x, stage):
        """Activations after stage 'A'..'G' (or 'logits') for windows x [N, 1024]."""
        idx = {'A': 0, 'B': 1, 'C': 2, 'D': 3, 'E': 4, 'F': 5, 'G': 6, 'logits': 7}[stage]
        x = np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(-1, self.input_size))
        per = ctypes.c_int64(0)
        check(self._lib.dbh_stage_floats(idx, ctypes.byref(per)), 'dbh_stage_floats')
        out = np.empty((x.shape[0], per.value), dtype=np.float32)
        check(self._lib.dbh_debug_forward(self._handle, x, x.shape[0], idx, out),
              'dbh_debug_forward')
        shapes = {0: (512, 48), 1: (256, 48), 2: (128, 48), 3: (64, 48), 4: (32, 192),
                  5: (16, 48), 6: (8, 48), 7: (32,)}
        return out.reshape((x.shape[0],) + shapes[idx])


def forward_executed_mfmas(n_classes):
    """(MFMA instructions, FLOP) the forward kernel issues per window."""
    lib = load_library()
    n, f = ctypes.c_int64(0), ctypes.c_int64(0)
    check(lib.dbh_forward_executed_mfmas(int(n_classes), ctypes.byref(n), ctypes.byref(f)),
          'dbh_forward_executed_mfmas')
    return n.value, f.value


def forward_kernel_info():
    lib = load_library()
    t, l, v = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    check(lib.dbh_forward_kernel_info(ctypes.byref(t), ctypes.byref(l), ctypes.byref(v)),
          'dbh_forward_kernel_info')
    return {'threads_per_block': t.value, 'lds_bytes': l.value, 'vgprs': v.value}

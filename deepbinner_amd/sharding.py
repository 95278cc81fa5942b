"""
Multi-GPU layer of the classify path.  Reads are independent, so they shard across the GPUs of a
node with no data-path collective; the only exchange is an all-gather of per-read int32 barcode
calls, done by RCCL over xGMI behind the C ABI (``dbh_comm_*`` in
``include/deepbinner_hip.h``).  No torch anywhere: device memory, streams and the collective all
go through ``libdeepbinner_hip.so``; what little host-side coordination a multi-process launch
needs (shipping RCCL's 128-byte unique id, barriers, the MAX of the ranks' timings) runs over a
socket rendezvous of its own.

Two host models:

* **one process, N devices** (``DeviceGroup``): a worker thread per device, each with its own
  model replica and stream; ``dbh_comm_init_all`` (``ncclCommInitAll``) and one grouped
  all-gather per exchange.  This is what ``python bench.py --gpus N`` and
  ``deepbinner classify --devices N`` use.
* **one process per GPU** (``RankGroup``), for launchers that start one rank per GPU
  (``python -m torch.distributed.run ... bench.py``): RANK / LOCAL_RANK / WORLD_SIZE come from
  the environment, rank 0 creates the RCCL unique id and the ``Rendezvous`` broadcasts it.

The reference has no counterpart (single process, single device, ``classify.py:416-423``).
"""

import ctypes
import os
import socket
import struct
import sys
import threading
import time

import numpy as np


def shard_bounds(n_items, world_size, rank):
    """Contiguous block of ``n_items`` owned by ``rank``: sizes differ by at most one, earlier
    ranks take the remainder, concatenating the blocks in rank order restores the input order."""
    base, extra = divmod(int(n_items), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def env_world():
    """(rank, local_rank, world_size) from the launcher's environment; (0, 0, 1) when absent."""
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)),
            int(os.environ.get('WORLD_SIZE', 1)))


# -------------------------------------------------------------------------------------------------
# Host-side rendezvous for one-process-per-GPU launches
# -------------------------------------------------------------------------------------------------

class RendezvousError(RuntimeError):
    pass


def _rendezvous_endpoint():
    """Where the ranks of one launch meet: ('unix', name) - an abstract unix socket, which needs
    no file and disappears with rank 0 - or ('tcp', host, port).  All ranks must arrive at the same
    answer and no other launch at it, from nothing but their environment:

    * under a launcher (MASTER_PORT set: ``torch.distributed.run`` and the like) the name is made
      of MASTER_ADDR, MASTER_PORT and the run id - the port belongs to the launcher's own store, so
      it cannot be bound again, but it is unique per launch on this host.  (Not the parent's pid:
      a launcher that wraps every rank in a shell of its own gives every rank another parent.)
    * ranks started by hand from one shell (no MASTER_PORT) share that shell's pid;
    * ``DEEPBINNER_RDZV=<name>`` names the socket outright; ``DEEPBINNER_RDZV=tcp`` meets on TCP
      port MASTER_PORT + 1 of MASTER_ADDR instead (ranks that do not share a network namespace),
      ``DEEPBINNER_RDZV=tcp://host:port`` on that address."""
    explicit = os.environ.get('DEEPBINNER_RDZV', '')
    if explicit.startswith('tcp://'):
        host, _, port = explicit[6:].rpartition(':')
        return ('tcp', host or '127.0.0.1', int(port))
    if explicit == 'tcp':
        return ('tcp', os.environ.get('MASTER_ADDR', '127.0.0.1'),
                int(os.environ.get('MASTER_PORT', '29500')) + 1)
    if explicit:
        return ('unix', explicit)
    if os.environ.get('MASTER_PORT'):
        return ('unix', 'deepbinner-{}-{}-{}'.format(os.environ.get('MASTER_ADDR', 'localhost'),
                                                     os.environ['MASTER_PORT'],
                                                     os.environ.get('TORCHELASTIC_RUN_ID', 'none')))
    return ('unix', 'deepbinner-by-hand-{}'.format(os.getppid()))


MAX_MESSAGE_BYTES = 1 << 30        # nothing the ranks tell each other comes near it


def _send_msg(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv_exact(sock, n):
    chunks, got = [], 0
    while got < n:
        chunk = sock.recv(min(n - got, 1 << 20))
        if not chunk:
            raise RendezvousError('a rank closed its rendezvous connection')
        chunks.append(chunk)
        got += len(chunk)
    return b''.join(chunks)


def _recv_msg(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    if n > MAX_MESSAGE_BYTES:
        raise RendezvousError('rendezvous: a peer announced a message of {} bytes'.format(n))
    return _recv_exact(sock, n)


class Rendezvous:
    """A star of stream sockets around rank 0 with one primitive, ``all_gather(bytes) ->
    [bytes per rank]``; barrier, broadcast and the MAX of a float are built on it.  Every rank
    calls the same sequence of operations (as with any collective)."""

    def __init__(self, rank, world, name=None, timeout=None):
        self.rank, self.world = int(rank), int(world)
        self.timeout = float(timeout if timeout is not None
                             else os.environ.get('DEEPBINNER_RDZV_TIMEOUT', 120))
        self._peers = {}
        self._sock = None
        if self.world == 1:
            return
        if not 0 <= self.rank < self.world:
            raise RendezvousError('rendezvous: rank {} of {}'.format(self.rank, self.world))
        endpoint = ('unix', name) if name else _rendezvous_endpoint()
        if endpoint[0] == 'unix':
            family, address = socket.AF_UNIX, '\0' + endpoint[1]
        else:
            family, address = socket.AF_INET, (endpoint[1], endpoint[2])
        self.endpoint = endpoint
        if self.rank == 0:
            server = socket.socket(family, socket.SOCK_STREAM)
            if family == socket.AF_INET:
                server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                server.bind(address)
            except OSError as e:
                server.close()
                raise RendezvousError('rendezvous: cannot bind {} ({})'.format(endpoint[1:], e))
            server.listen(self.world)
            deadline = time.monotonic() + self.timeout
            try:
                while len(self._peers) < self.world - 1:
                    server.settimeout(max(deadline - time.monotonic(), 0.001))
                    conn, _ = server.accept()
                    conn.settimeout(self.timeout)
                    (peer,) = struct.unpack('<I', _recv_exact(conn, 4))
                    # (a stray connection, or two ranks that both think they are rank k, must
                    # not pass for the missing rank: say so now, not as a KeyError later)
                    if not 0 < peer < self.world or peer in self._peers:
                        conn.close()
                        raise RendezvousError(
                            'rendezvous: a peer introduced itself as rank {} ({} ranks; seen so '
                            'far: {})'.format(peer, self.world, sorted(self._peers)))
                    self._peers[peer] = conn
            except socket.timeout:
                raise RendezvousError('rendezvous: only {} of {} ranks arrived within {:.0f} s '
                                      '(at {}; DEEPBINNER_RDZV_TIMEOUT sets the wait)'
                                      .format(len(self._peers) + 1, self.world, self.timeout,
                                              endpoint[1:]))
            finally:
                server.close()
        else:
            deadline = time.monotonic() + self.timeout
            while True:
                sock = socket.socket(family, socket.SOCK_STREAM)
                try:
                    sock.connect(address)
                    break
                except (ConnectionRefusedError, FileNotFoundError, OSError):
                    sock.close()
                    if time.monotonic() > deadline:
                        raise RendezvousError('rendezvous: rank 0 did not appear within {:.0f} s '
                                              '(at {})'.format(self.timeout, endpoint[1:]))
                    time.sleep(0.02)
            sock.settimeout(self.timeout)
            sock.sendall(struct.pack('<I', self.rank))
            self._sock = sock

    @classmethod
    def from_env(cls, **kwargs):
        rank, _, world = env_world()
        return cls(rank, world, **kwargs)

    def all_gather(self, payload=b''):
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        try:
            if self.rank == 0:
                parts = [payload] + [_recv_msg(self._peers[r]) for r in range(1, self.world)]
                blob = b''.join(struct.pack('<Q', len(p)) + p for p in parts)
                for r in range(1, self.world):
                    _send_msg(self._peers[r], blob)
                return parts
            _send_msg(self._sock, payload)
            blob = _recv_msg(self._sock)
        except socket.timeout:
            raise RendezvousError('rendezvous: a rank did not answer within {:.0f} s'
                                  .format(self.timeout))
        parts, at = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from('<Q', blob, at)
            parts.append(blob[at + 8:at + 8 + n])
            at += 8 + n
        return parts

    def barrier(self):
        self.all_gather(b'')

    def broadcast(self, payload=None, root=0):
        return self.all_gather(payload if self.rank == root else b'')[root]

    def max_float(self, value):
        return max(struct.unpack('<d', p)[0] for p in self.all_gather(struct.pack('<d', value)))

    def agree(self, ok, message=''):
        """Everyone learns whether everyone is fine: -> (all_ok, first failing rank's message)."""
        parts = self.all_gather((b'\1' if ok else b'\0') + message.encode())
        for p in parts:
            if p[:1] != b'\1':
                return False, p[1:].decode()
        return True, ''

    def close(self):
        for conn in self._peers.values():
            conn.close()
        if self._sock is not None:
            self._sock.close()
        self._peers, self._sock = {}, None


# -------------------------------------------------------------------------------------------------
# The exchange itself
# -------------------------------------------------------------------------------------------------

TRANSPORT_RCCL, TRANSPORT_COPY = 0, 1


def _pointer_array(values):
    return (ctypes.c_void_p * len(values))(*[ctypes.c_void_p(v) for v in values])


class Communicator:
    """A ``dbh_comm``: RCCL all-gather of int32 blocks between devices, one entry per LOCAL device
    in every argument list (all of them in the single-process form, one in the per-rank form)."""

    def __init__(self, handle, n_ranks, n_local, transport):
        from . import hip_backend
        self._lib = hip_backend.load_library()
        self._check = hip_backend.check
        self._handle = handle
        self.n_ranks, self.n_local, self.transport = n_ranks, n_local, transport

    @classmethod
    def init_all(cls, devices, transport=TRANSPORT_RCCL):
        """One process, ``devices`` = list of HIP ordinals (``ncclCommInitAll``)."""
        from . import hip_backend
        lib = hip_backend.load_library()
        ordinals = (ctypes.c_int * len(devices))(*devices)
        handle = ctypes.c_void_p()
        hip_backend.check(lib.dbh_comm_init_all(len(devices), ordinals, transport,
                                                ctypes.byref(handle)), 'dbh_comm_init_all')
        return cls(handle, len(devices), len(devices), transport)

    @classmethod
    def init_rank(cls, rendezvous):
        """One process per GPU: rank 0's unique id reaches the others through the rendezvous;
        the calling thread's current device is the rank's GPU.  All ranks learn whether all of
        them succeeded (-> a Communicator everywhere, or a HipBackendError everywhere)."""
        from . import hip_backend
        lib = hip_backend.load_library()
        uid = ctypes.create_string_buffer(128)
        status, detail = 0, ''
        if rendezvous.rank == 0:
            status = lib.dbh_comm_unique_id(uid)
            detail = lib.dbh_comm_last_error().decode()
        blob = rendezvous.broadcast(bytes([status]) + uid.raw)
        handle = ctypes.c_void_p()
        if blob[0] == 0:
            status = lib.dbh_comm_init_rank(blob[1:129], rendezvous.world, rendezvous.rank,
                                            ctypes.byref(handle))
            detail = lib.dbh_comm_last_error().decode()
        else:
            status = blob[0]
        ok, why = rendezvous.agree(status == 0, 'rank {}: {}'.format(rendezvous.rank, detail))
        if not ok:
            if handle:
                lib.dbh_comm_destroy(handle)
            raise hip_backend.HipBackendError('RCCL communicator could not be set up ({})'
                                              .format(why))
        return cls(handle, rendezvous.world, 1, TRANSPORT_RCCL)

    def all_gather_i32(self, send_ptrs, recv_ptrs, count, streams):
        self._check(self._lib.dbh_comm_all_gather_i32(
            self._handle, _pointer_array(send_ptrs), _pointer_array(recv_ptrs), int(count),
            _pointer_array(streams)), 'dbh_comm_all_gather_i32')

    def close(self):
        if self._handle:
            self._lib.dbh_comm_destroy(self._handle)
            self._handle = None


def transport_from_env():
    """DEEPBINNER_COMM = rccl (default) | copy (device copies, single-process form) | host (the
    calls travel through host memory: the fallback when RCCL cannot be set up, and the test
    mode of boxes whose "devices" are all the same GPU)."""
    kind = os.environ.get('DEEPBINNER_COMM', 'rccl').lower()
    if kind not in ('rccl', 'copy', 'host'):
        raise ValueError('DEEPBINNER_COMM must be rccl, copy or host')
    return kind


def devices_from_env(n_devices):
    """HIP ordinals of the ``n_devices`` "devices" of a single-process run: 0..n-1, or - test mode
    for one-GPU boxes - DEEPBINNER_DEVICE_ORDINALS=0,0 to run several shards on the same GPU."""
    explicit = os.environ.get('DEEPBINNER_DEVICE_ORDINALS')
    if explicit:
        ordinals = [int(v) for v in explicit.split(',')]
        if len(ordinals) != n_devices:
            raise ValueError('DEEPBINNER_DEVICE_ORDINALS names {} devices, {} wanted'
                             .format(len(ordinals), n_devices))
        return ordinals
    return list(range(n_devices))


class DeviceShard:
    """One device's share of a sharded job: a model replica, a stream, and the device buffers of
    the reads it owns.  Every method must be called with the shard's device current on the
    calling thread (``DeviceGroup`` runs each shard on a thread of its own that does that once)."""

    def __init__(self, weights, device, rank=0, n_ranks=1):
        from . import hip_backend
        self.hip = hip_backend
        self.device, self.rank, self.n_ranks = device, rank, n_ranks
        hip_backend.set_device(device)
        self.model = hip_backend.HipModel(weights)
        self.extra_models = []
        self.stream = hip_backend.Stream()
        self.n_reads = 0
        self.samples = self.offsets = self.probs = self.calls = self.gathered = None
        self.block = 0

    def add_model(self, weights):
        self.hip.set_device(self.device)
        self.extra_models.append(self.hip.HipModel(weights))
        return self.extra_models[-1]

    def upload(self, samples, offsets, block):
        """This shard's reads (packed int16 + int64 offsets starting at 0) become resident;
        ``block`` is the per-rank length of the gathered call array (the longest shard)."""
        hip = self.hip
        self.n_reads = len(offsets) - 1
        self.block = int(block)
        self.samples = hip.DeviceBuffer.from_array(samples if len(samples) else
                                                   np.zeros(1, np.int16))
        self.offsets = hip.DeviceBuffer.from_array(np.asarray(offsets, dtype=np.int64))
        self.probs = hip.DeviceBuffer(max(self.n_reads, 1) * self.model.n_classes * 4)
        self.calls = hip.DeviceBuffer(max(self.block, 1) * 4)
        # a job of one device has nothing to exchange: its calls ARE the gathered calls, unless
        # a communicator is there to be exercised (see DeviceGroup / RankGroup.all_gather)
        self.gathered = (hip.DeviceBuffer(max(self.block, 1) * self.n_ranks * 4)
                         if self.n_ranks > 1 or os.environ.get('DEEPBINNER_COMM_FORCE') == '1'
                         else self.calls)
        # the padding of a short shard must not be garbage: it is gathered too
        zeros = np.zeros(max(self.block, 1), dtype=np.int32)
        self.calls.upload(zeros)

    def classify(self, batch_size, side, scan_size, score_diff, model=None, calls_ptr=None,
                 probs_ptr=None):
        (model or self.model).classify_batched_dev(
            self.samples.ptr, self.offsets.ptr, self.n_reads, batch_size, side, scan_size,
            score_diff, probs_ptr or self.probs.ptr, calls_ptr or self.calls.ptr,
            self.stream.ptr)

    def synchronize(self):
        self.stream.synchronize()

    def gathered_calls(self, shard_sizes):
        """The calls of the whole job in read order, from THIS device's gathered buffer."""
        flat = self.gathered.download((self.n_ranks * max(self.block, 1),), np.int32,
                                      self.stream.ptr)
        pieces = [flat[r * self.block:r * self.block + n] for r, n in enumerate(shard_sizes)]
        return np.concatenate(pieces) if pieces else np.zeros(0, np.int32)


class SideGather:
    """The exchange of a shard's calls on a stream of its own, so that the classification stream
    carries nothing but forward launches: two call arrays and two gathered arrays used in turn
    (``slot`` = step & 1), an event from the classification stream to the side stream when a
    step's calls are final, and one back when the side stream has finished with a slot's arrays
    (the step after next writes them again).  A collective or a copy command queued BETWEEN two
    launches of one stream costs ~20 us of idle GPU each time (tools/step_gap.py), 1 % of a
    10,000-read step, before the collective's own latency.  Create and use on the shard's thread."""

    def __init__(self, shard):
        hip = shard.hip
        self.shard = shard
        self.stream = hip.Stream()
        block = max(shard.block, 1)
        own_gathered = shard.gathered is not shard.calls
        self.calls = [shard.calls, hip.DeviceBuffer(block * 4)]
        self.gathered = [shard.gathered if own_gathered else hip.DeviceBuffer(block * shard.n_ranks * 4),
                         hip.DeviceBuffer(block * shard.n_ranks * 4)]
        self.calls[1].upload(np.zeros(block, dtype=np.int32))      # (a short shard's padding)
        self.final = [hip.Event(), hip.Event()]      # slot's calls are final (classification stream)
        self.released = [hip.Event(), hip.Event()]   # slot's arrays are free again (side stream)
        self.in_use = [False, False]

    def before_classify(self, slot):
        if self.in_use[slot]:
            self.shard.stream.wait_event(self.released[slot])

    def after_classify(self, slot):
        self.final[slot].record(self.shard.stream.ptr)
        self.stream.wait_event(self.final[slot])

    def release(self, slot):
        """Call once everything that reads the slot's arrays has been queued on the side stream."""
        self.released[slot].record(self.stream.ptr)
        self.in_use[slot] = True

    def synchronize(self):
        self.stream.synchronize()


class DeviceGroup:
    """One process, N devices: a long-lived worker thread per device (HIP's current device is a
    per-thread setting, and the C ABI releases the GIL, so the devices' launch queues fill side
    by side).  ``run(fn)`` calls ``fn(shard)`` on every device's thread and returns the results
    in device order; ``all_gather()`` queues the exchange of the shards' call buffers."""

    def __init__(self, weights, n_devices, devices=None, transport=None):
        from . import hip_backend
        self.hip = hip_backend
        self.devices = list(devices) if devices is not None else devices_from_env(n_devices)
        self.n = len(self.devices)
        if self.n < 1:
            raise ValueError('at least one device')
        visible = hip_backend.device_count()
        if max(self.devices) >= visible:
            raise hip_backend.HipBackendError(
                '{} devices wanted, {} visible'.format(max(self.devices) + 1, visible))
        kind = transport or transport_from_env()
        if kind == 'rccl' and len(set(self.devices)) < self.n:
            kind = 'copy'               # RCCL refuses two ranks on one GPU
        self.transport = kind
        self._threads, self._inbox, self._outbox = [], [], []
        import queue
        if self.n == 1:                  # nothing to run side by side: stay on the caller's thread
            hip_backend.set_device(self.devices[0])
        for i in range(self.n if self.n > 1 else 0):
            self._inbox.append(queue.Queue())
            self._outbox.append(queue.Queue())
            t = threading.Thread(target=self._worker, args=(i,), daemon=True,
                                 name='deepbinner-device-{}'.format(self.devices[i]))
            t.start()
            self._threads.append(t)
        self.shards = self.run_indexed(
            lambda i: DeviceShard(weights, self.devices[i], rank=i, n_ranks=self.n))
        self.comm = None
        self.fallback_reason = None
        forced = os.environ.get('DEEPBINNER_COMM_FORCE') == '1'     # (test: RCCL with one device)
        if (self.n > 1 or forced) and kind != 'host':
            try:
                self.comm = Communicator.init_all(
                    self.devices, TRANSPORT_RCCL if kind == 'rccl' else TRANSPORT_COPY)
            except hip_backend.HipBackendError as e:
                # never silently: the caller reports which transport carried the calls
                self.fallback_reason = str(e)
                print('deepbinner: RCCL unavailable ({}); gathering calls through host memory'
                      .format(e), file=sys.stderr)
                self.transport = 'host'
        self.shard_sizes = [0] * self.n
        self._host_calls = None

    def _worker(self, i):
        self.hip.set_device(self.devices[i])
        while True:
            job = self._inbox[i].get()
            if job is None:
                return
            try:
                self._outbox[i].put((True, job()))
            except BaseException as e:          # delivered to the caller of run()
                self._outbox[i].put((False, e))

    def run_on(self, i, fn):
        """fn() on device i's thread."""
        if self.n == 1:
            return fn()
        self._inbox[i].put(fn)
        ok, value = self._outbox[i].get()
        if not ok:
            raise value
        return value

    def run_indexed(self, fn):
        if self.n == 1:
            return [fn(0)]
        for i in range(self.n):
            self._inbox[i].put(lambda i=i: fn(i))
        results = [self._outbox[i].get() for i in range(self.n)]
        for ok, value in results:
            if not ok:
                raise value
        return [value for _, value in results]

    def run(self, fn):
        return self.run_indexed(lambda i: fn(self.shards[i]))

    def upload_sharded(self, samples, offsets):
        """Contiguous read shards of one packed job (``shard_bounds``) become resident, one per
        device."""
        offsets = np.asarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        bounds = [shard_bounds(n, self.n, r) for r in range(self.n)]
        self.shard_sizes = [b - a for a, b in bounds]
        block = max(self.shard_sizes + [1])

        def put(i):
            a, b = bounds[i]
            self.shards[i].upload(samples[offsets[a]:offsets[b]], offsets[a:b + 1] - offsets[a],
                                  block)
        self.run_indexed(put)

    def all_gather(self):
        """Queue the exchange behind each device's classification (RCCL / device copies), or - in
        host mode - bring every shard's calls to the host and hand all of them back."""
        if self.n == 1 and self.comm is None:
            return
        if self.comm is not None:
            self.comm.all_gather_i32([s.calls.ptr for s in self.shards],
                                     [s.gathered.ptr for s in self.shards],
                                     self.shards[0].block, [s.stream.ptr for s in self.shards])
            return
        parts = self.run(lambda s: s.calls.download((s.block,), np.int32, s.stream.ptr))
        whole = np.concatenate(parts)
        self.run(lambda s: s.gathered.upload(whole, s.stream.ptr))

    def all_gather_side(self, sides, slot):
        """The same exchange between the ``SideGather`` arrays of slot ``slot``, queued on the
        shards' side streams (RCCL / device copies only)."""
        self.comm.all_gather_i32([g.calls[slot].ptr for g in sides],
                                 [g.gathered[slot].ptr for g in sides],
                                 self.shards[0].block, [g.stream.ptr for g in sides])

    def synchronize(self):
        self.run(lambda s: s.synchronize())

    def gathered_calls(self, device_index=0):
        return self.run_on(device_index,
                           lambda: self.shards[device_index].gathered_calls(self.shard_sizes))

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        for q in self._inbox:
            q.put(None)


class RankGroup:
    """One process per GPU: this rank's ``DeviceShard`` plus the communicator that joins it to the
    other ranks.  ``DEEPBINNER_COMM=host`` (or an RCCL set-up failure, reported on stderr and in
    ``transport``) sends the calls through host memory and the rendezvous instead."""

    def __init__(self, weights, rendezvous, device=None):
        from . import hip_backend
        self.hip = hip_backend
        self.rdzv = rendezvous
        rank, local_rank, world = env_world()
        self.rank, self.world = rendezvous.rank, rendezvous.world
        if device is None:
            device = local_rank
            if os.environ.get('DEEPBINNER_DEVICE_ORDINALS'):       # one-GPU test boxes
                device = devices_from_env(self.world)[self.rank]
        self.shard = DeviceShard(weights, device, rank=self.rank, n_ranks=self.world)
        self.transport = transport_from_env()
        if self.transport == 'copy':
            self.transport = 'host'          # device copies need one process
        self.comm = None
        self.fallback_reason = None
        forced = os.environ.get('DEEPBINNER_COMM_FORCE') == '1'     # (test: RCCL with one rank)
        if (self.world > 1 or forced) and self.transport == 'rccl':
            try:
                self.comm = Communicator.init_rank(rendezvous)
            except hip_backend.HipBackendError as e:
                self.fallback_reason = str(e)
                if self.rank == 0:
                    print('deepbinner: {}; gathering calls through host memory'.format(e),
                          file=sys.stderr)
                self.transport = 'host'
        self.shard_sizes = [0] * self.world

    def upload(self, samples, offsets, shard_sizes):
        self.shard_sizes = list(shard_sizes)
        self.shard.upload(samples, offsets, max(self.shard_sizes + [1]))

    def all_gather(self):
        s = self.shard
        if self.world == 1 and self.comm is None:
            return
        if self.comm is not None:
            self.comm.all_gather_i32([s.calls.ptr], [s.gathered.ptr], s.block, [s.stream.ptr])
        else:
            mine = s.calls.download((s.block,), np.int32, s.stream.ptr)
            parts = self.rdzv.all_gather(mine.tobytes())
            s.gathered.upload(np.frombuffer(b''.join(parts), dtype=np.int32), s.stream.ptr)

    def all_gather_side(self, sides, slot):
        g = sides[0]
        self.comm.all_gather_i32([g.calls[slot].ptr], [g.gathered[slot].ptr], self.shard.block,
                                 [g.stream.ptr])

    def gathered_calls(self):
        return self.shard.gathered_calls(self.shard_sizes)

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None


# -------------------------------------------------------------------------------------------------
# `deepbinner classify DIR` with one process per GPU
# -------------------------------------------------------------------------------------------------

def classify_fast5_files_sharded(fast5_files, start_model, start_input_size, end_model,
                                 end_input_size, output_size, args):
    """``deepbinner classify DIR`` started once per GPU by a launcher (RANK / LOCAL_RANK /
    WORLD_SIZE in the environment): every rank - its model(s) already on its own device -
    classifies a contiguous shard of the sorted file list with the ordinary per-batch loop and
    prints its own table rows when its turn comes (rank order = file order; the text never
    travels); the per-read int32 calls are all-gathered (RCCL between the devices, or through
    the rendezvous with DEEPBINNER_COMM=host) so that rank 0 can print the summary and return
    what the single-process path returns (reference classify.py:106-180); the other ranks return
    ``({}, {})``.  A fatal error on any rank (the reference's ``sys.exit('Error: ...')`` cases)
    is agreed on before any collective, so that all ranks leave together with that message."""
    from . import classify as c
    from .load_fast5s import determine_single_or_multi_fast5s
    from .misc import print_summary_table

    rank, local_rank, world = env_world()
    rdzv = Rendezvous(rank, world)

    def together(fn):
        """Run fn on this rank; if it fails anywhere - the reference's ``sys.exit('Error: ...')``
        cases, or anything else a rank can die of (a HIP error, an unreadable directory) - every
        rank learns of it here, before the next collective, and leaves with the first message
        instead of waiting out the rendezvous timeout for a peer that is gone."""
        result, failure, reraise = None, None, None
        try:
            result = fn()
        except SystemExit as e:
            failure = str(e.code) if e.code is not None else 'exit'
        except BaseException as e:       # noqa: B902 - KeyboardInterrupt included: all must leave
            failure = 'Error: rank {} failed: {}: {}'.format(rank, type(e).__name__, e)
            reraise = e
        ok, why = rdzv.agree(failure is None, failure or '')
        if reraise is not None:
            raise reraise
        if not ok:
            sys.exit(why)
        return result

    def checked_files():
        if not fast5_files:
            sys.exit('Error: no fast5 files found')
        files = sorted(fast5_files)             # os.walk order may differ between processes
        # the reference samples five files at random; every rank must look at the same five
        if rank == 0 and determine_single_or_multi_fast5s(files) == 'multi':
            sys.exit('Error: deepbinner classify requires one-read-per-file fast5s - convert '
                     'with multi_to_single_fast5 before running')
        return files

    files = together(checked_files)
    a, b = shard_bounds(len(files), world, rank)
    mine = files[a:b]

    if rank == 0:
        c.print_classification_progress(0, len(files), 'fast5s')
        c.print_output_header(args.verbose, start_model is not None, end_model is not None,
                              output_size)
        sys.stdout.flush()
    classifications, id_to_file, lines = {}, {}, []

    def classify_shard():
        for loaded in c.load_in_batches(mine, args):
            read_ids, signals = [], []
            for fast5_file, read_id, signal in loaded:
                if signal is None:
                    continue
                id_to_file[read_id] = fast5_file
                read_ids.append(read_id)
                signals.append(signal)
            if getattr(loaded, 'complete', False):      # the loader's packed buffer is these reads
                signals = c.PackedSignals(signals, loaded.samples, loaded.offsets)
            lines.extend(c.classify_read_batch(read_ids, signals, start_model, start_input_size,
                                               end_model, end_input_size, output_size, args,
                                               classifications))
            if rank == 0:    # rank 0's shard is as large as any: its progress stands for the job
                c.print_classification_progress(min(len(classifications) * world, len(files)),
                                                len(files), 'fast5s')

    together(classify_shard)

    # the table: every rank writes its own rows when the ranks before it are done
    for turn in range(world):
        if turn == rank:
            for line in lines:
                print(line)
            sys.stdout.flush()
        rdzv.barrier()

    # the collective of SURVEY.md section 8e: per-read calls, int32, 0 = 'none'
    order = list(classifications)
    local = np.array([0 if classifications[r] == 'none' else int(classifications[r])
                      for r in order], dtype=np.int32)
    counts = [struct.unpack('<q', p)[0] for p in rdzv.all_gather(struct.pack('<q', len(order)))]
    all_calls = gather_calls(local, counts, rdzv)
    id_blobs = rdzv.all_gather('\n'.join('{}\t{}'.format(r, id_to_file[r]) for r in order)
                               .encode() if rank != 0 else b'')
    result = ({}, {})
    if rank == 0:
        merged, merged_files = {}, dict(id_to_file)
        at = 0
        for r in range(world):
            ids = order if r == 0 else [l.split('\t')[0] for l in id_blobs[r].decode().split('\n') if l]
            if r != 0:
                merged_files.update(l.split('\t', 1) for l in id_blobs[r].decode().split('\n') if l)
            for k, read_id in enumerate(ids):
                call = int(all_calls[at + k])
                merged[read_id] = 'none' if call == 0 else str(call)
            at += counts[r]
        c.print_classification_progress(len(merged), len(files), 'fast5s')
        print('', file=sys.stderr)
        print_summary_table(merged)
        result = (merged, merged_files)
    rdzv.barrier()
    rdzv.close()
    return result


def gather_calls(local_calls, counts, rendezvous):
    """All ranks' int32 calls in rank order.  Between GPUs: device buffers + RCCL all-gather
    (blocks padded to the longest shard); with DEEPBINNER_COMM=host, or where no GPU library can
    be loaded (CPU tests of the host logic), through the rendezvous."""
    local_calls = np.ascontiguousarray(local_calls, dtype=np.int32)
    world, rank = rendezvous.world, rendezvous.rank
    if world == 1:
        return local_calls
    use_device = transport_from_env() == 'rccl'
    if use_device:
        try:
            from . import hip_backend
            use_device = hip_backend.device_count() > 0
        except Exception:
            use_device = False
    ok, _ = rendezvous.agree(use_device)
    if ok:
        from . import hip_backend
        try:
            comm = Communicator.init_rank(rendezvous)
        except hip_backend.HipBackendError as e:
            if rank == 0:
                print('deepbinner: {}; gathering calls through host memory'.format(e),
                      file=sys.stderr)
            comm = None
        if comm is not None:
            block = max(max(counts), 1)
            padded = np.zeros(block, dtype=np.int32)
            padded[:len(local_calls)] = local_calls
            stream = hip_backend.Stream()
            send = hip_backend.DeviceBuffer.from_array(padded, stream.ptr)
            recv = hip_backend.DeviceBuffer(block * world * 4)
            comm.all_gather_i32([send.ptr], [recv.ptr], block, [stream.ptr])
            flat = recv.download((world * block,), np.int32, stream.ptr)
            comm.close()
            stream.close()
            return np.concatenate([flat[r * block:r * block + counts[r]] for r in range(world)])
    parts = rendezvous.all_gather(local_calls.tobytes())
    return np.concatenate([np.frombuffer(p, dtype=np.int32) for p in parts])

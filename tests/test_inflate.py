"""
The inflate that runs on the GPU (deepbinner_amd/csrc/dbh_inflate_core.h + dbh_inflate.hip) held
to zlib - Python's zlib module, i.e. the library libhdf5 inflates fast5 Signal chunks with
(reference: load_fast5s.py:33-43 through h5py).  Bit-exact bytes for every valid stream, and the
same accept / reject decision for damaged ones.

CPU part: the decoder core compiled for the host (oracle/_build/inflate_host_test, built by
oracle/Makefile from the very header the kernels use), one lane at a time.
GPU part (`-m gpu`): the kernels through the C ABI (dbh_inflate), whole batches of streams.
"""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import GOLD, REPO

HARNESS = os.path.join(REPO, 'oracle', '_build', 'inflate_host_test')


def squiggle(rng, n):
    levels = np.repeat(rng.normal(450, 80, n // 8 + 1), 8)[:n]
    return np.clip(np.rint(levels + rng.normal(0, 8, n)), 0, 2047).astype('<i2').tobytes()


def raw_deflate_stream(blocks):
    """A zlib stream assembled by hand from (kind, payload) blocks: 'stored' bytes, or 'fixed' /
    'dynamic' bytes compressed by zlib with the matching strategy - to get block types and
    sequences a one-shot compress() never produces."""
    out = bytearray(b'\x78\x01')
    adler = 1
    bits, nbits = 0, 0
    data_all = b''
    body = bytearray()
    # simplest faithful way: let zlib do the bit packing with flush points between blocks
    comp = None
    for kind, payload in blocks:
        strategy = zlib.Z_FIXED if kind == 'fixed' else zlib.Z_DEFAULT_STRATEGY
        level = 0 if kind == 'stored' else 6
        if comp is None:
            comp = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
            first = comp.compress(payload) + comp.flush(zlib.Z_FULL_FLUSH)
            body += first
        else:
            c2 = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
            body += c2.compress(payload) + c2.flush(zlib.Z_FULL_FLUSH)
        data_all += payload
    # terminate: an empty final stored block, then the Adler-32 of everything
    body += b'\x01\x00\x00\xff\xff'
    del out, adler, bits, nbits
    return bytes(body) + struct.pack('>I', zlib.adler32(data_all)), data_all


def valid_cases():
    rng = np.random.default_rng(7)
    cases = []
    reads = np.load(os.path.join(GOLD, 'reads.npz'))
    offsets = reads['offsets']
    real = [reads['samples'][offsets[i]:offsets[i + 1]].astype('<i2').tobytes() for i in range(4)]
    payloads = [b'', b'a', b'ab' * 3, bytes(1000), bytes(range(256)) * 40,
                squiggle(rng, 3000), squiggle(rng, 40000), real[0][:60000], real[1],
                rng.integers(0, 256, 70000, dtype=np.uint8).tobytes(),       # incompressible
                (b'deepbinner ' * 9000),                                       # long matches
                bytes(rng.integers(0, 4, 50000, dtype=np.uint8)),              # tiny alphabet
                b'\x00' * 100000,                                              # RLE, distance 1
                real[2][:20000] + bytes(33000) + real[2][:20000]]              # distance ~ 32 K
    # long matches at distances near the window size: a step of the resolve kernel (64 tokens, up
    # to 16.5 KB) then wraps around its 32 KiB ring onto bytes its own first matches still read
    # (dbh_inflate_core.h: ring_hazard) - 258-byte matches back to back, and far matches followed
    # by literals inside one step
    far = rng.integers(0, 256, 32000, dtype=np.uint8).tobytes()
    payloads += [far + far, far + far[:774] + b'Q' + far[1000:1600] + b'literal' + far[5000:9000],
                 far[:32768 - 3] + far[:20000] + far[100:400],
                 far[:17000] + far[:17000] + far[:17000]]
    for data in payloads:
        for level in (0, 1, 6, 9):
            cases.append((zlib.compress(data, level), len(data), data))
    for data in payloads[5:9]:
        for strategy in (zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            for mem_level in (1, 9):
                c = zlib.compressobj(6, zlib.DEFLATED, 15, mem_level, strategy)
                cases.append((c.compress(data) + c.flush(), len(data), data))
        c = zlib.compressobj(1, zlib.DEFLATED, 9)            # a 512-byte window
        cases.append((c.compress(data) + c.flush(), len(data), data))
    stream, data = raw_deflate_stream([('stored', b'abc' * 50), ('dynamic', squiggle(rng, 5000)),
                                       ('fixed', b'xyz' * 400), ('stored', b''),
                                       ('dynamic', squiggle(rng, 9000))])
    cases.append((stream, len(data), data))
    # what the one-wavefront-per-stream decoder meets at its seams: hundreds of short blocks (a
    # block boundary inside nearly every chunk of 64 homes, dynamic / fixed / stored in turn, empty
    # stored blocks between them), one-byte blocks, and a read of 1.5 M samples (a long DNA read:
    # hundreds of chunks, every block full)
    many = [(('dynamic', 'fixed', 'stored')[k % 3], squiggle(rng, int(rng.integers(1, 400))))
            for k in range(300)]
    stream, data = raw_deflate_stream(many)
    cases.append((stream, len(data), data))
    stream, data = raw_deflate_stream([('dynamic', b'a'), ('fixed', b'b'), ('stored', b'c')] * 20)
    cases.append((stream, len(data), data))
    long_read = squiggle(rng, 1500000)
    cases.append((zlib.compress(long_read, 1), len(long_read), long_read))
    cases.append((zlib.compress(long_read, 1), 12288, long_read[:12288]))
    # fewer bytes wanted than the stream holds (a partial last chunk; scanning read starts only),
    # and more (MinKNOW's short final chunk: libhdf5 zero-extends it)
    for data in (payloads[6], payloads[10], real[0][:30000]):
        comp = zlib.compress(data, 1)
        for cap in (1, 2, 1000, len(data) - 1):
            cases.append((comp, cap, data[:cap]))
        cases.append((comp, len(data) + 1000, data))         # (the harness reports the bytes the
    return cases                                             # stream holds; extension is phase 2's)


def damaged_cases():
    rng = np.random.default_rng(11)
    good = zlib.compress(squiggle(rng, 20000), 1)
    cases = [good[:k] for k in (0, 1, 2, 5, 6, 100, len(good) // 2, len(good) - 5, len(good) - 1)]
    cases.append(b'\x78\x9d' + good[2:])                     # FCHECK
    cases.append(b'\x79\x01' + good[2:])
    cases.append(b'\x78\x20' + good[2:])                     # preset dictionary
    cases.append(good[:-4] + b'\0\0\0\0')                    # Adler-32
    cases.append(good[:2] + b'\x07' + good[3:])              # block type 3
    cases.append(b'\x78\x01\x01\x05\x00\x00\x00hello' + struct.pack('>I', zlib.adler32(b'hello')))
    cases.append(b'\x78\x01\x01\x05\x00\xfa\xffhello' + struct.pack('>I', zlib.adler32(b'hello')))
    cases.append(good + b'bytes behind the end of the stream')      # (zlib ignores them)
    for _ in range(400):
        flipped = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            flipped[int(rng.integers(2, len(good)))] ^= 1 << int(rng.integers(0, 8))
        cases.append(bytes(flipped))
    for _ in range(100):
        cases.append(b'\x78\x01' + rng.integers(0, 256, int(rng.integers(1, 400)),
                                                dtype=np.uint8).tobytes())
    return cases


def run_harness(cases, tmp_path):
    path_in, path_out = str(tmp_path / 'cases.bin'), str(tmp_path / 'results.bin')
    with open(path_in, 'wb') as f:
        f.write(struct.pack('<I', len(cases)))
        for stream, cap in cases:
            f.write(struct.pack('<II', len(stream), cap) + stream)
    done = subprocess.run([HARNESS, path_in, path_out], capture_output=True, text=True)
    assert done.returncode == 0, done.stderr[-2000:]
    results = []
    with open(path_out, 'rb') as f:
        blob = f.read()
    at = 0
    for _ in cases:
        status, ended, adler_ok, n_tokens, n_bytes = struct.unpack_from('<iiiII', blob, at)
        at += 20
        results.append((status, ended, adler_ok, n_tokens, blob[at:at + n_bytes]))
        at += n_bytes
    assert at == len(blob)
    return results


needs_harness = pytest.mark.skipif(not os.path.exists(HARNESS),
                                   reason='oracle/_build/inflate_host_test not built (make -C oracle)')


@needs_harness
def test_decoder_core_matches_zlib_on_valid_streams(tmp_path):
    cases = valid_cases()
    results = run_harness([(stream, cap) for stream, cap, _ in cases], tmp_path)
    for k, ((stream, cap, want), (status, ended, adler_ok, n_tokens, got)) in enumerate(
            zip(cases, results)):
        assert status == 0, (k, status, len(stream), cap)
        assert got == want, (k, len(got), len(want))
        whole = cap >= len(zlib.decompress(stream))
        assert ended == (1 if whole else 0), (k, ended, cap)
        assert adler_ok == (1 if whole else -1)
        assert n_tokens <= max(len(want), 1)


@needs_harness
def test_resolve_schedules_on_patchwork_streams(tmp_path):
    """The harness's models of both forms of kernel 2 (which copy of a byte every read sees) on
    streams built for them: the tokens resolved as the kernels schedule them give zlib's bytes."""
    cases = patchwork_cases()
    results = run_harness([(stream, cap) for stream, cap, _ in cases], tmp_path)
    for k, ((stream, cap, want), (status, ended, adler_ok, n_tokens, got)) in enumerate(zip(cases, results)):
        assert status == 0, (k, status)
        assert got == want, k


@needs_harness
def test_decoder_core_rejects_what_zlib_rejects(tmp_path):
    """Damaged streams: where zlib reports an error (bad header, bad block, bad codes, truncation,
    checksum) the decoder reports one too and hands out nothing; where zlib still decodes - a
    flipped bit behind the end of the data, say - the bytes are zlib's."""
    cases = damaged_cases()
    cap = 100000
    results = run_harness([(stream, cap) for stream in cases], tmp_path)
    rejected = accepted = 0
    for k, (stream, (status, ended, adler_ok, n_tokens, got)) in enumerate(zip(cases, results)):
        try:
            want = zlib.decompress(stream)
        except zlib.error:
            want = None
        if want is None:
            assert status != 0 and got == b'', (k, status, len(got))
            rejected += 1
        elif len(want) > cap:
            assert status == 0 and got == want[:cap] and ended == 0, (k, status)
        else:
            assert status == 0 and got == want and ended == 1 and adler_ok == 1, (k, status)
            accepted += 1
    assert rejected > 300 and accepted >= 1


def patchwork_cases(n=120, seed=23):
    """Streams made to exercise kernel 2's second form (dbh_inflate_core.h, "Phase 2's second
    form"): random bytes, copies of earlier stretches from 1 byte to the whole window back (short
    and long, so that sources lie in the step itself, in the 8 KiB ring, or in the flushed output),
    runs with periods 1-7 (matches that overlap themselves), zeros; at several levels, strategies
    and window sizes, some cut short by `wanted`."""
    rng = np.random.default_rng(seed)
    cases = []
    for k in range(n):
        size = int(rng.integers(200, 120000))
        data = bytearray()
        while len(data) < size:
            kind = int(rng.integers(0, 6))
            if kind == 0 or not data:
                data += rng.integers(0, 256, int(rng.integers(1, 60)), dtype=np.uint8).tobytes()
            elif kind == 1:                                    # a copy from anywhere in the window
                d = int(rng.integers(1, min(len(data), 32768) + 1))
                ln = int(rng.integers(3, 12)) if rng.random() < 0.8 else int(rng.integers(12, 1200))
                for _ in range(ln):
                    data.append(data[-d])
            elif kind == 2:                                    # a near copy (inside a step)
                d = int(rng.integers(1, min(len(data), 300) + 1))
                for _ in range(int(rng.integers(3, 9))):
                    data.append(data[-d])
            elif kind == 3:                                    # a short period, repeated
                period = rng.integers(0, 256, int(rng.integers(1, 8)), dtype=np.uint8).tobytes()
                data += period * int(rng.integers(2, 80))
            elif kind == 4:
                data += bytes(int(rng.integers(1, 700)))
            else:                                              # squiggle-like: high byte repeats
                m = int(rng.integers(4, 400))
                data += squiggle(rng, m)
        data = bytes(data[:size])
        level = (1, 1, 6, 9)[k % 4]
        strategy = (zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE,
                    zlib.Z_FIXED)[k % 5]
        wbits = (15, 15, 15, 13, 9)[(k // 3) % 5]
        c = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
        stream = c.compress(data) + c.flush()
        cap = len(data) if k % 7 else int(rng.integers(1, len(data) + 1))
        cases.append((stream, cap, data[:cap]))
    return cases


# ---- the kernels themselves ---------------------------------------------------------------------
def pack_streams(hip, cases):
    """[(stream bytes, wanted bytes, mode)] -> (comp buffer, stream records, output size, where
    each stream's output lies)."""
    records = np.zeros(len(cases), dtype=hip.INFLATE_STREAM)
    comp, at, out_at, places = bytearray(), 0, 0, []
    for k, (stream, cap, mode) in enumerate(cases):
        records[k] = (at, len(stream), out_at, cap, mode, 0)
        comp += stream
        at += len(stream)
        places.append((out_at, cap))
        out_at += cap + (cap & 1)                     # outputs start at even bytes
    return np.frombuffer(bytes(comp) if comp else b'\0', dtype=np.uint8), records, out_at, places


@pytest.fixture(params=['lane+pre', 'wave+pre', 'wave+rounds', 'wave+pre+pair'])
def kernel1(request, monkeypatch):
    """Both forms of kernel 1 - one lane per stream, one wavefront per stream - and both forms of
    kernel 2 - short matches read at the step boundary, every match through the rounds - as two
    launches, and the default forms as ONE launch of a pair of waves per stream, kernel 2
    resolving a stream's tokens while kernel 1 still decodes it (dbh_inflate.hip) - whichever of
    them are the defaults."""
    parts = request.param.split('+')
    monkeypatch.setenv('DEEPBINNER_INFLATE_KERNEL', parts[0])
    monkeypatch.setenv('DEEPBINNER_INFLATE_RESOLVE', parts[1])
    monkeypatch.setenv('DEEPBINNER_INFLATE_PAIR', '1' if 'pair' in parts else '0')
    return request.param


@pytest.mark.gpu
def test_gpu_inflate_matches_zlib(hip, kernel1):
    """Every valid case of the CPU test, all in one launch (lanes of one wave at different block
    types, lengths and states), plus stored-as-is streams and zero-extension."""
    cases = valid_cases()
    batch = [(stream, cap, hip.INFLATE_ZLIB) for stream, cap, _ in cases]
    want = [w for _, _, w in cases]
    rng = np.random.default_rng(3)
    raw = squiggle(rng, 5000)
    batch += [(raw, len(raw), hip.INFLATE_STORED), (raw, len(raw) + 77, hip.INFLATE_STORED),
              (raw, 100, hip.INFLATE_STORED), (b'', 10, hip.INFLATE_STORED)]
    want += [raw, raw + bytes(77), raw[:100], bytes(10)]
    comp, records, out_bytes, places = pack_streams(hip, batch)
    # one stream per lane, and lanes that take several one after the other (each at the first
    # block boundary behind the end of its stream - while its neighbours are inside theirs)
    for per_lane in (0, 3, 8, 1000):
        out, status, ms = hip.inflate(comp, records, out_bytes, per_lane)
        assert (status == 0).all(), (per_lane, np.nonzero(status)[0][:10])
        for k, ((at, cap), w) in enumerate(zip(places, want)):
            got = out[at:at + cap].tobytes()
            assert got == w + bytes(cap - len(w)), (per_lane, k, len(w), cap)     # zero-extended
        assert ms > 0


@pytest.mark.gpu
def test_gpu_inflate_patchwork_streams(hip, kernel1):
    """Streams built for kernel 2's second form (sources in the step, in the ring, in the flushed
    output; self-overlapping, long and cut matches): every byte as zlib gives it."""
    cases = patchwork_cases()
    comp, records, out_bytes, places = pack_streams(
        hip, [(stream, cap, hip.INFLATE_ZLIB) for stream, cap, _ in cases])
    out, status, _ = hip.inflate(comp, records, out_bytes)
    assert (status == 0).all(), np.nonzero(status)[0][:10]
    for k, ((at, cap), (_, _, want)) in enumerate(zip(places, cases)):
        assert out[at:at + cap].tobytes() == want, k


@pytest.mark.gpu
def test_gpu_inflate_rejects_what_zlib_rejects(hip, kernel1):
    cases = damaged_cases()
    cap = 100000
    comp, records, out_bytes, places = pack_streams(hip, [(s, cap, hip.INFLATE_ZLIB) for s in cases])
    for per_lane in (0, 5):
        out, status, _ = hip.inflate(comp, records, out_bytes, per_lane)
        rejected = 0
        for k, (stream, (at, _)) in enumerate(zip(cases, places)):
            try:
                want = zlib.decompress(stream)
            except zlib.error:
                want = None
            got = out[at:at + cap].tobytes()
            if want is None:
                assert status[k] != 0 and got == bytes(cap), (per_lane, k, status[k])
                rejected += 1
            else:
                assert status[k] == 0 and got == want[:cap] + bytes(cap - len(want[:cap])), k
        assert rejected > 300


@pytest.mark.gpu
def test_gpu_inflate_a_container_of_reads(hip, kernel1):
    """4,000 squiggles of 2,000-60,000 samples, deflated at level 1 (what MinKNOW and h5py's
    gzip=1 write): every byte as zlib gives it; the rate goes to the log."""
    rng = np.random.default_rng(17)
    pool = [squiggle(rng, int(rng.integers(2000, 60000))) for _ in range(400)]
    deflated = [zlib.compress(p, 1) for p in pool]
    picks = rng.integers(0, len(pool), 4000)
    batch = [(deflated[j], len(pool[j]), hip.INFLATE_ZLIB) for j in picks]
    comp, records, out_bytes, places = pack_streams(hip, batch)
    out, status, ms = hip.inflate(comp, records, out_bytes)
    assert (status == 0).all()
    for (at, cap), j in zip(places, picks):
        assert out[at:at + cap].tobytes() == pool[j]
    print('gpu inflate (kernel 1: one %s per stream): 4,000 streams, %.1f MB out, %.2f ms in the kernels = %.0f streams/s, '
          '%.2f GB/s of output' % (kernel1, out_bytes / 1e6, ms, 4000 / (ms * 1e-3), out_bytes / ms / 1e6))

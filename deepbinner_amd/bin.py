"""
``deepbinner bin``: split a FASTA/FASTQ file of basecalled reads into one gzipped file per barcode,
using the table that ``deepbinner classify`` wrote (behaviour of the
reference's ``deepbinner/bin.py:26-205``).

What stays as the reference has it: the classification table is tab separated with ``read_id`` and
the call in the first two columns (a header row and short rows are skipped, ``none`` means no
barcode, anything else must be an integer: bin.py:34-66); the read id is the first UUID anywhere
in a record's header line (:107,129-132); records are two (FASTA) or four (FASTQ) lines, taken as
they come (:121-127); one ``barcodeNN.<type>.gz`` / ``unclassified.<type>.gz`` per class that
occurs in the table, refused if it or its un-gzipped name exists (:69-82,93-98); the messages and
the closing ``Barcode / Reads / File`` table (:182-205).

What is this package's own: the input is scanned in 16 MB blocks with one compiled pattern per
record instead of line by line, and every output is written once, already compressed - 1 MB
blocks deflated on worker threads (libdeflate when the system has it, else zlib; both release
the GIL) and appended in order as gzip members -
where the reference writes plain files and runs ``gzip``/``pigz`` over them afterwards (:111-113,
186-198).  The decompressed contents are byte for byte what the reference writes.

One deliberate difference: a read that is missing from the table makes the reference fail with a
``KeyError`` (bin.py:134-143 looks up an output file it never opened; its summary code at :192-193
shows that such reads were meant to be counted as ``not found`` and left out).  Here they are
counted, left out and listed in the table with no file.
"""

import collections
import concurrent.futures
import ctypes
import gzip
import os
import pathlib
import re
import sys
import threading
import zlib

from .misc import usable_cpus

UUID = re.compile(rb'[0-9a-fA-F]{8}-[0-9a-fA-F]{4}-[0-9a-fA-F]{4}-[0-9a-fA-F]{4}-[0-9a-fA-F]{12}')
RECORD = {'fasta': re.compile(rb'([^\n]*\n)[^\n]*\n'),
          'fastq': re.compile(rb'([^\n]*\n)[^\n]*\n[^\n]*\n[^\n]*\n')}
LINES = {'fasta': 2, 'fastq': 4}
READ_BLOCK = 16 << 20
MEMBER_BYTES = 1 << 20          # uncompressed bytes per gzip member
MAX_PENDING = 8                 # members per output that may be in flight
NOT_FOUND = 'not found'


def bin_reads(args):
    """Entry point of the ``bin`` sub-command (reference bin.py:26-32)."""
    classes = load_classifications(args.classes)
    names = sorted({class_name(c) for c in classes.values()})
    kind = get_sequence_file_type(args.reads)
    targets = get_output_filenames(names, args.out_dir, kind)
    make_output_dir(args.out_dir, targets)
    counts = write_read_files(args.reads, classes, targets, kind,
                              threads=int(getattr(args, 'threads', 0) or 0))
    print_summary(counts, targets)


def class_name(call):
    """None -> unclassified, 3 -> barcode03 (reference bin.py:85-91)."""
    if call is None:
        return 'unclassified'
    return call if isinstance(call, str) else 'barcode%02d' % call


class_to_class_names = class_name       # the reference's name for it


def _looks_like_uuid(text):
    return len(text) == 36 and all(text[k] == '-' for k in (8, 13, 18, 23))


def load_classifications(class_filename):
    """{read_id: int | None} from the first two columns of the table (reference bin.py:35-66)."""
    print('\nLoading classifications...', end='', flush=True)
    if not pathlib.Path(class_filename).is_file():
        sys.exit('Error: {} does not exist'.format(class_filename))
    classes, odd_id = {}, False
    with open(class_filename, 'rt') as table:
        for row in table:
            cells = row.strip().split('\t')
            if len(cells) < 2 or cells[0].lower() == 'read_id':
                continue
            read_id, call = cells[0], cells[1]
            odd_id = odd_id or not _looks_like_uuid(read_id)
            if call == 'none':
                classes[read_id] = None
                continue
            try:
                classes[read_id] = int(call)
            except ValueError:
                sys.exit('Error: read {} has a non-integer bin of {}'.format(read_id, call))
    print(' done')
    if odd_id:
        print('Warning: one or more read IDs did not conform to the expected format (UUID)')
    print('{:,} total classifications found'.format(len(classes)))
    print()
    return classes


def get_compression_type(filename):
    """'plain' or 'gz' by magic number; bzip2 and zip are refused (reference misc.py:39-58)."""
    with open(filename, 'rb') as f:
        magic = f.read(4)
    if magic.startswith(b'BZh'):
        sys.exit('Error: cannot use bzip2 format - use gzip instead')
    if magic.startswith(b'PK\x03\x04'):
        sys.exit('Error: cannot use zip format - use gzip instead')
    return 'gz' if magic.startswith(b'\x1f\x8b\x08') else 'plain'


def get_open_function(filename):
    return gzip.open if get_compression_type(filename) == 'gz' else open


def get_sequence_file_type(filename):
    """'fasta' or 'fastq' by the first character (reference bin.py:163-179)."""
    if not pathlib.Path(filename).is_file():
        sys.exit('Error: could not find ' + filename)
    with get_open_function(filename)(filename, 'rb') as f:
        first = f.read(1)
    if first == b'>':
        return 'fasta'
    if first == b'@':
        return 'fastq'
    sys.exit('Error: could not determine file format (should be fasta or fastq)')


def get_output_filenames(class_names, out_dir, input_type):
    """{class name: path WITHOUT the .gz} in the given order (reference bin.py:93-98)."""
    return collections.OrderedDict(
        (name, str(pathlib.Path(out_dir) / (name + '.' + input_type))) for name in class_names)


def make_output_dir(out_dir, out_filenames):
    if pathlib.Path(out_dir).is_file():
        sys.exit('Error: {} is an existing file'.format(out_dir))
    if not pathlib.Path(out_dir).is_dir():
        try:
            os.makedirs(out_dir, exist_ok=True)
            print('Making output directory: {}/'.format(out_dir))
        except OSError:
            sys.exit('Error: unable to create output directory {}'.format(out_dir))
    for target in out_filenames.values():
        for candidate in (target, target + '.gz'):
            if pathlib.Path(candidate).exists():
                sys.exit('Error: {} already exists'.format(candidate))
    print()


class _LibDeflate:
    """gzip members through libdeflate when the system has it (about twice zlib's speed at the
    same level; ctypes releases the GIL around the call), one compressor per worker thread.
    DEEPBINNER_BIN_DEFLATE=zlib keeps it out."""

    def __init__(self):
        self.lib = None
        if os.environ.get('DEEPBINNER_BIN_DEFLATE') == 'zlib':
            return
        try:
            lib = ctypes.CDLL('libdeflate.so.0')
            lib.libdeflate_alloc_compressor.restype = ctypes.c_void_p
            lib.libdeflate_alloc_compressor.argtypes = [ctypes.c_int]
            lib.libdeflate_gzip_compress.restype = ctypes.c_size_t
            lib.libdeflate_gzip_compress.argtypes = [ctypes.c_void_p, ctypes.c_char_p,
                                                     ctypes.c_size_t, ctypes.c_char_p,
                                                     ctypes.c_size_t]
            lib.libdeflate_gzip_compress_bound.restype = ctypes.c_size_t
            lib.libdeflate_gzip_compress_bound.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        except (OSError, AttributeError):
            return
        self.lib = lib
        self.local = threading.local()

    def member(self, data):
        """One gzip member holding `data`, or None if libdeflate is not to be had."""
        if self.lib is None:
            return None
        compressor = getattr(self.local, 'compressor', None)
        if compressor is None:
            compressor = self.local.compressor = self.lib.libdeflate_alloc_compressor(6)
            if not compressor:
                return None
        room = self.lib.libdeflate_gzip_compress_bound(compressor, len(data))
        out = ctypes.create_string_buffer(room)
        used = self.lib.libdeflate_gzip_compress(compressor, data, len(data), out, room)
        return out.raw[:used] if used else None


_DEFLATE = None


def _deflater():
    global _DEFLATE
    if _DEFLATE is None:
        _DEFLATE = _LibDeflate()
    return _DEFLATE


class GzipSink:
    """One output file.  Text is collected up to MEMBER_BYTES, deflated on a pool thread as a
    gzip member of its own and appended in submission order."""

    def __init__(self, path, pool):
        self.file = open(path, 'wb')
        self.pool = pool
        self.buffer = bytearray()
        self.pending = collections.deque()
        self.members = 0

    @staticmethod
    def _member(data):
        return _deflater().member(data) or gzip.compress(data, compresslevel=6, mtime=0)

    def write(self, data):
        self.buffer += data
        if len(self.buffer) >= MEMBER_BYTES:
            self._submit()

    def _submit(self):
        self.pending.append(self.pool.submit(self._member, bytes(self.buffer)))
        self.members += 1
        self.buffer.clear()
        while self.pending and (self.pending[0].done() or len(self.pending) > MAX_PENDING):
            self.file.write(self.pending.popleft().result())

    def close(self):
        if self.buffer or self.members == 0:      # an empty output is still a valid .gz
            self._submit()
        while self.pending:
            self.file.write(self.pending.popleft().result())
        self.file.close()


def _records(reads_filename, input_type):
    """(header line, whole record) as bytes, record = 2 or 4 lines exactly as they are in the
    file.  The last line may lack its newline (the reference copies it as it is)."""
    pattern, carry = RECORD[input_type], b''
    with get_open_function(reads_filename)(reads_filename, 'rb') as f:
        while True:
            block = f.read(READ_BLOCK)
            if not block:
                break
            data = carry + block if carry else block
            end = 0
            for m in pattern.finditer(data):
                end = m.end()
                yield m.group(1), m.group(0)
            carry = data[end:]
    if carry:
        lines = carry.split(b'\n')
        if len(lines) != LINES[input_type] or not lines[-1]:
            sys.exit('Error: {} ends in the middle of a record'.format(reads_filename))
        yield lines[0], carry


def write_read_files(reads_filename, classifications, out_filenames, input_type, threads=0):
    """Deals the records out; returns {class name: reads} (reference bin.py:101-153)."""
    by_id = {read_id.encode(): class_name(call) for read_id, call in classifications.items()}
    counts = collections.defaultdict(int)
    workers = threads if threads > 0 else max(1, min(32, usable_cpus()))
    total, next_report = 0, 0
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        sinks = {name: GzipSink(path + '.gz', pool) for name, path in out_filenames.items()}
        try:
            for header, record in _records(reads_filename, input_type):
                if total >= next_report:
                    print_progress(total)
                    next_report = total + 1000
                total += 1
                found = UUID.search(header)
                if found is None:
                    sys.exit('Error: could not find read ID in header: {}'.format(
                        header.decode(errors='replace')))
                name = by_id.get(found.group(0), NOT_FOUND)
                counts[name] += 1
                if name != NOT_FOUND:
                    sinks[name].write(record)
        except (OSError, EOFError, zlib.error) as e:
            sys.exit('Error: could not read {}: {}'.format(reads_filename, e))
        finally:
            for sink in sinks.values():
                sink.close()
    print_progress(total, carriage_return=False)
    print('\n')
    return counts


def print_progress(count, carriage_return=True):
    print('Writing reads: {:,} '.format(count), end='\r' if carriage_return else '')


def print_summary(bin_counts, out_filenames):
    """The closing table (reference bin.py:182-205; the files are compressed already)."""
    print('Gzipping reads:')
    print('  Barcode       Reads     File')
    rows = [(name, path + '.gz') for name, path in out_filenames.items()]
    if NOT_FOUND in bin_counts:
        rows.append((NOT_FOUND, ''))
    for name, path in rows:
        shown = 'none' if name == 'unclassified' else name
        print('  {:<9} {:>9}     {}'.format(shown, bin_counts.get(name, 0), path))
    print()

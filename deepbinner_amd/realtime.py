"""
``deepbinner realtime`` — watch a directory during a sequencing run and sort fast5 files into
``barcodeNN/`` / ``unclassified/`` — mirror of the reference's ``deepbinner/realtime.py``
(``realtime`` :28-70, ``classify_and_move`` :81-108, ``move_classified_fast5s`` :111-143).

Same polling loop, 20,000-single / 5-multi file caps per pass, ``ignore_files`` bookkeeping,
directory names, progress and error text.  Difference: multi-read fast5 files are read directly
with this package's HDF5 reader (``load_fast5s.iter_reads``) instead of being unpacked through the
external ``multi_to_single_fast5`` tool into a temporary directory (reference :183-190); since a
multi-read file holds reads of different barcodes it cannot be *moved* into one bin, so each pass
appends ``read_id<TAB>barcode<TAB>source_file`` lines to ``<out_dir>/multi_read_classifications.tsv``
and, when ``multi_to_single_fast5`` is installed, still unpacks and bins exactly as the reference.
"""

import collections
import os
import pathlib
import shutil
import subprocess
import sys
import tempfile
import time

from .classify import load_and_check_models, classify_fast5_files, set_tensorflow_threads, \
    classify_read_batch, chunker, print_classification_progress
from .load_fast5s import determine_single_or_multi_fast5s, iter_reads
from .misc import print_summary_table

POLL_SECONDS = 5


def realtime(args):
    print()
    args.verbose = False
    nested_out_dir = pathlib.Path(args.in_dir) in pathlib.Path(args.out_dir).parents

    set_tensorflow_threads(args)
    start_model, start_input_size, end_model, end_input_size, output_size, model_count = \
        load_and_check_models(args.start_model, args.end_model, args.scan_size,
                              out_dest=sys.stdout)

    make_output_dir(args.out_dir)
    try:
        waiting = False
        ignore_files = set()
        while True:
            fast5s = look_for_new_fast5s(args.in_dir, args.out_dir, nested_out_dir)
            fast5s = [x for x in fast5s if x not in ignore_files]
            single_or_multi = determine_single_or_multi_fast5s(fast5s)

            if fast5s:
                time.sleep(POLL_SECONDS)  # let any in-flight file moves finish
                classify_and_move(fast5s, single_or_multi, args, start_model, start_input_size,
                                  end_model, end_input_size, output_size, ignore_files)
                waiting = False
            elif args.stop:
                break
            else:
                if waiting:
                    print('.', end='', flush=True)
                else:
                    print('\nWaiting for new fast5 files (Ctrl-C to stop)', end='', flush=True)
                    waiting = True
                time.sleep(POLL_SECONDS)
    except KeyboardInterrupt:
        print('\n\nStopping Deepbinner real-time binning\n')


def look_for_new_fast5s(in_dir, out_dir, nested_out_dir):
    in_dir_fast5s = [str(x) for x in sorted(pathlib.Path(in_dir).glob('**/*.fast5'))]
    if nested_out_dir:
        out_dir_fast5s = set(str(x) for x in sorted(pathlib.Path(out_dir).glob('**/*.fast5')))
        in_dir_fast5s = [x for x in in_dir_fast5s if x not in out_dir_fast5s]
    return in_dir_fast5s


def classify_and_move(fast5s, single_or_multi, args, start_model, start_input_size, end_model,
                      end_input_size, output_size, ignore_files):
    print()
    print('Found {:,} fast5 files in {}'.format(len(fast5s), args.in_dir))

    # Work on a subset per pass so files start moving soon (reference realtime.py:86-94).
    if single_or_multi == 'single':
        fast5s = fast5s[:20000]
    elif single_or_multi == 'multi':
        fast5s = fast5s[:5]
    else:
        assert False

    if single_or_multi == 'multi' and shutil.which('multi_to_single_fast5') is None:
        ignore_files.update(fast5s)
        classifications = classify_multi_read_fast5s(fast5s, args, start_model, start_input_size,
                                                     end_model, end_input_size, output_size)
        print()
        print_summary_table(classifications, output=sys.stdout)
        return

    with tempfile.TemporaryDirectory() as temp_single_read_dir:
        if single_or_multi == 'multi':
            ignore_files.update(fast5s)
            fast5s = unpack_multi_read_fast5s(fast5s, temp_single_read_dir)

        classifications, read_id_to_fast5_file = \
            classify_fast5_files(fast5s, start_model, start_input_size, end_model, end_input_size,
                                 output_size, args, full_output=False, verified_single_read=True)
        print()
        move_classified_fast5s(classifications, read_id_to_fast5_file, args, fast5s, ignore_files)
        print_summary_table(classifications, output=sys.stdout)


def classify_multi_read_fast5s(fast5s, args, start_model, start_input_size, end_model,
                               end_input_size, output_size):
    """Classify every read of the given multi-read files straight from the container."""
    reads = []
    for path in fast5s:
        for read_id, signal in iter_reads(path):
            reads.append((read_id, signal, path))
    classifications = {}
    total = max(len(reads), 1)
    print_classification_progress(0, total, 'reads', out_dest=sys.stdout)
    rows = []
    for batch in chunker(reads, args.batch_size):
        ids = [r[0] for r in batch]
        classify_read_batch(ids, [r[1] for r in batch], start_model, start_input_size, end_model,
                            end_input_size, output_size, args, classifications)
        rows += ['{}\t{}\t{}'.format(r[0], classifications[r[0]], r[2]) for r in batch]
        print_classification_progress(len(classifications), total, 'reads', out_dest=sys.stdout)
    with open(os.path.join(args.out_dir, 'multi_read_classifications.tsv'), 'at') as out:
        for row in rows:
            print(row, file=out)
    return classifications


def move_classified_fast5s(classifications, read_id_to_fast5_file, args, fast5s, ignore_files):
    move_count, fail_move_already_exists, fail_move_other_reason = 0, 0, 0
    counts = collections.defaultdict(int)
    for read_id, barcode_call in classifications.items():
        fast5_file = read_id_to_fast5_file[read_id]

        out_dir = pathlib.Path(args.out_dir) / get_directory_name(barcode_call)
        if not out_dir.is_dir():
            try:
                os.makedirs(str(out_dir))
            except OSError:
                sys.exit('Error: unable to create output directory {}'.format(out_dir))

        dest_filepath = out_dir / pathlib.Path(fast5_file).name
        if dest_filepath.is_file():
            fail_move_already_exists += 1
            ignore_files.add(fast5_file)
        else:
            try:
                shutil.move(fast5_file, str(out_dir))
                move_count += 1
            except OSError:
                fail_move_other_reason += 1

        counts[barcode_call] += 1
        print_moving_progress(move_count, len(fast5s))

    print()
    print_moving_error_messages(fail_move_already_exists, fail_move_other_reason, args.out_dir)

    if fail_move_other_reason == len(fast5s):
        sys.exit('Error: no files were successfully moved to {}'.format(args.out_dir))


def get_directory_name(barcode_call):
    if barcode_call == 'none':
        return 'unclassified'
    return 'barcode{:02d}'.format(int(barcode_call))


def print_moving_error_messages(already_exists, other_reason, out_dir):
    if already_exists == 1:
        print('Error: could not move 1 fast5 file because it already exists in {}'.format(out_dir))
    elif already_exists > 1:
        print('Error: could not move {} fast5 files because they already exist '
              'in {}'.format(already_exists, out_dir))
    if other_reason == 1:
        print('Error: failed to move 1 fast5 file to {}'.format(out_dir))
    elif other_reason > 1:
        print('Error: failed to move {} fast5 files to {}'.format(other_reason, out_dir))


def make_output_dir(out_dir):
    if pathlib.Path(out_dir).is_file():
        sys.exit('Error: {} is an existing file'.format(out_dir))
    if not pathlib.Path(out_dir).is_dir():
        try:
            os.makedirs(out_dir, exist_ok=True)
            print()
            print('Making output directory: {}/'.format(out_dir))
        except OSError:
            sys.exit('Error: unable to create output directory {}'.format(out_dir))


def print_moving_progress(completed, total):
    percent = 100.0 * completed / total
    print('\rMoving fast5s:      {} / {} ({:.1f}%)'.format(completed, total, percent),
          end='', flush=True)


def unpack_multi_read_fast5s(fast5s, temp_single_read_dir):
    print('Unpacking fast5s with multi_to_single_fast5:')
    for fast5 in fast5s:
        subprocess.check_output(['multi_to_single_fast5', '-i', fast5, '-s', temp_single_read_dir])
    print()
    return [str(x) for x in sorted(pathlib.Path(temp_single_read_dir).glob('**/*.fast5'))]

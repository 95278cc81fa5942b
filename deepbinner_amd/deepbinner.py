"""
Command line entry point — the ``classify`` and ``realtime`` subcommands of the reference's
``deepbinner/deepbinner.py`` with the same flags, defaults and validation messages
(reference ``deepbinner.py:90-156`` for the options, ``:283-345`` for the checks and preset
resolution).  The other reference subcommands (bin, prep, balance, train, refine) are outside the
GPU hot path and are not provided.
"""

import argparse
import pathlib
import sys

from .version import __version__


def main(argv=None):
    parser = argparse.ArgumentParser(
        prog='deepbinner',
        description='Deepbinner: a deep convolutional neural network barcode demultiplexer for '
                    'Oxford Nanopore reads (MI355X / HIP implementation of the classify path)',
        add_help=False)
    subparsers = parser.add_subparsers(title='Commands', dest='subparser_name')
    classify_subparser(subparsers)
    realtime_subparser(subparsers)

    help_args = parser.add_argument_group('Help')
    help_args.add_argument('-h', '--help', action='help', default=argparse.SUPPRESS,
                           help='Show this help message and exit')
    help_args.add_argument('--version', action='version', version=__version__,
                           help="Show program's version number and exit")

    argv = sys.argv[1:] if argv is None else list(argv)
    if not argv:
        parser.print_help(file=sys.stderr)
        sys.exit(1)
    if argv[0] in ('bin', 'prep', 'balance', 'train', 'refine'):
        sys.exit('Error: the {} command is not part of this build - it covers the classify and '
                 'realtime commands only'.format(argv[0]))
    args = parser.parse_args(argv)

    if args.subparser_name == 'classify':
        check_classify_and_realtime_arguments(args)
        from .classify import classify
        classify(args)
    elif args.subparser_name == 'realtime':
        check_classify_and_realtime_arguments(args)
        from .realtime import realtime
        realtime(args)


def classify_subparser(subparsers):
    group = subparsers.add_parser('classify', description='Classify fast5 reads', add_help=False)
    positional_args = group.add_argument_group('Positional')
    positional_args.add_argument('input', type=str,
                                 help='One of the following: a single fast5 file, a directory of '
                                      'fast5 files (will be searched recursively) or a '
                                      'tab-delimited file of training data')
    classify_and_realtime_options(group)
    other_args = group.add_argument_group('Other')
    other_args.add_argument('--verbose', action='store_true',
                            help='Include the output probabilities for all barcodes in the '
                                 'results (default: just show the final barcode call)')
    other_args.add_argument('-h', '--help', action='help', default=argparse.SUPPRESS,
                            help='Show this help message and exit')


def realtime_subparser(subparsers):
    group = subparsers.add_parser('realtime', description='Sort fast5 files during sequencing',
                                  add_help=False)
    required_args = group.add_argument_group('Required')
    required_args.add_argument('--in_dir', type=str, required=True,
                               help='Directory where sequencer deposits fast5 files')
    required_args.add_argument('--out_dir', type=str, required=True,
                               help='Directory to output binned fast5 files')
    classify_and_realtime_options(group)
    other_args = group.add_argument_group('Other')
    other_args.add_argument('--stop', action='store_true',
                            help='Automatically stop when there are no more input reads (default: '
                                 'continue to run and wait for more reads)')
    other_args.add_argument('-h', '--help', action='help', default=argparse.SUPPRESS,
                            help='Show this help message and exit')


def classify_and_realtime_options(group):
    """Options shared by classify and realtime (reference deepbinner.py:109-156)."""
    model_args = group.add_argument_group('Model presets')
    model_args.add_argument('--native', action='store_true',
                            help='Preset for EXP-NBD103 read start and end models')
    model_args.add_argument('--rapid', action='store_true',
                            help='Preset for SQK-RBK004 read start model')

    model_args = group.add_argument_group('Models (at least one is required if not using a preset)')
    model_args.add_argument('-s', '--start_model', type=str, required=False,
                            help='Model trained on the starts of reads')
    model_args.add_argument('-e', '--end_model', type=str, required=False,
                            help='Model trained on the ends of reads')

    barcode_args = group.add_argument_group('Barcoding')
    barcode_args.add_argument('--scan_size', type=float, required=False, default=6144,
                              help="This much of a read's start/end signal will examined for "
                                   "barcode signals")
    barcode_args.add_argument('--score_diff', type=float, required=False, default=0.5,
                              help='For a read to be classified, there must be this much '
                                   'difference between the best and second-best barcode scores')

    two_model_args = group.add_argument_group('Two model (read start and read end) behaviour')
    two_model_args.add_argument('--require_either', action='store_true',
                                help='Most lenient approach: a barcode call on either the start '
                                     'or end is sufficient to classify a read, as long as they do '
                                     'not disagree on the barcode (default behaviour)')
    two_model_args.add_argument('--require_start', action='store_true',
                                help='Moderate approach: a start barcode is required to classify '
                                     'a read but an end barcode is optional')
    two_model_args.add_argument('--require_both', action='store_true',
                                help='Most stringent approach: both start and end barcodes must be '
                                     'present and agree to classify a read')

    perf_args = group.add_argument_group('Performance')
    perf_args.add_argument('--batch_size', type=int, required=False, default=256,
                           help='Number of reads handed to the GPU per call')
    perf_args.add_argument('--loader_procs', type=int, required=False, default=0,
                           help='Threads (native reader) or processes (Python reader) that load '
                                'and decompress fast5 files ahead of the GPU (0 = automatic)')
    # TensorFlow knobs of the reference: accepted for command-line compatibility, ignored.
    perf_args.add_argument('--intra_op_parallelism_threads', type=int, required=False, default=12,
                           help='Accepted for compatibility with the TensorFlow build (ignored)')
    perf_args.add_argument('--inter_op_parallelism_threads', type=int, required=False, default=1,
                           help='Accepted for compatibility with the TensorFlow build (ignored)')
    perf_args.add_argument('--device_count', type=int, required=False, default=1,
                           help='Accepted for compatibility with the TensorFlow build (ignored)')
    perf_args.add_argument('--omp_num_threads', type=int, required=False, default=12,
                           help='Accepted for compatibility with the TensorFlow build (ignored)')


def check_classify_and_realtime_arguments(args):
    """Reference deepbinner.py:283-317 (same messages; default two-model mode is
    require_either, deepbinner.py:315-316)."""
    if args.native and args.rapid:
        sys.exit('Error: you can only use one model preset (--native or --rapid)')
    if args.native or args.rapid:
        preset_name = 'native' if args.native else 'rapid'
        if args.start_model is not None or args.end_model is not None:
            sys.exit('Error: you cannot explicitly specify a model and '
                     'also use a model preset (--{})'.format(preset_name))
    if args.native:
        args.start_model = find_native_start_model()
        args.end_model = find_native_end_model()
    if args.rapid:
        args.start_model = find_rapid_start_model()

    model_count = (args.start_model is not None) + (args.end_model is not None)
    if model_count == 0:
        sys.exit('Error: you must provide at least one model')
    if args.score_diff <= 0.0 or args.score_diff > 1.0:
        sys.exit('Error: --score_diff must be in the range (0, 1] (greater than 0 and less than or '
                 'equal to 1)')
    for flag in ('require_either', 'require_start', 'require_both'):
        if model_count < 2 and getattr(args, flag):
            sys.exit('Error: --{} can only be used with two models (start and end)'.format(flag))
    if two_model_args_used(args) > 1:
        sys.exit('Error: only one of the following options can be used: --require_either, '
                 '--require_start, --require_both')
    if two_model_args_used(args) == 0:
        args.require_either = True
    assert two_model_args_used(args) == 1


def find_native_start_model():
    return find_model('EXP-NBD103_read_starts')


def find_native_end_model():
    return find_model('EXP-NBD103_read_ends')


def find_rapid_start_model():
    return find_model('SQK-RBK004_read_starts')


def find_model(model_name):
    """Look where the reference looks (``models/`` beside or inside the package, reference
    deepbinner.py:332-345) for the Keras file, then for this package's converted ``.dbw``."""
    here = pathlib.Path(__file__).resolve()
    for base in (here.parents[1] / 'models', here.parents[0] / 'models'):
        for name in (model_name, model_name + '.dbw'):
            if (base / name).is_file():
                return str(base / name)
    sys.exit('Error: could not find {} - did Deepbinner install correctly?'.format(model_name))


def two_model_args_used(args):
    return sum(1 for flag in (args.require_either, args.require_start, args.require_both) if flag)


if __name__ == '__main__':
    main()

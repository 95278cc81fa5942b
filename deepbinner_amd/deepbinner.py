"""
Command line of the classify path: the ``classify`` and ``realtime`` sub-commands with the flags,
defaults, validation and error messages of the reference's ``deepbinner/deepbinner.py``
(:90-156 options, :283-345 checks and preset resolution), so that existing command lines keep
working, plus ``bin`` (:159-176), the consumer of the table ``classify`` writes.  The options are
declared as data (``OPTIONS``) and turned into argparse calls by one loop.  The reference's
training-side sub-commands (prep, balance, train, refine) are outside the GPU hot path and answer
with a one-line refusal.
"""

import argparse
import pathlib
import sys

from .version import __version__

NOT_PROVIDED = ('prep', 'balance', 'train', 'refine')
PRESETS = {'native': ('EXP-NBD103_read_starts', 'EXP-NBD103_read_ends'),
           'rapid': ('SQK-RBK004_read_starts', None)}
TWO_MODEL_FLAGS = ('require_either', 'require_start', 'require_both')
_IGNORED = 'Accepted for compatibility with the TensorFlow build (ignored)'
_HELP = (('-h', '--help'), dict(action='help', default=argparse.SUPPRESS,
                                help='Show this help message and exit'))

# (group title, [(flags, argparse keywords)]) shared by classify and realtime
OPTIONS = [
    ('Model presets', [
        (('--native',), dict(action='store_true',
                             help='Preset for EXP-NBD103 read start and end models')),
        (('--rapid',), dict(action='store_true', help='Preset for SQK-RBK004 read start model')),
    ]),
    ('Models (at least one is required if not using a preset)', [
        (('-s', '--start_model'), dict(type=str, help='Model trained on the starts of reads')),
        (('-e', '--end_model'), dict(type=str, help='Model trained on the ends of reads')),
    ]),
    ('Barcoding', [
        (('--scan_size',), dict(type=float, default=6144,
                                help="This much of a read's start/end signal will examined for "
                                     "barcode signals")),
        (('--score_diff',), dict(type=float, default=0.5,
                                 help='For a read to be classified, there must be this much '
                                      'difference between the best and second-best barcode '
                                      'scores')),
    ]),
    ('Two model (read start and read end) behaviour', [
        (('--require_either',), dict(action='store_true',
                                     help='Most lenient approach: a barcode call on either the '
                                          'start or end is sufficient to classify a read, as long '
                                          'as they do not disagree on the barcode (default '
                                          'behaviour)')),
        (('--require_start',), dict(action='store_true',
                                    help='Moderate approach: a start barcode is required to '
                                         'classify a read but an end barcode is optional')),
        (('--require_both',), dict(action='store_true',
                                   help='Most stringent approach: both start and end barcodes '
                                        'must be present and agree to classify a read')),
    ]),
    ('Performance', [
        (('--batch_size',), dict(type=int, default=256,
                                 help='Number of reads handed to the GPU per call')),
        (('--loader_procs',), dict(type=int, default=0,
                                   help='Threads (native reader) or processes (Python reader) '
                                        'that load and decompress fast5 files ahead of the GPU '
                                        '(0 = automatic)')),
        (('--devices',), dict(type=int, default=0,
                              help='GPUs to spread the batches over, one model replica each '
                                   '(0 = DEEPBINNER_DEVICES or one)')),
        # TensorFlow knobs of the reference
        (('--intra_op_parallelism_threads',), dict(type=int, default=12, help=_IGNORED)),
        (('--inter_op_parallelism_threads',), dict(type=int, default=1, help=_IGNORED)),
        (('--device_count',), dict(type=int, default=1, help=_IGNORED)),
        (('--omp_num_threads',), dict(type=int, default=12, help=_IGNORED)),
    ]),
]

# per sub-command: description, groups in front of the shared ones (None in that list: no shared
# ones), the 'Other' group behind them
COMMANDS = {
    'classify': ('Classify fast5 reads',
                 [('Positional', [
                     (('input',), dict(type=str,
                                       help='One of the following: a single fast5 file, a '
                                            'directory of fast5 files (will be searched '
                                            'recursively) or a tab-delimited file of training '
                                            'data'))])],
                 [(('--verbose',), dict(action='store_true',
                                        help='Include the output probabilities for all barcodes '
                                             'in the results (default: just show the final '
                                             'barcode call)')), _HELP]),
    'realtime': ('Sort fast5 files during sequencing',
                 [('Required', [
                     (('--in_dir',), dict(type=str, required=True,
                                          help='Directory where sequencer deposits fast5 files')),
                     (('--out_dir',), dict(type=str, required=True,
                                           help='Directory to output binned fast5 files'))])],
                 [(('--stop',), dict(action='store_true',
                                     help='Automatically stop when there are no more input reads '
                                          '(default: continue to run and wait for more reads)')),
                  _HELP]),
    'bin': ('Bin fasta/q reads',
            [('Required', [
                (('--classes',), dict(type=str, required=True,
                                      help='Deepbinner classification file (made with the '
                                           'deepbinner classify command)')),
                (('--reads',), dict(type=str, required=True, help='FASTA or FASTQ reads')),
                (('--out_dir',), dict(type=str, required=True,
                                      help='Directory to output binned read files'))]),
             None],
            [(('--threads',), dict(type=int, default=0,
                                   help='Threads that compress the output (0 = automatic)')),
             _HELP]),
}
USES_MODELS = ('classify', 'realtime')


def _add_groups(parser, groups):
    for title, options in groups:
        group = parser.add_argument_group(title)
        for flags, keywords in options:
            group.add_argument(*flags, **keywords)


def build_parser():
    parser = argparse.ArgumentParser(
        prog='deepbinner',
        description='Deepbinner: a deep convolutional neural network barcode demultiplexer for '
                    'Oxford Nanopore reads (MI355X / HIP implementation of the classify path)',
        add_help=False)
    subparsers = parser.add_subparsers(title='Commands', dest='subparser_name')
    for name, (description, leading, other) in COMMANDS.items():
        sub = subparsers.add_parser(name, description=description, add_help=False)
        shared = [] if None in leading else OPTIONS
        _add_groups(sub, [g for g in leading if g] + shared + [('Other', other)])
    _add_groups(parser, [('Help', [
        _HELP, (('--version',), dict(action='version', version=__version__,
                                     help="Show program's version number and exit"))])])
    return parser


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    parser = build_parser()
    if not argv:
        parser.print_help(file=sys.stderr)
        sys.exit(1)
    if argv[0] in NOT_PROVIDED:
        sys.exit('Error: the {} command is not part of this build - it covers the classify, '
                 'realtime and bin commands only'.format(argv[0]))
    args = parser.parse_args(argv)
    if args.subparser_name in USES_MODELS:
        check_classify_and_realtime_arguments(args)
    if args.subparser_name == 'classify':
        from .classify import classify as run
    elif args.subparser_name == 'realtime':
        from .realtime import realtime as run
    elif args.subparser_name == 'bin':
        from .bin import bin_reads as run
    else:
        return
    run(args)


def check_classify_and_realtime_arguments(args):
    """Preset resolution and option checks with the reference's messages (deepbinner.py:283-317);
    with two models and no mode given the mode is require_either (:315-316)."""
    chosen = [name for name in PRESETS if getattr(args, name)]
    if len(chosen) > 1:
        sys.exit('Error: you can only use one model preset (--native or --rapid)')
    if chosen:
        if args.start_model is not None or args.end_model is not None:
            sys.exit('Error: you cannot explicitly specify a model and '
                     'also use a model preset (--{})'.format(chosen[0]))
        start, end = PRESETS[chosen[0]]
        args.start_model = find_model(start)
        args.end_model = find_model(end) if end else None

    n_models = sum(m is not None for m in (args.start_model, args.end_model))
    if n_models == 0:
        sys.exit('Error: you must provide at least one model')
    if not 0.0 < args.score_diff <= 1.0:
        sys.exit('Error: --score_diff must be in the range (0, 1] (greater than 0 and less than or '
                 'equal to 1)')
    given = [flag for flag in TWO_MODEL_FLAGS if getattr(args, flag)]
    if given and n_models < 2:
        sys.exit('Error: --{} can only be used with two models (start and end)'.format(given[0]))
    if len(given) > 1:
        sys.exit('Error: only one of the following options can be used: --require_either, '
                 '--require_start, --require_both')
    if not given:
        args.require_either = True
    assert two_model_args_used(args) == 1


def two_model_args_used(args):
    return sum(bool(getattr(args, flag)) for flag in TWO_MODEL_FLAGS)


def find_model(model_name):
    """Where the reference looks - ``models/`` beside or inside the package (deepbinner.py:332-345)
    - first for the Keras file, then for this package's converted ``.dbw``."""
    here = pathlib.Path(__file__).resolve()
    candidates = [base / name
                  for base in (here.parents[1] / 'models', here.parents[0] / 'models')
                  for name in (model_name, model_name + '.dbw')]
    for path in candidates:
        if path.is_file():
            return str(path)
    sys.exit('Error: could not find {} - did Deepbinner install correctly?'.format(model_name))


if __name__ == '__main__':
    main()

"""Small shared helpers — mirror of the parts of the reference's ``deepbinner/misc.py`` that the
classify / realtime path uses (``print_summary_table``, reference ``misc.py:19-36``)."""

import collections
import sys


def print_summary_table(classifications, output=None):
    output = sys.stderr if output is None else output      # (looked up when called, not imported)
    counts = collections.Counter(classifications.values())
    numeric = sorted(int(b) for b in counts if _is_int(b))
    other = sorted(b for b in counts if not _is_int(b))
    print('', file=output)
    print('Barcode     Count', file=output)
    for barcode in numeric + other:
        print('{:>7} {:>9}'.format(barcode, counts[str(barcode)]), file=output)
    print('', file=output)


def _is_int(text):
    try:
        int(text)
        return True
    except ValueError:
        return False

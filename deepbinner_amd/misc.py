"""Small shared helpers — mirror of the parts of the reference's ``deepbinner/misc.py`` that the
classify / realtime path uses (``print_summary_table``, reference ``misc.py:19-36``)."""

import collections
import os
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


def usable_cpus():
    """Hardware threads this process can keep busy: the online CPUs cut down to the scheduler's
    affinity mask and to the cgroup's CPU quota.  ``os.cpu_count()`` alone says 256 inside a
    container that is allowed 16 - and thread teams sized by it run slower than teams of 16
    (every thread beyond the quota only adds throttling stalls; profiles/r03_cpu_capacity.txt)."""
    cpus = os.cpu_count() or 1
    try:
        cpus = min(cpus, len(os.sched_getaffinity(0)) or cpus)
    except (AttributeError, OSError):
        pass
    quota = period = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:                       # cgroup v2
            first, second = f.read().split()[:2]
            if first != 'max':
                quota, period = int(first), int(second)
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f:      # cgroup v1
                quota = int(f.read())
            with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
                period = int(f.read())
        except (OSError, ValueError):
            quota = period = None
    if quota and period and quota > 0 and period > 0:
        cpus = min(cpus, max(1, -(-quota // period)))
    return max(1, cpus)

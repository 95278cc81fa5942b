#!/usr/bin/env python3
"""What lies between two forward launches of bench.py's steps: from a rocprofv3 --kernel-trace
--memory-copy-trace run (csv), the gap from the end of one dbh_forward_kernel to the start of the
next and the kernels / copies inside it.  Usage: python tools/step_gap.py DIR"""
import csv
import glob
import sys


def rows(pattern):
    for f in glob.glob(sys.argv[1] + '/**/' + pattern, recursive=True):
        yield from csv.DictReader(open(f))


def main():
    events = []
    for r in rows('*kernel_trace.csv'):
        events.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]))
    for r in rows('*memory_copy_trace.csv'):
        events.append((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                       'copy ' + r.get('Direction', '') + ' ' + r.get('Bytes', '')))
    events.sort()
    fwd = [e for e in events if 'dbh_forward_kernel' in e[2]]
    gaps = []
    for a, b in zip(fwd[5:], fwd[6:]):
        inside = [(e[2], (e[0] - a[1]) / 1e3, (e[1] - e[0]) / 1e3) for e in events
                  if a[1] <= e[0] < b[0] and e is not b]
        gaps.append(((b[0] - a[1]) / 1e3, inside))
    gaps.sort(key=lambda g: g[0])
    mid = gaps[len(gaps) // 2]
    print('launches: %d   gap between forward kernels (us): median %.1f  min %.1f  max %.1f'
          % (len(fwd), mid[0], gaps[0][0], gaps[-1][0]))
    print('kernel duration (us): median %.1f' % sorted((e[1] - e[0]) / 1e3 for e in fwd)[len(fwd) // 2])
    print('inside the median gap (name, starts after the kernel ended, lasts; us):')
    for name, at, dur in mid[1]:
        print('   %-45s %8.1f %8.1f' % (name, at, dur))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Static census of the forward kernel's ISA between barriers: how many MFMA, other VALU, LDS,
vector-memory and scalar instructions each phase of the (fully unrolled, nearly branch-free) kernel
carries.  Input: the .s file of `hipcc --save-temps` (tools/isa_census.sh builds it).
Usage: python tools/isa_census.py file.s [kernel-symbol-substring]"""
import collections
import re
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else '_ZN3dbh18dbh_forward_kernel'
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith(want) and l.rstrip().endswith(':') or
                 (l.startswith(want) and ':' in l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.size'))
    seg = collections.Counter()
    total = collections.Counter()
    rows = []
    k = 0
    label = 'entry'
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith(';') or s.startswith('.'):
            continue
        if s.endswith(':') or re.match(r'^\.?LBB\d+_\d+:', s):
            rows.append((k, label, dict(seg)))
            seg = collections.Counter()
            label = s
            continue
        op = s.split()[0]
        if op.startswith('v_mfma'):
            c = 'mfma'
        elif op.startswith('v_pk_'):
            c = 'valu_pk'
        elif op.startswith('v_'):
            c = 'valu'
        elif op.startswith('ds_read') or op.startswith('ds_load'):
            c = 'ds_read'
        elif op.startswith('ds_'):
            c = 'ds_write'
        elif op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'):
            c = 'vmem'
        elif op == 's_waitcnt':
            c = 's_waitcnt'
        elif op == 's_nop':
            c = 's_nop'
        elif op == 's_barrier':
            c = 's_barrier'
        elif op.startswith('s_cbranch') or op.startswith('s_branch'):
            c = 's_branch'
        elif op.startswith('s_'):
            c = 'salu'
        else:
            c = 'other'
        seg[c] += 1
        total[c] += 1
        if c == 's_barrier':
            rows.append((k, label, dict(seg)))
            seg = collections.Counter()
            k += 1
            label = 'after barrier %d' % k
    rows.append((k, label, dict(seg)))
    cols = ['mfma', 'valu', 'valu_pk', 'ds_read', 'ds_write', 'vmem', 'salu', 's_waitcnt', 's_nop', 's_branch']
    print('%-28s' % 'segment' + ''.join('%9s' % c for c in cols))
    for k, label, d in rows:
        if sum(d.values()) == 0:
            continue
        print('%-28s' % label[:28] + ''.join('%9d' % d.get(c, 0) for c in cols))
    print('%-28s' % 'TOTAL (static)' + ''.join('%9d' % total.get(c, 0) for c in cols))


if __name__ == '__main__':
    main()

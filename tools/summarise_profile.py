#!/usr/bin/env python3
"""Collects what tools/profile_gpu.sh left under gpurun_out/ into a profiles/<name>/ directory:
rocprofv3 kernel stats, a per-launch PMC summary (mean over the forward kernel's dispatches) and
the HBM traffic figure bench.py reports (profiles/pmc_traffic.json).
Usage: python tools/summarise_profile.py profiles/r01_v5"""
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, 'gpurun_out')
KERNEL = 'dbh_forward_kernel'


def pmc_summary():
    out, durs = {}, {}
    for d in sorted(glob.glob(os.path.join(SRC, 'prof_pmc_*'))):
        if not os.path.isdir(d):
            continue
        path = os.path.join(d, 'pmc_counter_collection.csv')
        if not os.path.exists(path):
            continue
        sums, counts = {}, {}
        with open(path) as f:
            for row in csv.DictReader(f):
                # full 256-window launches only (the last launch of a bench step is partial)
                if KERNEL not in row['Kernel_Name'] or int(row['Grid_Size']) != 256 * 512:
                    continue
                name = row['Counter_Name']
                sums[name] = sums.get(name, 0.0) + float(row['Counter_Value'])
                counts[name] = counts.get(name, 0) + 1
        for name in sums:
            out[name] = sums[name] / counts[name]
        trace = os.path.join(d, 'pmc_kernel_trace.csv')
        if os.path.exists(trace):
            with open(trace) as f:
                t = [int(r['End_Timestamp']) - int(r['Start_Timestamp'])
                     for r in csv.DictReader(f)
                     if KERNEL in r['Kernel_Name'] and int(r['Grid_Size_X']) == 256 * 512]
            if t:
                durs['_dur_ns_' + os.path.basename(d)] = sum(t) / len(t)
    out.update(durs)
    return out


def main():
    dst = os.path.join(REPO, sys.argv[1])
    os.makedirs(dst, exist_ok=True)
    stats = os.path.join(SRC, 'prof_stats', 'bench_kernel_stats.csv')
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(dst, 'rocprofv3_kernel_stats.csv'))
    for name in ('stage_times_256.txt', 'stage_times_4096.txt'):
        if os.path.exists(os.path.join(SRC, name)):
            shutil.copy(os.path.join(SRC, name), os.path.join(dst, name))
    log = os.path.join(SRC, 'prof_stats_bench.log')
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith('{')]
        if lines:
            open(os.path.join(dst, 'bench_under_rocprofv3.json'), 'w').write(lines[-1])
    summary = {'dbh::' + KERNEL: pmc_summary()}
    json.dump(summary, open(os.path.join(dst, 'pmc_summary.json'), 'w'), indent=1, sort_keys=True)
    s = summary['dbh::' + KERNEL]
    if 'FETCH_SIZE' in s and 'WRITE_SIZE' in s:
        # MI355X_MICROARCH.md, HBM/rocprofv3 section: FETCH_SIZE / WRITE_SIZE are in KiB-like
        # 1,024-byte units and FETCH_SIZE under-reports by 2x on gfx950
        traffic = {
            'kernel': KERNEL,
            'windows_per_launch': 256,
            'FETCH_SIZE_KB': s['FETCH_SIZE'],
            'WRITE_SIZE_KB': s['WRITE_SIZE'],
            'hbm_bytes_per_launch': (2.0 * s['FETCH_SIZE'] + s['WRITE_SIZE']) * 1024.0,
            'source': os.path.relpath(dst, REPO) + '/pmc_summary.json',
            'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes '
                    '(tools/profile_gpu.sh), averaged over the 256-window launches; FETCH_SIZE '
                    'doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); '
                    'algorithmic bytes per launch = 256 x (2048 B int16 in + 52 B probs + 4 B '
                    'call) = 538,624 B in fused seam-b2 mode',
        }
        json.dump(traffic, open(os.path.join(REPO, 'profiles', 'pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps(s, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()

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
def windows_per_launch():
    """Windows one forward launch of the PMC passes carried: from the bench line those passes
    printed (roofline.windows_per_launch; bench.py queues --steps-per-launch steps as one launch)."""
    for log in sorted(glob.glob(os.path.join(SRC, 'prof_pmc_*.log'))):
        for line in open(log, errors='replace'):
            if line.startswith('{'):
                try:
                    return int(round(json.loads(line)['roofline']['windows_per_launch']))
                except (ValueError, KeyError):
                    pass
    return 10000


WINDOWS_PER_LAUNCH = windows_per_launch()
CLOCK_GHZ = 2.4


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
                # the persistent launches of bench.py's steps: 256 workgroups x 512 threads, each
                # launch = WINDOWS_PER_LAUNCH windows (profile runs skip bench.py's side rates)
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
    for name in ('timeline6_groups.txt',):
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
            'windows_per_launch': WINDOWS_PER_LAUNCH,
            'FETCH_SIZE_KB': s['FETCH_SIZE'],
            'WRITE_SIZE_KB': s['WRITE_SIZE'],
            'hbm_bytes_per_launch': (2.0 * s['FETCH_SIZE'] + s['WRITE_SIZE']) * 1024.0,
            'algorithmic_bytes_per_launch': WINDOWS_PER_LAUNCH * (2048 + 52 + 4),
            'source': os.path.relpath(dst, REPO) + '/pmc_summary.json',
            'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes '
                    '(tools/profile_gpu.sh), averaged over the persistent launches (windows_per_launch each) of '
                    'bench.py --config 1; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 '
                    'counts 128-B requests as 64 B); algorithmic bytes per window = 2048 B int16 '
                    'in + 52 B probs + 4 B call in fused seam-b2 mode',
        }
        # matrix-pipe busy cycles (summed over the chip's 1,024 SIMDs): a constant of the build per
        # window (32 cycles x the MFMAs issued); bench.py divides it by ITS launch duration - the
        # PMC passes themselves run ~8 % slower than an unprofiled launch
        dur = s.get('_dur_ns_prof_pmc_SQ_WAVE_CYCLES')
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in s and dur:
            traffic['mfma_busy_cycles_per_window'] = s['SQ_VALU_MFMA_BUSY_CYCLES'] / WINDOWS_PER_LAUNCH
            traffic['mfma_pipe_util_in_the_pmc_pass'] = (
                s['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * dur * CLOCK_GHZ))
            traffic['pmc_pass_launch_ns'] = dur
            traffic['mfma_per_window_measured'] = s.get('SQ_INSTS_MFMA', 0) / WINDOWS_PER_LAUNCH
            traffic['valu_per_mfma'] = s.get('SQ_INSTS_VALU', 0) / max(s.get('SQ_INSTS_MFMA', 1), 1)
            traffic['lds_conflict_fraction'] = (s.get('SQ_LDS_BANK_CONFLICT', 0) /
                                                max(s.get('SQ_LDS_IDX_ACTIVE', 1), 1))
        json.dump(traffic, open(os.path.join(REPO, 'profiles', 'pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps(s, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()

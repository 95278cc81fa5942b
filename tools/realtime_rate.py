#!/usr/bin/env python3
"""`deepbinner realtime` (start + end models, table only) over containers written on the spot: the
rate of the containers by which a small and a large run differ (the process's start cancels).
Usage: python tools/realtime_rate.py READS_PER_CONTAINER MEAN_SAMPLES [small large]"""
import contextlib
import io
import json
import os
import shutil
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))


def main():
    reads, mean = int(sys.argv[1]), int(sys.argv[2])
    small, large = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (4, 20)
    import multi_read_rate
    from deepbinner_amd import deepbinner as cli
    import deepbinner_amd.realtime as realtime
    tmp = tempfile.mkdtemp(prefix='rt_rate_')
    dirs = {}
    for name, count in (('small', small), ('large', large)):
        dirs[name] = os.path.join(tmp, name)
        os.makedirs(dirs[name])
        for k in range(count):
            multi_read_rate.write_with_own_writer(os.path.join(dirs[name], 'c%02d.fast5' % k), reads, mean, 100 + k)
    models = os.path.join(REPO, 'deepbinner_amd', 'models')
    realtime.POLL_SECONDS = 0
    os.environ['DEEPBINNER_REALTIME_TABLE_ONLY'] = '1'
    seconds = {}
    for name in ('small', 'large', 'small', 'large'):
        out = os.path.join(tmp, 'out_' + name)
        shutil.rmtree(out, ignore_errors=True)
        argv = ['realtime', '--in_dir', dirs[name], '--out_dir', out, '--stop',
                '-s', os.path.join(models, 'EXP-NBD103_read_starts.dbw'),
                '-e', os.path.join(models, 'EXP-NBD103_read_ends.dbw')]
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(argv)
        seconds[name] = time.perf_counter() - t0
    n = (large - small) * reads
    print(json.dumps({'reads_per_container': reads, 'mean_samples': mean, 'containers': [small, large],
                      'reads_per_s': round(n / (seconds['large'] - seconds['small'])),
                      'host_share_percent': realtime.host_inflate_share(1),
                      'inflate_cus_env': os.environ.get('DEEPBINNER_INFLATE_CUS')}))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()

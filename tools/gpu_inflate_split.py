#!/usr/bin/env python3
"""configs[4] with the chunks inflated on the GPU: reads/s of raw loader -> dispatcher ->
dbh_classify_pair_deflated for ONE setting of (host share of the inflating, device queues, CUs
left to the inflate kernels) - the pipeline deepbinner_amd/realtime.py runs on multi-read files:

    python tools/gpu_inflate_split.py DIR --share 53 --queues 3 --cus 32

DIR: containers written before with --write (h5py where the box has it, the package's own writer
otherwise), so that a sweep of processes reads the same files.
"""
import argparse
import glob
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--write', type=int, default=0, help='write this many containers and stop')
    ap.add_argument('--reads', type=int, default=4000)
    ap.add_argument('--mean-length', type=int, default=27000, help='samples per read (log-normal, 2,000 .. 400,000)')
    ap.add_argument('--share', type=int, default=35, help='%% of the bytes the host inflates')
    ap.add_argument('--queues', type=int, default=0, help='0 = what realtime.py would take')
    ap.add_argument('--cus', type=int, default=32, help='CUs left out of the forward launches')
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--repeat', type=int, default=2)
    ap.add_argument('--loops', type=int, default=1,
                    help='stream the containers this many times per pass (filling and draining the '
                         'pipeline costs ~50 ms a pass: a sixth of a pass of 16 containers)')
    opts = ap.parse_args()
    if opts.write:
        import subprocess
        import multi_read_rate
        os.makedirs(opts.dir, exist_ok=True)
        paths = [os.path.join(opts.dir, 'batch_%02d.fast5' % k) for k in range(opts.write)]
        if os.path.exists(multi_read_rate.CONDA_PYTHON):
            jobs = [subprocess.Popen([multi_read_rate.CONDA_PYTHON, '-c', multi_read_rate.WRITER, p,
                                      str(opts.reads), str(opts.mean_length), str(100 + k)])
                    for k, p in enumerate(paths)]
            if any(j.wait() != 0 for j in jobs):
                sys.exit('writing the containers with h5py failed')
        else:
            for k, p in enumerate(paths):
                multi_read_rate.write_with_own_writer(p, opts.reads, opts.mean_length, 100 + k)
        return
    from deepbinner_amd import classify, fast5_native, hip_backend
    paths = sorted(glob.glob(os.path.join(opts.dir, '*.fast5'))) * max(1, opts.loops)
    team = opts.threads or min(16, classify.usable_cpus())
    import io
    models = os.path.join(REPO, 'deepbinner_amd', 'models')
    sm, _, em, _, _, _ = classify.load_and_check_models(
        os.path.join(models, 'EXP-NBD103_read_starts.dbw'),
        os.path.join(models, 'EXP-NBD103_read_ends.dbw'), 6144, out_dest=io.StringIO())
    from deepbinner_amd import realtime
    if opts.queues:
        os.environ['DEEPBINNER_INFLATE_QUEUES'] = str(opts.queues)
    os.environ['DEEPBINNER_INFLATE_CUS'] = str(opts.cus)
    replicas, _ = realtime.inflate_queues(classify.device_replicas(sm, em), opts.share)

    import threading
    worker_cpu = [0.0]
    lock = threading.Lock()

    def work(item, start_replica, end_replica):
        c0 = time.thread_time()
        _, ids, offsets, _, comp, records = item
        calls, status = hip_backend.classify_pair_deflated(
            start_replica, end_replica, comp, records, offsets, 6144, 0.5)
        assert (status == 0).all()
        with lock:
            worker_cpu[0] += time.thread_time() - c0
        return len(calls)

    # the CPU time of a pass, split: the threads that call the GPU (classify_pair_deflated: record
    # tables, copies queued, the wait), the consumer thread (the stream's Python side + the
    # dispatcher), and the rest of the process - the loader's team and the HIP runtime's own threads
    # (tools/loader_cost.py has the loader's share alone)
    # ... and by thread name, sampled from /proc while the passes run (a thread's last sample stands
    # for it once it is gone): f5-stream = the loader's team (fast5_reader.cpp names it)
    by_name, seen, stop_sampling = {}, {}, threading.Event()
    ticks = os.sysconf('SC_CLK_TCK')

    def sample():
        while not stop_sampling.is_set():
            for tid in os.listdir('/proc/self/task'):
                try:
                    with open('/proc/self/task/%s/stat' % tid) as f:
                        text = f.read()
                    name = text[text.index('(') + 1:text.rindex(')')]
                    fields = text[text.rindex(')') + 2:].split()
                    seen[tid] = (name, (int(fields[11]) + int(fields[12])) / ticks)
                except (OSError, ValueError):
                    pass
            stop_sampling.wait(0.02)

    sampler = threading.Thread(target=sample, daemon=True)
    sampler.start()
    total_done = 0
    best, best_cpu, best_split = 0.0, 0.0, None
    for _ in range(opts.repeat + 1):
        worker_cpu[0] = 0.0
        t0, c0, m0 = time.perf_counter(), time.process_time(), time.thread_time()
        stream = fast5_native.stream_raw(paths, threads=team, depth=len(replicas) + 2,
                                         host_inflate_above=-opts.share)
        done = sum(classify.dispatch_batches(stream, replicas, work))
        wall, cpu, main_cpu = time.perf_counter() - t0, time.process_time() - c0, time.thread_time() - m0
        total_done += done
        if done / wall > best:
            best, best_cpu = done / wall, cpu / done
            best_split = {'gpu_call_threads': round(worker_cpu[0] / done * 1e6, 2),
                          'consumer_thread': round(main_cpu / done * 1e6, 2),
                          'loader_team_and_runtime_threads': round((cpu - worker_cpu[0] - main_cpu) / done * 1e6, 2)}
    stop_sampling.set()
    sampler.join()
    for name, cpu_s in seen.values():
        by_name[name] = by_name.get(name, 0.0) + cpu_s
    by_name = {k: round(v / total_done * 1e6, 2) for k, v in sorted(by_name.items()) if v > 0}
    print(json.dumps({'cpu_us_per_read_by_thread_name_all_passes_and_setup': by_name}), file=sys.stderr)
    print(json.dumps({'host_share_per_cent': opts.share, 'queues': len(replicas),
                      'cus_left_to_inflate': opts.cus, 'containers_per_pass': len(paths),
                      'forward_stream': os.environ.get('DEEPBINNER_FORWARD_STREAM', 'shared'), 'loader_threads': team,
                      'reads_per_s': round(best), 'host_cpu_us_per_read': round(best_cpu * 1e6, 1),
                      'host_cpu_us_per_read_split': best_split}))


if __name__ == '__main__':
    main()

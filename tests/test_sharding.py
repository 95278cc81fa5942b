"""N>1 path on CPU: world_size-2 gloo process group, read sharding + gather of calls."""
import os
import socket
import subprocess
import sys

from conftest import REPO
from deepbinner_amd.sharding import shard_bounds


def test_shard_bounds_partition():
    for n in (0, 1, 7, 37, 10000, 1000003):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1


WORKER = r'''
import os, sys
import numpy as np
import torch
sys.path.insert(0, sys.argv[1])
from deepbinner_amd.sharding import init_process_group, gather_calls, shard_bounds, env_world
from deepbinner_amd.model_format import ModelWeights
from deepbinner_amd import classify
from oracle import network_ref
import argparse

rank, local_rank, world = env_world()
dist = init_process_group('gloo')
reads = np.load(os.path.join(sys.argv[1], 'tests', 'golden', 'reads.npz'))
offs = reads['multi_offsets']
signals = [reads['multi_samples'][offs[i]:offs[i + 1]] for i in range(len(offs) - 1)][:9]
a, b = shard_bounds(len(signals), world, rank)
w, _ = ModelWeights.load(os.path.join(sys.argv[1], 'deepbinner_amd', 'models',
                                      'SQK-RBK004_read_starts.dbw'))

class M:
    def predict(self, x, batch_size=None):
        return network_ref.forward(w, np.asarray(x, dtype=np.float32), dtype=np.float32).astype(np.float32)

args = argparse.Namespace(scan_size=1024, batch_size=8, score_diff=0.5)
calls, _ = classify.call_batch(1024, 13, list(range(a, b)), signals[a:b], M(), args, 'start')
local = torch.tensor([0 if c == 'none' else int(c) for c in calls], dtype=torch.int32)
full = gather_calls(local, len(signals), world, rank)
if rank == 0:
    all_calls, _ = classify.call_batch(1024, 13, list(range(len(signals))), signals, M(), args, 'start')
    want = [0 if c == 'none' else int(c) for c in all_calls]
    assert full.tolist() == want, (full.tolist(), want)
    print('GATHER_OK', full.tolist())
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), str(script), REPO]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'GATHER_OK' in out.stdout


CLI_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import deepbinner_amd.classify as classify
from deepbinner_amd import deepbinner as cli
from conftest import OracleModel
classify.build_model = lambda w: OracleModel(w)      # CPU box: oracle-backed model double
classify.set_tensorflow_threads = lambda args: None  # ... and no GPU to select
cli.main(['classify', '--native', '--verbose', '--batch_size', '3',
          os.path.join(sys.argv[1], 'tests', 'golden', 'fast5', 'single')])
'''


def test_cli_classify_sharded_over_two_ranks(tmp_path):
    """`deepbinner classify DIR` under torchrun with 2 ranks (gloo): rank 0 prints the same TSV
    the single-process run prints, each read exactly once, with the reference's expected calls."""
    from test_oracle_golden import EXPECTED_START
    script = tmp_path / 'cli_worker.py'
    script.write_text(CLI_WORKER)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=REPO, OMP_NUM_THREADS='2', DEEPBINNER_DIST_BACKEND='gloo')
    env.pop('LOCAL_RANK', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), str(script), REPO]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if '\t' in l]
    assert lines[0].startswith('read_ID\tbarcode_call\tstart_none')
    rows = {l.split('\t')[0]: l.split('\t') for l in lines[1:]}
    assert len(lines) == 8 and set(rows) == set(EXPECTED_START)
    # default two-model mode is require_either -> the start column of the reference's tests
    assert {k: v[1] for k, v in rows.items()} == EXPECTED_START
    assert all(len(v) == 30 for v in rows.values())
    assert 'Barcode     Count' in out.stderr

#!/bin/bash
# One-GPU A/B of where bench.py queues the exchange of the calls (the N > 1 step): on a side stream
# against double-buffered call arrays (default) or on the classification stream
# (DEEPBINNER_BENCH_GATHER=instream), with RCCL itself (one rank, both set-up forms), with device
# copies between two "devices" that are the same GPU, and 8 shards on the one GPU in both forms.
# Outputs: gpurun_out/gather_ab/*.json (copy what should be judged into profiles/).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/gather_ab
mkdir -p $OUT
cd $R
FLAGS="--no-cpu-baseline --no-side-rates"
run() { name=$1; shift; "$@" > $OUT/$name.json 2> $OUT/$name.err; python - $OUT/$name.json $name <<'PY'
import json, sys
try:
    d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')][0]
    g = d['gather']
    print('%-34s %9.0f reads/s  %.4f ms/step  gather: %s, %s, %s ms/step' % (
        sys.argv[2], d['value'], d['ms_per_step'], g['transport'], g.get('queued_on'), g.get('ms_per_step')))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run single_gpu_direct python bench.py $FLAGS
for mode in side instream; do
  export DEEPBINNER_BENCH_GATHER=$mode
  DEEPBINNER_COMM_FORCE=1 run rccl_1rank_one_process_$mode python bench.py --gpus 1 $FLAGS
  DEEPBINNER_BENCH_FORCE_RANKS=1 DEEPBINNER_COMM_FORCE=1 run rccl_1rank_torchrun_$mode \
      python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 $FLAGS
  DEEPBINNER_DEVICE_ORDINALS=0,0 run copy_2way_one_gpu_$mode python bench.py --gpus 2 --steps 40 $FLAGS
  DEEPBINNER_DEVICE_ORDINALS=0,0,0,0,0,0,0,0 run copy_8way_one_gpu_$mode python bench.py --gpus 8 --steps 20 $FLAGS
done
unset DEEPBINNER_BENCH_GATHER
DEEPBINNER_DEVICE_ORDINALS=0,0,0,0,0,0,0,0 DEEPBINNER_COMM=host run host_8way_torchrun_ranks \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 8 --steps 20 $FLAGS

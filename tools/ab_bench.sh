#!/bin/bash
# A/B of two builds of libdeepbinner_hip.so on the SAME GPU box (different boxes differ by ~1 %):
#   tools/ab_bench.sh a.so b.so [rounds]   -> avg launch us per build, interleaved rounds
R=${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for lib in $A $B; do
    DEEPBINNER_HIP_LIB=$lib python $R/bench.py --no-cpu-baseline --no-other-configs --no-side-rates | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['roofline']['avg_launch_ms']*1000,2), 'us', round(d['value']))"
  done
done

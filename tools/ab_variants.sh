#!/bin/bash
# A/B of kernel variants on the SAME GPU box (boxes differ by ~1 %).  The variants are built HERE
# (build container) first:   tools/ab_variants.sh build NAME "-DDBH_EXP_X=1" ...
# and compared on the box:   tools/ab_variants.sh run [rounds] NAME...     ("base" = the product)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=$R/deepbinner_amd/csrc/_variants
if [ "$1" = build ]; then
  mkdir -p $V
  shift
  while [ $# -gt 1 ]; do
    make -s -B -C $R/deepbinner_amd/csrc OUT=$V/$1.so EXTRA="$2" $V/$1.so || exit 1
    shift 2
  done
  exit 0
fi
shift
N=$1; shift
for i in $(seq $N); do
  for name in "$@"; do
    lib=$V/$name.so
    [ "$name" = base ] && lib=$R/deepbinner_amd/libdeepbinner_hip.so
    DEEPBINNER_HIP_LIB=$lib timeout 120 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-side-rates --no-other-configs 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$name', round(d['value']), 'reads/s', round(d['roofline']['avg_launch_ms']*1000/d['roofline']['windows_per_launch']*256,2), 'us per 256 windows', d['calls_not_none_rank0'])"
  done
done

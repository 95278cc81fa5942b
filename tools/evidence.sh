#!/bin/bash
# The round's evidence set, on the GPU box (via gpurun), for the kernels as built in-tree:
#   gpurun --timeout 2400 -- 'bash tools/evidence.sh'
# leaves under gpurun_out/evidence/: the GPU test log, smoke(), the bench lines of configs 1-3 and of
# config 1 with one launch per step, the soak against the C port (opt-in test), and - through
# tools/profile_gpu.sh - rocprofv3 kernel stats, the PMC passes and the stage timeline under
# gpurun_out/.  Afterwards, in the build container:
#   rm -rf gpurun_out/prof_pmc_*   BEFORE the run (stale passes pollute the summary), then
#   python tools/summarise_profile.py profiles/<name>; cp gpurun_out/evidence/* profiles/<name>/
R=${GRAFT_REPO_ROOT:-/root/repo}
E=$R/gpurun_out/evidence
mkdir -p $E
cd $R
python -m pytest tests -m gpu -q > $E/pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $E/smoke.log 2>&1
for c in 1 2 3; do
  python bench.py --config $c 2>/dev/null | grep '^{' > $E/bench_config$c.json
done
python bench.py --steps-per-launch 1 2>/dev/null | grep '^{' > $E/bench_config1_one_launch_per_step.json
DEEPBINNER_SOAK=1 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k soak 2>&1 | grep -E "soak:|passed|failed" > $E/soak.log
bash tools/profile_gpu.sh > $E/profile_gpu.log 2>&1
cp $R/gpurun_out/prof_stats_bench.log $E/bench_under_rocprofv3.log
grep -E "passed|failed" $E/pytest_gpu.log | tail -1; cat $E/smoke.log | tail -3; cat $E/soak.log
for f in $E/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], round(d['value']), d['unit'], 'frac', round(d['roofline']['frac'],4), 'avg_launch_ms', round(d['roofline']['avg_launch_ms'],3))
PY
done

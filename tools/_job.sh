cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/e16_pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/e16_pytest.log | tail -2
bash tools/ab_variants.sh run 3 s0f base 2>&1 | tee gpurun_out/ab_e16.txt
python bench.py --no-cpu-baseline --no-other-configs > gpurun_out/e16_bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/e16_bench.json')); print(d['value'], d['roofline']['frac'], d['value_one_launch_per_step']); print(d['roofline'].get('phase_cycles_per_group'))"

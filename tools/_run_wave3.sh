cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/wave3_pytest.log 2>&1
grep -E "passed|failed|error" gpurun_out/wave3_pytest.log | tail -3
cd /tmp; export TMPDIR=/tmp
for shape in lognormal sorted; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wave3_$shape -o s -- python $GRAFT_REPO_ROOT/tools/inflate_rate.py 4000 27000 $shape > /dev/null 2>&1
echo $shape; python - <<PY
import csv,glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/wave3_$shape/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)): print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
done

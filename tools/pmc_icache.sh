R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/prof_pmc_ic -o pmc -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-side-rates > $OUT/prof_pmc_ic.log 2>&1
python - <<'PY'
import csv,glob,collections,os
f=glob.glob(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/prof_pmc_ic/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if 'forward' in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in acc: print(k, acc[k]/n[k], n[k])
PY

R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/inflate_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/tools/inflate_rate.py 4000 27000 uniform > /dev/null 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
 tag=$(echo $set | cut -d' ' -f1)
 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o p -- python $R/tools/inflate_rate.py 4000 27000 uniform > /dev/null 2>&1
done
python - <<'PY'
import csv,glob,collections,os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/inflate_pmc'
for f in glob.glob(root+'/stats/*kernel_stats.csv')+glob.glob(root+'/stats/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)): print(r['Name'][:60], r['Calls'], r['AverageNs'])
for f in sorted(glob.glob(root+'/*/p_counter_collection.csv')+glob.glob(root+'/*/*/p_counter_collection.csv')):
    acc=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        k=('k1' if 'tokens' in r['Kernel_Name'] else 'k2' if 'resolve' in r['Kernel_Name'] else None)
        if k: acc[(k,r['Counter_Name'])]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
    for k in sorted(acc): print(k, acc[k]/n[k])
PY

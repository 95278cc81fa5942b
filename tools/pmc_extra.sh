#!/bin/bash
# Extra PMC passes (GPU box, via gpurun) for the forward kernel under bench.py: VALU/MFMA
# co-execution, scalar and LDS-queue stalls, and the dynamic vector-instruction mix.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_CYCLES" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64" \
           "SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_BUSY_CU_CYCLES"; do
  tag=X_$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/prof_pmc_$tag -o pmc -- \
      python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-side-rates > $OUT/prof_pmc_$tag.log 2>&1
done
python - <<'PY'
import csv,glob,collections,os
root=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out'
for f in sorted(glob.glob(root+'/prof_pmc_X_*/pmc_counter_collection.csv')):
    acc=collections.defaultdict(float); n=collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if 'forward' in r['Kernel_Name'] and int(r['Grid_Size'])==256*512:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
    for k in acc: print('%-34s %16.1f per launch  %10.1f per window'%(k, acc[k]/n[k], acc[k]/n[k]/10000))
PY

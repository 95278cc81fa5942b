#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + PMC passes for bench.py, and the
# per-stage timing tool.  Outputs under gpurun_out/ (copy what should be judged into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/timeline6.py 6 > $OUT/timeline6_groups.txt 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- \
    python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-side-rates > $OUT/prof_stats_bench.log 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/prof_pmc_$tag -o pmc -- \
      python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-side-rates > $OUT/prof_pmc_$tag.log 2>&1
done
ls -R $OUT | head -60

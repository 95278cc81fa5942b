#!/bin/bash
# Per-stage PMC attribution of the forward kernel (GPU box, via gpurun): LDS bank conflicts,
# instruction mix and wait cycles per stage, by differencing truncated launches.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
csvs=""
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/stage_pmc_$tag -o pmc -- \
      python $R/tools/stage_pmc.py run > $OUT/stage_pmc_$tag.log 2>&1
  csvs="$csvs $(find $OUT/stage_pmc_$tag -name 'pmc_counter_collection.csv' | head -1)"
done
python $R/tools/stage_pmc.py report $csvs > $OUT/stage_pmc.json 2>&1
cat $OUT/stage_pmc.json | head -150

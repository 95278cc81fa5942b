set -x
cd $GRAFT_REPO_ROOT
B=$PWD/deepbinner_amd/csrc/_variants/libdeepbinner_fast5_before.so
for s in 2000,9000 20000,34000; do
  t=$(echo $s | tr , _)
  LOADER_COST_SAMPLES=$s LOADER_COST_TAG=after_$t timeout 600 python tools/loader_cost.py 8 > gpurun_out/loader_after_$t.log 2>&1
  LOADER_COST_SAMPLES=$s LOADER_COST_TAG=before_$t DEEPBINNER_FAST5_LIB=$B timeout 600 python tools/loader_cost.py 8 > gpurun_out/loader_before_$t.log 2>&1
done
LOADER_COST_SAMPLES=20000,34000 DEEPBINNER_FAST5_TIMING=1 LOADER_COST_TAG=timing timeout 600 python tools/loader_cost.py 4 > gpurun_out/loader_timing.log 2>&1
tail -4 gpurun_out/loader_*.log | cut -c1-400
nproc; lscpu | grep -i "model name"

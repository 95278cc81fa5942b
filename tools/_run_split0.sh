cd $GRAFT_REPO_ROOT
D=/tmp/containers
python tools/gpu_inflate_split.py $D --write 16 > gpurun_out/split0_write.log 2>&1
for q in 3 6 8; do
  echo "share 0 queues $q: $(python tools/gpu_inflate_split.py $D --share 0 --queues $q --cus 0 2>&1 | tail -1 | cut -c1-300)"
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/split0_trace -o t -- python $GRAFT_REPO_ROOT/tools/gpu_inflate_split.py $D --share 0 --queues 6 --cus 0 > $GRAFT_REPO_ROOT/gpurun_out/split0_trace.log 2>&1
python $GRAFT_REPO_ROOT/tools/kernel_trace_summary.py $GRAFT_REPO_ROOT/gpurun_out/split0_trace
tail -2 $GRAFT_REPO_ROOT/gpurun_out/split0_trace.log | cut -c1-300
